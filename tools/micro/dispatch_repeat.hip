// dispatch_repeat.hip — is the placement of the FIRST workgroups of a launch the same from launch to launch?  (kernels_as.hip would like to
// hand the long waves of a later round to workgroups that land on distinct SIMDs.)  Launches a 4096-block grid of 64-thread workgroups R times
// with three register footprints, records (XCC, SE, CU, SIMD) of every block, and reports for the first 1024 / 1536 / 2048 blocks: how many
// blocks changed their SIMD between launches, and how many of the first-on-their-SIMD blocks are the same set.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/dispatch_repeat.hip -o tools/micro/dispatch_repeat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
template <int REGS>
__global__ void __launch_bounds__(64) k_where(unsigned *out, int spin, double *sink) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  double acc[REGS];  // register footprint
  for (int r = 0; r < REGS; r++) acc[r] = threadIdx.x + r;
  for (int k = 0; k < spin; k++)
    for (int r = 0; r < REGS; r++) acc[r] = acc[r] * 1.0000001 + 1e-9;
  double s = 0.0;
  for (int r = 0; r < REGS; r++) s += acc[r];
  if (s == 12345.678) sink[0] = s;
  if (threadIdx.x == 0) out[blockIdx.x] = ((xcc & 0xF) << 16) | (hw & 0xFFF0);  // xcc | se sh cu simd (wave slot masked out)
}
template <int REGS>
void run(const char *name, int spin) {
  const int B = 4096, R = 4;
  unsigned *d;
  double *sink;
  hipMalloc(&d, B * sizeof(unsigned));
  hipMalloc(&sink, 8);
  std::vector<std::vector<unsigned>> h(R, std::vector<unsigned>(B));
  for (int r = 0; r < R; r++) {
    hipLaunchKernelGGL(k_where<REGS>, dim3(B), dim3(64), 0, 0, d, spin, sink);
    hipDeviceSynchronize();
    hipMemcpy(h[r].data(), d, B * sizeof(unsigned), hipMemcpyDeviceToHost);
  }
  for (int first : {1024, 1536, 2048}) {
    int moved = 0;
    for (int r = 1; r < R; r++)
      for (int b = 0; b < first; b++) moved += h[r][b] != h[0][b];
    std::set<unsigned> seen;
    int firsts = 0;
    for (int b = 0; b < first; b++) firsts += seen.insert(h[0][b]).second;
    printf("%s: first %d blocks: %d of %d placements differ from launch 0 over %d more launches; %d distinct SIMDs\n", name, first, moved, first * (R - 1), R - 1, firsts);
  }
  hipFree(d); hipFree(sink);
}
int main() {
  run<8>("16 doubles of state (~40 VGPRs), long", 20000);
  run<48>("48 doubles (~110 VGPRs: 4 waves per SIMD), long", 4000);
  run<48>("48 doubles, SHORT waves (they leave while the grid is still being placed)", 10);
  return 0;
}
