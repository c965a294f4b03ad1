// dispatch_map.hip — where does workgroup b of a grid of 64-thread workgroups land?  (XCC, SE, CU, SIMD) of every block of a 4096-block
// launch with the register budget of the forward sweep (4 waves per SIMD), read from HW_REG_HW_ID / HW_REG_XCC_ID.  For speed only
// (HIP promises nothing about placement): kernels_as.hip orders a later round's workgroups so that the long waves spread over the SIMDs.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/dispatch_map.hip -o tools/micro/dispatch_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void __launch_bounds__(64, 4) k_where(unsigned *out, int spin) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  // keep the wave resident for a while so that the whole grid is placed as a resident set (4096 = 256 CUs x 16 waves)
  volatile double x = 1.0;
  for (int k = 0; k < spin; k++) x = x * 1.0000001 + 1e-9;
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc & 0xF; }
}
int main() {
  const int B = 4096;
  unsigned *d;
  hipMalloc(&d, 2 * B * sizeof(unsigned));
  // force ~100 VGPRs?  (launch_bounds(64,4) only caps; occupancy here is whatever the kernel needs: tiny) — placement of 64-thread blocks is what we want
  hipLaunchKernelGGL(k_where, dim3(B), dim3(64), 0, 0, d, 20000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(2 * B);
  hipMemcpy(h.data(), d, 2 * B * sizeof(unsigned), hipMemcpyDeviceToHost);
  auto f = [&](int b, int &xcc, int &se, int &sh, int &cu, int &simd, int &wv) {
    const unsigned hw = h[2 * b];
    wv = hw & 0xF; simd = (hw >> 4) & 0x3; cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 0x1; se = (hw >> 13) & 0x7; xcc = h[2 * b + 1];
  };
  printf("block: xcc se sh cu simd wave\n");
  for (int b = 0; b < 80; b++) { int x, se, sh, cu, si, wv; f(b, x, se, sh, cu, si, wv); printf("%4d: %d %d %d %2d %d %2d\n", b, x, se, sh, cu, si, wv); }
  // how many of the FIRST 1024 blocks share a SIMD?  and of blocks b, b + 1024, ...
  for (int first : {512, 1024, 2048}) {
    std::map<unsigned, int> cnt;
    for (int b = 0; b < first; b++) { int x, se, sh, cu, si, wv; f(b, x, se, sh, cu, si, wv); cnt[(x << 16) | (se << 12) | (sh << 10) | (cu << 4) | si]++; }
    int mx = 0; for (auto &kv : cnt) mx = kv.second > mx ? kv.second : mx;
    printf("first %d blocks: %zu distinct SIMDs, at most %d blocks on one SIMD\n", first, cnt.size(), mx);
  }
  return 0;
}
