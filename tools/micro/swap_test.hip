// Semantics of v_permlane32_swap / v_permlane16_swap on gfx950 with both operands equal (what kernels_fast.hip relies on):
//   permlane32_swap(v, v) -> [0] = v of lane (l & 31)       (rows {0,1}, same position), [1] = v of lane 32 + (l & 31)
//   permlane16_swap(v, v) -> [0] = v of lane (l & ~16)      (even row of this row pair), [1] = v of lane (l | 16)
// build + run on an MI355X:  hipcc --offload-arch=gfx950 swap_test.hip -o swap_test && ./swap_test
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *o) {
  const int l = threadIdx.x;
  auto a = __builtin_amdgcn_permlane32_swap(l, l, false, false);
  auto b = __builtin_amdgcn_permlane16_swap(l, l, false, false);
  o[l] = a[0]; o[64 + l] = a[1]; o[128 + l] = b[0]; o[192 + l] = b[1];
}
int main() {
  int *d, h[256];
  if (hipMalloc(&d, sizeof(h)) != hipSuccess) { printf("no device\n"); return 2; }
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; l++) {
    bad += h[l] != (l & 31);
    bad += h[64 + l] != 32 + (l & 31);
    bad += h[128 + l] != (l & ~16);
    bad += h[192 + l] != (l | 16);
  }
  printf(bad ? "MISMATCH in %d entries\n" : "permlane32_swap / permlane16_swap semantics as documented (%d mismatches)\n", bad);
  return bad ? 1 : 0;
}
