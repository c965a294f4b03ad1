// micro-test: accuracy of v_rsq_f64 and of 1 / 2 Newton refinements (run on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double* x, double* o, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double d = x[i];
  double r0 = __builtin_amdgcn_rsq(d);
  double h = 0.5 * d;
  double r1 = r0 * (1.5 - h * r0 * r0);
  double r2 = r1 * (1.5 - h * r1 * r1);
  o[3 * i] = r0; o[3 * i + 1] = r1; o[3 * i + 2] = r2;
}
int main() {
  const int n = 1 << 16;
  double *x, *o;
  hipMallocManaged(&x, n * 8); hipMallocManaged(&o, 3 * n * 8);
  for (int i = 0; i < n; i++) x[i] = exp(40.0 * (rand() / (double)RAND_MAX - 0.5));
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, x, o, n);
  hipDeviceSynchronize();
  double e[3] = {0, 0, 0};
  for (int i = 0; i < n; i++) { double t = 1.0 / sqrt(x[i]); for (int k = 0; k < 3; k++) e[k] = fmax(e[k], fabs(o[3 * i + k] - t) / t); }
  printf("max rel err: seed %.3e  1 newton %.3e  2 newton %.3e\n", e[0], e[1], e[2]);
  return 0;
}
