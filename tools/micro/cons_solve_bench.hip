// cons_solve_bench.hip — the dense consensus solve (Sum_i H_i) du = -Sum_i g_i of one workgroup, variant against variant: accuracy against a
// host Cholesky and time per launch.  The kernels are the library's own (the translation unit is included):
//   hipcc -O3 --offload-arch=gfx950 -I pmpc_amd/csrc tools/micro/cons_solve_bench.hip -o tools/micro/cons_solve_bench
//   PMPC_CONS_REG=0 tools/micro/cons_solve_bench    (the variants the register-resident kernel replaced)
#include "../../pmpc_amd/csrc/kernels_generic.hip"
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
#ifndef NCLIST
#define NCLIST {4, 24, 40, 60, 80, 81, 88, 97, 120, 160, 199, 200, 201, 202, 203, 240, 248, 249, 300}
#endif

int main(int argc, char **argv) {
  std::mt19937_64 rng(7);
  std::normal_distribution<double> nd;
  for (int nc : NCLIST) {
    // SPD matrix with the spread of a condensed Hessian: B'B + diag
    std::vector<double> B((size_t)nc * nc), H((size_t)nc * nc, 0.0), g(nc), x(nc), Lh((size_t)nc * nc, 0.0);
    for (auto &v : B) v = nd(rng);
    for (int r = 0; r < nc; r++)
      for (int c = 0; c < nc; c++) {
        double s = r == c ? 1.0 + (r % 7) : 0.0;
        for (int k = 0; k < nc; k++) s += B[k + (size_t)nc * r] * B[k + (size_t)nc * c] / nc;
        H[r + (size_t)nc * c] = s;
      }
    for (auto &v : g) v = nd(rng);
    // host reference in long double
    std::vector<long double> L((size_t)nc * nc, 0.0L), y(nc);
    for (int c = 0; c < nc; c++) {
      long double d = H[c + (size_t)nc * c];
      for (int k = 0; k < c; k++) d -= L[c + (size_t)nc * k] * L[c + (size_t)nc * k];
      d = sqrtl(d);
      L[c + (size_t)nc * c] = d;
      for (int r = c + 1; r < nc; r++) {
        long double v = H[r + (size_t)nc * c];
        for (int k = 0; k < c; k++) v -= L[r + (size_t)nc * k] * L[c + (size_t)nc * k];
        L[r + (size_t)nc * c] = v / d;
      }
    }
    for (int r = 0; r < nc; r++) {
      long double v = -g[r];
      for (int k = 0; k < r; k++) v -= L[r + (size_t)nc * k] * y[k];
      y[r] = v / L[r + (size_t)nc * r];
    }
    for (int r = nc - 1; r >= 0; r--) {
      long double v = y[r];
      for (int k = r + 1; k < nc; k++) v -= L[k + (size_t)nc * r] * y[k];
      y[r] = v / L[r + (size_t)nc * r];
    }
    double *dH, *dL, *dg, *dx;
    int *dfail;
    hipMalloc(&dH, H.size() * 8); hipMalloc(&dL, (H.size() + 32 * 272) * 8); hipMalloc(&dg, nc * 8); hipMalloc(&dx, nc * 8); hipMalloc(&dfail, 4);
    hipMemcpy(dH, H.data(), H.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dg, g.data(), nc * 8, hipMemcpyHostToDevice);
    hipMemset(dfail, 0, 4);
    hipMemset(dx, 0xff, nc * 8);
    launch_cons_solve(dH, dL, dg, dx, nc, true, dfail, 0);
    hipDeviceSynchronize();
    hipMemcpy(x.data(), dx, nc * 8, hipMemcpyDeviceToHost);
    hipMemcpy(Lh.data(), dL, H.size() * 8, hipMemcpyDeviceToHost);
    int fail = 0;
    hipMemcpy(&fail, dfail, 4, hipMemcpyDeviceToHost);
    double ex = 0.0, nx = 0.0, eL = 0.0;
    for (int r = 0; r < nc; r++) { ex = fmax(ex, fabs(x[r] - (double)y[r])); nx = fmax(nx, fabs((double)y[r])); }
    for (int c = 0; c < nc; c++)
      for (int r = c; r < nc; r++) eL = fmax(eL, fabs(Lh[r + (size_t)nc * c] - (double)L[r + (size_t)nc * c]));  // (lower triangle: what the solves read)
    // the vector-only solve that may follow (stored factor), and the timing
    hipMemset(dx, 0xff, nc * 8);
    launch_cons_solve(dH, dL, dg, dx, nc, false, dfail, 0);
    hipDeviceSynchronize();
    std::vector<double> x2(nc);
    hipMemcpy(x2.data(), dx, nc * 8, hipMemcpyDeviceToHost);
    double e2 = 0.0;
    for (int r = 0; r < nc; r++) e2 = fmax(e2, fabs(x2[r] - (double)y[r]));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 50;
    for (int k = 0; k < 5; k++) launch_cons_solve(dH, dL, dg, dx, nc, true, dfail, 0);
    hipEventRecord(e0, 0);
    for (int k = 0; k < reps; k++) launch_cons_solve(dH, dL, dg, dx, nc, true, dfail, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    printf("nc %3d: fail %d  |x - x*| %.2e (|x*| %.2e)  |L - L*| %.2e  stored-factor solve %.2e   factor + solve %.1f us per launch%s\n", nc, fail, ex, nx, eL, e2,
           1e3 * ms / reps, cons_reg_fits(nc) ? "" : "  (does not fit the register-resident kernel)");
    hipFree(dH); hipFree(dL); hipFree(dg); hipFree(dx); hipFree(dfail);
  }
  return 0;
}
