// dynamics.hip — on-device linearisation: the f_fx_fu_fn step of the SCP loop
// (pmpc/scp_mpc.py:338-342) for the built-in models, writing f, fx, fu straight into the ABI
// layout the Riccati kernels stream (no host round trip of the (M,N,x,x) Jacobian stacks).
//
//   X_ = [x0, X_prev[:-1]]  (scp_mpc.py:338);  f[j] = F(X_[j], U_prev[j]), fx[j] = dF/dx, fu[j] = dF/du
//
// model 0: unicycle of the reference's tests/dubins_car.py:48-90 (closed-form Jacobians; the
//          reference uses torch.autograd, :11-30).  params (3,M) = [v_scale, w_scale, T].
// model 1: synthetic quadrotor of SURVEY.md §8(d) (not in the reference). params (4,M) = [m,Jx,Jy,Jz].
// Host (numpy) specifications: pmpc_amd/dynamics.py.  One thread evaluates one (particle, stage) into an LDS record
// [f | fx | fu]; the workgroup then streams its records out: consecutive (particle, stage) units are consecutive
// in all three stacks, so every store instruction writes 64 consecutive doubles (8-byte stores scattered inside each
// thread's own 1152-byte block reached 1.9 TB/s; the staged version is write-bandwidth bound).
#include "pmpc_dev.h"
#include <cstring>
#include <cstdlib>

namespace {

struct Unicycle {
  static constexpr int X = 4, U = 2, UNITS = 64;
  static __device__ __forceinline__ void eval(long long idx, int N, const double *x0, const double *X_prev, const double *U_prev,
                                              const double *params, double *fo, double *A, double *B);
};
struct Quadrotor {
  static constexpr int X = 12, U = 4, UNITS = 32;  // 32 records of 204 doubles = 52 KB of LDS
  static __device__ __forceinline__ void eval(long long idx, int N, const double *x0, const double *X_prev, const double *U_prev,
                                              const double *params, double *fo, double *A, double *B);
};

__device__ __forceinline__ void Unicycle::eval(long long idx, int N, const double *x0, const double *X_prev, const double *U_prev,
                                               const double *params, double *fo, double *A, double *B) {
  const int i = (int)(idx / N), j = (int)(idx % N);
  const double *xs = j == 0 ? x0 + 4 * (size_t)i : X_prev + (idx - 1) * 4;
  const double *us = U_prev + idx * 2, *p = params + 3 * (size_t)i;
  const double vs = p[0], ws = p[1], T = p[2], eps = 1e-6;
  double u1 = vs * us[0], u2 = -ws * us[1];
  u1 += (u1 >= 0.0 ? eps : -eps);
  u2 += (u2 >= 0.0 ? eps : -eps);
  const double px = xs[0], py = xs[1], v0 = xs[2], th0 = xs[3];
  const double a = T * u2 + th0;
  double sa, ca, s0, c0;
  sincos(a, &sa, &ca);
  sincos(th0, &s0, &c0);
  const double iu2 = 1.0 / u2, iu22 = iu2 * iu2;
  const double n1 = u2 * sa * v0 + T * u1 * u2 * sa + u1 * ca - s0 * u2 * v0 - c0 * u1;
  const double n2 = -(u2 * ca * v0 - u1 * sa + T * u1 * u2 * ca) + c0 * u2 * v0 - s0 * u1;
  fo[0] = px + n1 * iu22; fo[1] = py + n2 * iu22; fo[2] = v0 + T * u1; fo[3] = a;
  // (A and B arrive zeroed: the whole block clears the record buffer before the model runs, k_linearize)
  // column-major blocks: A[r + 4*t] = dF_r/dx_t
  A[0 + 4 * 0] = 1.0; A[1 + 4 * 1] = 1.0; A[2 + 4 * 2] = 1.0; A[3 + 4 * 3] = 1.0;
  A[0 + 4 * 2] = (u2 * sa - s0 * u2) * iu22;
  A[0 + 4 * 3] = (u2 * ca * v0 + T * u1 * u2 * ca - u1 * sa - c0 * u2 * v0 + s0 * u1) * iu22;
  A[1 + 4 * 2] = (-u2 * ca + c0 * u2) * iu22;
  A[1 + 4 * 3] = (-(-u2 * sa * v0 - u1 * ca - T * u1 * u2 * sa) - s0 * u2 * v0 - c0 * u1) * iu22;
  const double dn1_du1 = T * u2 * sa + ca - c0, dn2_du1 = sa - T * u2 * ca - s0;
  const double dn1_du2 = sa * v0 + u2 * ca * T * v0 + T * u1 * sa + T * u1 * u2 * ca * T - u1 * sa * T - s0 * v0;
  const double dn2_du2 = -(ca * v0 - u2 * sa * T * v0 - u1 * ca * T + T * u1 * ca - T * T * u1 * u2 * sa) + c0 * v0;
  B[0 + 4 * 0] = dn1_du1 * iu22 * vs;
  B[1 + 4 * 0] = dn2_du1 * iu22 * vs;
  B[2 + 4 * 0] = T * vs;
  B[0 + 4 * 1] = (dn1_du2 * iu22 - 2.0 * n1 * iu22 * iu2) * (-ws);
  B[1 + 4 * 1] = (dn2_du2 * iu22 - 2.0 * n2 * iu22 * iu2) * (-ws);
  B[3 + 4 * 1] = T * (-ws);
}

__device__ __forceinline__ void Quadrotor::eval(long long idx, int N, const double *x0, const double *X_prev, const double *U_prev,
                                                const double *params, double *fo, double *A, double *B) {
  const int i = (int)(idx / N), j = (int)(idx % N);
  const double *xs = j == 0 ? x0 + 12 * (size_t)i : X_prev + (idx - 1) * 12;
  const double *us = U_prev + idx * 4, *p = params + 4 * (size_t)i;
  const double m = p[0], Jx = p[1], Jy = p[2], Jz = p[3], dt = 0.05, g = 9.81;
  const double ph = xs[6], th = xs[7], ps = xs[8], wx = xs[9], wy = xs[10], wz = xs[11];
  const double T = us[0], tx = us[1], ty = us[2], tz = us[3];
  double sph, cph, sth, cth, sps, cps;
  sincos(ph, &sph, &cph);
  sincos(th, &sth, &cth);
  sincos(ps, &sps, &cps);
  const double icth = 1.0 / cth, tth = sth * icth, sec2 = icth * icth;
  const double bx = cps * sth * cph + sps * sph, by = sps * sth * cph - cps * sph, bz = cth * cph;
  const double a = T / m;
  fo[0] = xs[0] + dt * xs[3]; fo[1] = xs[1] + dt * xs[4]; fo[2] = xs[2] + dt * xs[5];
  fo[3] = xs[3] + dt * a * bx; fo[4] = xs[4] + dt * a * by; fo[5] = xs[5] + dt * (a * bz - g);
  fo[6] = ph + dt * (wx + sph * tth * wy + cph * tth * wz);
  fo[7] = th + dt * (cph * wy - sph * wz);
  fo[8] = ps + dt * (sph * icth * wy + cph * icth * wz);
  fo[9] = wx + dt * (tx - (Jz - Jy) * wy * wz) / Jx;
  fo[10] = wy + dt * (ty - (Jx - Jz) * wz * wx) / Jy;
  fo[11] = wz + dt * (tz - (Jy - Jx) * wx * wy) / Jz;
  // (A and B arrive zeroed: the whole block clears the record buffer before the model runs, k_linearize)
#define AE(r, t) A[(r) + 12 * (t)]
#define BE(r, t) B[(r) + 12 * (t)]
  for (int k = 0; k < 12; k++) AE(k, k) = 1.0;
  for (int k = 0; k < 3; k++) AE(k, 3 + k) = dt;
  AE(3, 6) = dt * a * (-cps * sth * sph + sps * cph); AE(3, 7) = dt * a * (cps * cth * cph); AE(3, 8) = dt * a * (-sps * sth * cph + cps * sph);
  AE(4, 6) = dt * a * (-sps * sth * sph - cps * cph); AE(4, 7) = dt * a * (sps * cth * cph); AE(4, 8) = dt * a * (cps * sth * cph + sps * sph);
  AE(5, 6) = dt * a * (-cth * sph); AE(5, 7) = dt * a * (-sth * cph);
  AE(6, 6) += dt * (cph * tth * wy - sph * tth * wz);
  AE(6, 7) = dt * (sph * sec2 * wy + cph * sec2 * wz);
  AE(6, 9) = dt; AE(6, 10) = dt * sph * tth; AE(6, 11) = dt * cph * tth;
  AE(7, 6) = dt * (-sph * wy - cph * wz);
  AE(7, 10) = dt * cph; AE(7, 11) = -dt * sph;
  AE(8, 6) = dt * (cph * icth * wy - sph * icth * wz);
  AE(8, 7) = dt * (sph * wy + cph * wz) * sth * sec2;
  AE(8, 10) = dt * sph * icth; AE(8, 11) = dt * cph * icth;
  AE(9, 10) = -dt * (Jz - Jy) * wz / Jx; AE(9, 11) = -dt * (Jz - Jy) * wy / Jx;
  AE(10, 9) = -dt * (Jx - Jz) * wz / Jy; AE(10, 11) = -dt * (Jx - Jz) * wx / Jy;
  AE(11, 9) = -dt * (Jy - Jx) * wy / Jz; AE(11, 10) = -dt * (Jy - Jx) * wx / Jz;
  BE(3, 0) = dt * bx / m; BE(4, 0) = dt * by / m; BE(5, 0) = dt * bz / m;
  BE(9, 1) = dt / Jx; BE(10, 2) = dt / Jy; BE(11, 3) = dt / Jz;
#undef AE
#undef BE
}

// one 256-thread block of the SCP residual (k_scp_residual's work; also run by extra blocks of k_linearize)
__device__ __forceinline__ void residual_block(int bid, int nblk, const double *X, const double *Xp, long long rows_x, int x, const double *U,
                                               const double *Up, long long rows_u, int u, unsigned long long *out_bits) {
  __shared__ double sh[256];
  const int sub = threadIdx.x & 3;
  const long long grp = (bid * 256LL + threadIdx.x) >> 2, ngrp = ((long long)nblk * 256) >> 2;
  const int qx = (x + 3) / 4, qu = (u + 3) / 4;
  double m = 0.0;
  for (long long r = grp; r < rows_x + rows_u; r += ngrp) {  // (the 4 lanes of a group run the same trip count)
    const bool isx = r < rows_x;
    const long long rr = isx ? r : r - rows_x;
    const int d = isx ? x : u, q = isx ? qx : qu;
    const double *a = (isx ? X : U) + rr * d, *b = (isx ? Xp : Up) + rr * d;
    const int k1 = min(d, (sub + 1) * q);
    double acc = 0.0;
    for (int k = sub * q; k < k1; k++) { const double t = a[k] - b[k]; acc = fma(t, t, acc); }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    m = (acc == acc) ? fmax(m, acc) : INFINITY;  // a NaN trajectory must not look converged
  }
  sh[threadIdx.x] = sqrt(m);
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sh[threadIdx.x] = fmax(sh[threadIdx.x], sh[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicMax(out_bits, (unsigned long long)__double_as_longlong(sh[0]));
}
struct ResArgs {  // residual of the iteration just finished, riding in the launch that linearises the next one (nblk = 0: none)
  const double *X, *Xp, *U, *Up;
  long long rows;
  int x, u, nblk;
  unsigned long long *out_bits;
};
#ifndef PMPC_LIN_THREADS
#define PMPC_LIN_THREADS 256  // the model is evaluated by the first UNITS threads; all of them stream the records out
#endif
// JT: storage type of the Jacobian stacks (float = the fp32-storage mode of the active-set sweeps)
template <class Model, class JT>
__global__ void __launch_bounds__(PMPC_LIN_THREADS) k_linearize(int N, long long tot, const double *x0, const double *X_prev, const double *U_prev,
                                                  const double *params, double *f, JT *fx, JT *fu, ResArgs res) {
  constexpr int X = Model::X, U = Model::U, UNITS = Model::UNITS;
  {
    const int glin = (int)gridDim.x - res.nblk;  // the last res.nblk blocks compute the residual (independent work, one launch)
    if ((int)blockIdx.x >= glin) {
      residual_block((int)blockIdx.x - glin, res.nblk, res.X, res.Xp, res.rows, res.x, res.U, res.Up, res.rows, res.u, res.out_bits);
      return;
    }
  }
  constexpr int REC = X + X * X + X * U, LD = REC | 1;  // odd record stride: conflict-free LDS stores
  extern __shared__ double rec[];
  const int t = threadIdx.x;
  const long long first = (long long)blockIdx.x * UNITS;
  // the Jacobians are mostly zeros: cleared by all threads at once instead of ~200 serial LDS stores in every model thread
  for (int e = t; e < UNITS * LD; e += PMPC_LIN_THREADS) rec[e] = 0.0;
  __syncthreads();
  if (t < UNITS && first + t < tot) {
    double *mine = rec + t * LD;
    Model::eval(first + t, N, x0, X_prev, U_prev, params, mine, mine + X, mine + X + X * X);
  }
  __syncthreads();
  const int n = (int)((tot - first) < UNITS ? (tot - first) : UNITS);
  for (int e = t; e < n * X; e += PMPC_LIN_THREADS) f[first * X + e] = rec[(e / X) * LD + e % X];
  for (int e = t; e < n * X * X; e += PMPC_LIN_THREADS) fx[first * (X * X) + e] = (JT)rec[(e / (X * X)) * LD + X + e % (X * X)];
  for (int e = t; e < n * X * U; e += PMPC_LIN_THREADS) fu[first * (X * U) + e] = (JT)rec[(e / (X * U)) * LD + X + X * X + e % (X * U)];
}

template <class Model>
void launch_model(int N, int M, const double *x0, const double *X_prev, const double *U_prev, const double *params, double *f,
                  double *fx, double *fu, const ResArgs &res, hipStream_t s, int jac32) {
  const long long tot = (long long)M * N;
  constexpr int REC = Model::X + Model::X * Model::X + Model::X * Model::U, LD = REC | 1;
  const unsigned grid = (unsigned)((tot + Model::UNITS - 1) / Model::UNITS);
  if (jac32)
    hipLaunchKernelGGL((k_linearize<Model, float>), dim3(grid + (unsigned)res.nblk), dim3(PMPC_LIN_THREADS), Model::UNITS * LD * sizeof(double), s, N, tot, x0,
                       X_prev, U_prev, params, f, (float *)fx, (float *)fu, res);
  else
    hipLaunchKernelGGL((k_linearize<Model, double>), dim3(grid + (unsigned)res.nblk), dim3(PMPC_LIN_THREADS), Model::UNITS * LD * sizeof(double), s, N, tot, x0,
                       X_prev, U_prev, params, f, fx, fu, res);
}
__global__ void __launch_bounds__(256) k_widen_f32(const float *src, double *dst, long long n) {
  for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < n; k += (long long)gridDim.x * 256) dst[k] = (double)src[k];
}

}  // namespace

static long long residual_blocks(long long rows) {
  long long nb = (2 * rows * 4 + 255) / 256;
  // one atomic max per block on ONE address: few blocks for small inputs (13 instead of 18 us at 512 particles x 50 stages),
  // enough of them to cover the memory latency for large ones
  const long long cap = rows >= 200000 ? 1024 : 256;
  return nb > cap ? cap : nb;
}

void launch_linearize(int model, int N, int M, const double *x0, const double *X_prev, const double *U_prev,
                      const double *params, double *f, double *fx, double *fu, hipStream_t s, int jac32) {
  ResArgs none;
  memset(&none, 0, sizeof(none));
  if (model == 0) launch_model<Unicycle>(N, M, x0, X_prev, U_prev, params, f, fx, fu, none, s, jac32);
  else launch_model<Quadrotor>(N, M, x0, X_prev, U_prev, params, f, fx, fu, none, s, jac32);
}
void launch_widen_f32(const float *src, double *dst, long long n, hipStream_t s) {
  long long b = (n + 255) / 256;
  if (b > 4096) b = 4096;
  hipLaunchKernelGGL(k_widen_f32, dim3((unsigned)b), dim3(256), 0, s, src, dst, n);
}

// the same with the SCP residual of (Xr, Xrp, Ur, Urp) computed by extra blocks of the launch (the SCP loop's follow-up of an
// iteration: residual of the iteration + linearisation of the next — independent work); *res_out must be 0 on entry
void launch_linearize_with_residual(int model, int N, int M, const double *x0, const double *X_prev, const double *U_prev,
                                    const double *params, double *f, double *fx, double *fu, const double *Xr, const double *Xrp,
                                    const double *Ur, const double *Urp, int x, int u, double *res_out, hipStream_t s, int jac32) {
  ResArgs r;
  r.X = Xr; r.Xp = Xrp; r.U = Ur; r.Up = Urp; r.rows = (long long)M * N; r.x = x; r.u = u;
  r.nblk = (int)residual_blocks(r.rows);
  r.out_bits = (unsigned long long *)res_out;
  if (model == 0) launch_model<Unicycle>(N, M, x0, X_prev, U_prev, params, f, fx, fu, r, s, jac32);
  else launch_model<Quadrotor>(N, M, x0, X_prev, U_prev, params, f, fx, fu, r, s, jac32);
}

// SCP residual of pmpc/scp_mpc.py:397-403: max over (particle, stage) of the 2-norms of X - X_prev and U - U_prev, in one
// pass.  4 lanes share a row (each a contiguous quarter of it), 2-step quad sum; the maximum is taken over the SQUARED norms
// (one square root per thread at the end, not per row), block max, then an atomic max on the bit pattern of the result.
namespace {
__global__ void __launch_bounds__(256) k_scp_residual(const double *X, const double *Xp, long long rows_x, int x, const double *U,
                                                      const double *Up, long long rows_u, int u, unsigned long long *out_bits) {
  residual_block((int)blockIdx.x, (int)gridDim.x, X, Xp, rows_x, x, U, Up, rows_u, u, out_bits);
}
}  // namespace

void launch_scp_residual(const double *X, const double *Xp, const double *U, const double *Up, long long rows, int x, int u,
                         double *out, hipStream_t s, bool zero_out) {
  if (zero_out) HIP_CHECK(hipMemsetAsync(out, 0, sizeof(double), s));  // (else: the caller vouches that *out is 0)
  const long long nb = residual_blocks(rows);
  hipLaunchKernelGGL(k_scp_residual, dim3((unsigned)nb), dim3(256), 0, s, X, Xp, rows, x, U, Up, rows, u, (unsigned long long *)out);
}
