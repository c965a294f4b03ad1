// kernels_epi.hip — elementwise passes of the cone objective WITH log-barrier smoothing (solver.hip, lcone_smooth_body).  gfx950 only.
//
// Reference: PMPC.jl/src/main.jl:246-262 (smooth_cstr = "logbarrier": every box side becomes an exponential-cone row triple with a new
// epigraph variable of cost 1, cone_utils.jl:173-203, i.e. the term -(1/alpha) log(alpha slack) in the objective) next to the
// eps-anchored epigraph rows of main.jl:204-238.  The barrier terms are NOT scaled by the multipliers of the epigraph rows, so a
// particle's optimum given the shared controls depends on its multiplier and the reduction to the shared-control space of
// epigraph_host.hip does not apply: the Newton iteration runs in the full space, its linear algebra on the Riccati sweeps of
// kernels_fast.hip (two right-hand sides per particle: the rank-one term of an epigraph row on its threshold is a Sherman-Morrison
// correction).
#include "pmpc_dev.h"

namespace {

// Barrier terms of the boxes at (X, U): diagonal D = mu (1/sl^2 + 1/su^2) and gradient shift w = mu (-1/sl + 1/su) per entry (the
// shared controls of the consensus stages once, on the owner's particle 0 — whose bounds they are, lqp_utils.jl:329-330), block
// partials of the barrier value -mu sum log(slack) and of the smallest slack (<= 0: infeasible point).  One launch for both slabs.
__global__ void __launch_bounds__(256) k_bar_prep(const double *X, const double *U, const double *lx, const double *ux, const double *lu, const double *uu,
                                                  double *Dx, double *wx, double *Du, double *wu, double mu, long long nx, long long nu, int u, int N, int Nc,
                                                  int owner, double *part_val, double *part_min, int mode, double beta) {
  __shared__ double sv[256], sm[256];
  double val = 0.0, smin = 1e300;
  const long long stride = (long long)gridDim.x * 256;
  // mode 1, smooth_cstr = "squareplus" (main.jl:265-279, cone_utils.jl:222-228): every box side a'z <= b costs
  //   tau(v) = beta/2 (v + sqrt(v^2 + 1/alpha^2)),  v = a'z - b  (the second-order-cone rows (2/beta) tau - v >= |(v, 1/alpha)| with
  // cost 1 on tau): a smooth hinge — soft boxes, no interior to stay in.  `mu` carries 1/alpha here.
  auto sq = [&](double v, double sgn, double &D, double &w) {
    const double r = sqrt(v * v + mu * mu);
    val += 0.5 * beta * (v + r);
    w += sgn * 0.5 * beta * (1.0 + v / r);
    D += 0.5 * beta * mu * mu / (r * r * r);
  };
  auto one = [&](double z, double lo, double hi, double &D, double &w) {
    D = 0.0; w = 0.0;
    if (mode == 1) {
      if (lo > -1e300) sq(lo - z, -1.0, D, w);
      if (hi < 1e300) sq(z - hi, 1.0, D, w);
      smin = fmin(smin, 1.0);
      return;
    }
    if (lo > -1e300) {
      const double s = z - lo;
      smin = fmin(smin, s);
      if (s > 0.0) { D += mu / (s * s); w -= mu / s; val -= mu * log(s); }
    }
    if (hi < 1e300) {
      const double s = hi - z;
      smin = fmin(smin, s);
      if (s > 0.0) { D += mu / (s * s); w += mu / s; val -= mu * log(s); }
    }
  };
  if (lx)
    for (long long k = blockIdx.x * 256ll + threadIdx.x; k < nx; k += stride) {
      double D, w;
      one(X[k], lx[k], ux[k], D, w);
      Dx[k] = D; wx[k] = w;
    }
  if (lu)
    for (long long k = blockIdx.x * 256ll + threadIdx.x; k < nu; k += stride) {
      const int j = (int)((k / u) % N);
      const long long i = k / ((long long)N * u);
      double D = 0.0, w = 0.0;
      if (j >= Nc || (i == 0 && owner)) one(U[k], lu[k], uu[k], D, w);
      Du[k] = D; wu[k] = w;
    }
  sv[threadIdx.x] = val; sm[threadIdx.x] = smin;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { sv[threadIdx.x] += sv[threadIdx.x + o]; sm[threadIdx.x] = fmin(sm[threadIdx.x], sm[threadIdx.x + o]); }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part_val[blockIdx.x] = sv[0]; part_min[blockIdx.x] = sm[0]; }
}

// out[0] = sum of part_val, out[1] = min of part_min (one block)
__global__ void __launch_bounds__(256) k_bar_reduce(const double *part_val, const double *part_min, int n, double *out) {
  __shared__ double sv[256], sm[256];
  double v = 0.0, m = 1e300;
  for (int k = threadIdx.x; k < n; k += 256) { v += part_val[k]; m = fmin(m, part_min[k]); }
  sv[threadIdx.x] = v; sm[threadIdx.x] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { sv[threadIdx.x] += sv[threadIdx.x + o]; sm[threadIdx.x] = fmin(sm[threadIdx.x], sm[threadIdx.x + o]); }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[0] = sv[0]; out[1] = sm[0]; }
}

// Per particle: the cost gradient a_i = grad J_i at (X, U) (qp_utils.jl:60-162: Q (x - x_ref) + reg_x (x - x_prev), the same for the
// controls) against two directions: out[3 i + 0] = a_i . d1, out[3 i + 1] = a_i . d2, out[3 i + 2] = d1' (grad^2 J_i) d1 (the curvature of
// the particle's cost along d1).  One 256-thread block per particle.
// bx / bu (null: off): out[3 i + 1] becomes  b_i . d1  instead of  a_i . d2,  b_i = pw_i a_i + (bx; bu)  the right-hand side of the particle's
// Newton system — with d1 = -K^-1 a_i that is  a_i . (-K^-1 b_i)  by symmetry: the step for b_i itself is then not needed (one forward sweep less)
__global__ void __launch_bounds__(256) k_cost_dots(LQArgs a, const double *X, const double *U, const double *dX1, const double *dU1, const double *dX2,
                                                   const double *dU2, double *out, const double *bx, const double *bu) {
  // The particle's vectors go to LDS once (with the regularisation terms taken on the way); then the cost blocks are STREAMED, one
  // element per thread and pass in memory order:  a0 += Q[c, r] (x - x_ref)[c] d1[r],  a1 with d2[r],  a2 += Q[c, r] d1[c] d1[r].
  // (A thread per row read its 12 doubles in a loop — 64 lanes x 96-byte stride per instruction: 186 us for the 260 MB of config D's
  //  blocks, as long as a factor sweep; streamed: the same bytes at the rate of a copy.)
  extern __shared__ double sv[];
  __shared__ double s0[256], s1[256], s2[256];
  const int i = blockIdx.x, tid = threadIdx.x, x = a.x, u = a.u, N = a.N;
  const size_t pb = (size_t)i * N;
  const bool rhs = bx != nullptr || bu != nullptr;
  double wdot = 0.0;
  double *xr = sv, *d1x = xr + N * x, *d2x = d1x + N * x, *ur = d2x + N * x, *d1u = ur + N * u, *d2u = d1u + N * u;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0;
  for (int e = tid; e < N * x; e += 256) {
    const size_t o = pb * x + e;
    const double xv = X[o], v1 = dX1[o], v2 = rhs ? 0.0 : dX2[o], gp = a.reg_x * (xv - a.X_prev[o]);
    xr[e] = xv - a.X_ref[o]; d1x[e] = v1; d2x[e] = v2;
    a0 += gp * v1; a1 += gp * v2; a2 += a.reg_x * v1 * v1;
    if (bx) wdot += bx[o] * v1;
  }
  for (int e = tid; e < N * u; e += 256) {
    const size_t o = pb * u + e;
    const double uv = U[o], v1 = dU1[o], v2 = rhs ? 0.0 : dU2[o], gp = a.reg_u * (uv - a.U_prev[o]);
    ur[e] = uv - a.U_ref[o]; d1u[e] = v1; d2u[e] = v2;
    a0 += gp * v1; a1 += gp * v2; a2 += a.reg_u * v1 * v1;
    if (bu) wdot += bu[o] * v1;
  }
  __syncthreads();
  {
    const int xx = x * x, tot = N * xx;
    const double *Qp = a.Q + pb * xx;
    int e = tid;
    for (; e + 768 < tot; e += 1024) {  // (four loads in flight per thread: one at a time is latency-bound)
      const double qv[4] = {Qp[e], Qp[e + 256], Qp[e + 512], Qp[e + 768]};
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int ee = e + 256 * k, j = ee / xx, rem = ee - j * xx, r = rem / x, c = rem - r * x, b = j * x;
        const double q = qv[k], t1 = d1x[b + r];
        a0 = fma(q * xr[b + c], t1, a0);
        a1 = fma(q * xr[b + c], d2x[b + r], a1);
        a2 = fma(q * d1x[b + c], t1, a2);
      }
    }
    for (; e < tot; e += 256) {
      const int j = e / xx, rem = e - j * xx, r = rem / x, c = rem - r * x, b = j * x;  // (symmetric blocks on this path: row r read as column r)
      const double q = Qp[e], t1 = d1x[b + r];
      a0 = fma(q * xr[b + c], t1, a0);
      a1 = fma(q * xr[b + c], d2x[b + r], a1);
      a2 = fma(q * d1x[b + c], t1, a2);
    }
  }
  {
    const int uu = u * u, tot = N * uu;
    const double *Rp = a.R + pb * uu;
    for (int e = tid; e < tot; e += 256) {
      const int j = e / uu, rem = e - j * uu, r = rem / u, c = rem - r * u, b = j * u;
      const double q = Rp[e], t1 = d1u[b + r];
      a0 = fma(q * ur[b + c], t1, a0);
      a1 = fma(q * ur[b + c], d2u[b + r], a1);
      a2 = fma(q * d1u[b + c], t1, a2);
    }
  }
  s0[tid] = a0; s1[tid] = rhs ? wdot : a1; s2[tid] = a2;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) { s0[tid] += s0[tid + o]; s1[tid] += s1[tid + o]; s2[tid] += s2[tid + o]; }
    __syncthreads();
  }
  if (tid == 0) { out[3 * i] = s0[0]; out[3 * i + 1] = rhs ? fma(a.pw ? a.pw[i] : 1.0, s0[0], s1[0]) : s1[0]; out[3 * i + 2] = s2[0]; }
}

// The (Nc u + 1) system of the shared controls and t of one Newton step of the smoothed cone objective, on the device (one rank, Nc u <= 8):
// per-particle rank-one terms folded by Sherman-Morrison exactly as lcone_smooth_body's host loop does it —
//   A = sum_i [H_i + sp_i ga_i ga_i', -sp_i ga_i; -sp_i ga_i', sp_i],  rhs = [-sum_i (gb_i + sp_i pi_i ga_i); sum_i sp_i pi_i - (K - sum mu)],
//   sp_i = sig_i / (1 + sig_i kap_i),  kap_i = max(0, -dots[3i]),  pi_i = dots[3i + 1]
// — sums in a fixed order (per thread over its particles, wave shuffles, the 16 waves in turn), Gaussian elimination with partial pivoting by
// thread 0, then coef_i = sig_i (pi_i - dt + ga_i . du) / (1 + sig_i kap_i) for every particle and the shared step du.  What it removes
// from a Newton step: a stream synchronisation, the read-back of every (H_i, g_i) and two uploads.
// MODE 0: sums, solve and coefficients in one launch (one rank).  Sharded: MODE 1 leaves this rank's sums in `xch` (E doubles + its failure
// flag), the caller all-reduces them, MODE 2 solves from `xch` and writes the coefficients of this rank's particles — the exchange is E + 1
// doubles per Newton step instead of every particle's condensed blocks.
template <int NC, int MODE>
__global__ void __launch_bounds__(1024) k_epi_newton(const double *Hc_part, const double *gb, const double *ga, const double *dots, const double *sig, int M,
                                                      double k_minus_summu, double *coef, double *duc, int *fail, double *xch) {
  constexpr int NS = NC * (NC + 1) / 2, E = NS + 2 * NC + 2, N1 = NC + 1;
  __shared__ double part[16][E], tot[E], sol[N1];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  double acc[E];
  if (MODE == 2) {
    if (tid < E) tot[tid] = xch[tid];
    if (tid == 0 && xch[E] != 0.0) *fail = 2;  // (a rank's factor sweep failed: every rank gives up alike)
    __syncthreads();
  } else {
#pragma unroll
  for (int e = 0; e < E; e++) acc[e] = 0.0;
  for (int i = tid; i < M; i += 1024) {
    const double kap = fmax(0.0, -dots[3 * i]), pi_ = dots[3 * i + 1], sg = sig[i], sp = sg / (1.0 + sg * kap);
    const double *Hi = Hc_part + (size_t)i * NC * NC, *gai = ga + (size_t)i * NC, *gbi = gb + (size_t)i * NC;
    double g[NC];
#pragma unroll
    for (int r = 0; r < NC; r++) g[r] = gai[r];
    int e = 0;
#pragma unroll
    for (int q = 0; q < NC; q++)
#pragma unroll
      for (int r = 0; r <= q; r++) acc[e++] += Hi[r + NC * q] + sp * g[r] * g[q];  // (upper triangle of H_i)
#pragma unroll
    for (int r = 0; r < NC; r++) {
      acc[NS + r] -= sp * g[r];
      acc[NS + NC + r] -= gbi[r] + sp * pi_ * g[r];
    }
    acc[NS + 2 * NC] += sp;
    acc[NS + 2 * NC + 1] += sp * pi_;
  }
#pragma unroll
  for (int e = 0; e < E; e++) {
    double v = acc[e];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) part[wv][e] = v;
  }
  __syncthreads();
  if (tid < E) {
    double v = 0.0;
    for (int k = 0; k < 16; k++) v += part[k][tid];
    tot[tid] = v;
    if (MODE == 1) xch[tid] = v;
  }
  if (MODE == 1) {
    if (tid == 0) xch[E] = (double)*fail;
    return;
  }
  __syncthreads();
  }
  if (tid == 0) {
    double A[N1][N1], b[N1];
    int e = 0;
    for (int q = 0; q < NC; q++)
      for (int r = 0; r <= q; r++) { A[r][q] = tot[e]; A[q][r] = tot[e]; e++; }
    const double cc = tot[NS + 2 * NC];
    for (int r = 0; r < NC; r++) {
      A[r][NC] = A[NC][r] = cc > 0.0 ? tot[NS + r] : 0.0;  // (no row strictly inside: t stays — it is re-optimised exactly at the next point)
      b[r] = tot[NS + NC + r];
    }
    A[NC][NC] = cc > 0.0 ? cc : 1.0;
    b[NC] = cc > 0.0 ? tot[NS + 2 * NC + 1] - k_minus_summu : 0.0;
    bool bad = false;
    for (int k = 0; k < N1; k++) {  // Gaussian elimination with partial pivoting (the matrix is positive definite)
      int pv = k;
      for (int r = k + 1; r < N1; r++)
        if (fabs(A[r][k]) > fabs(A[pv][k])) pv = r;
      if (pv != k) {
        for (int q = 0; q < N1; q++) { const double t_ = A[k][q]; A[k][q] = A[pv][q]; A[pv][q] = t_; }
        const double t_ = b[k]; b[k] = b[pv]; b[pv] = t_;
      }
      const double d = A[k][k];
      if (!(fabs(d) > 0.0)) { bad = true; break; }
      for (int r = k + 1; r < N1; r++) {
        const double f_ = A[r][k] / d;
        for (int q = k; q < N1; q++) A[r][q] -= f_ * A[k][q];
        b[r] -= f_ * b[k];
      }
    }
    if (!bad)
      for (int k = N1 - 1; k >= 0; k--) {
        double v = b[k];
        for (int q = k + 1; q < N1; q++) v -= A[k][q] * sol[q];
        sol[k] = v / A[k][k];
      }
    if (bad) {
      *fail = 2;
      for (int k = 0; k < N1; k++) sol[k] = 0.0;
    }
    for (int r = 0; r < NC; r++) duc[r] = sol[r];
  }
  __syncthreads();
  const double dt = sol[NC];
  for (int i = tid; i < M; i += 1024) {
    const double kap = fmax(0.0, -dots[3 * i]), pi_ = dots[3 * i + 1], sg = sig[i];
    double e_ = pi_ - dt;
#pragma unroll
    for (int r = 0; r < NC; r++) e_ += ga[(size_t)i * NC + r] * sol[r];
    coef[i] = sg * e_ / (1.0 + sg * kap);
  }
}

// y[i, :] = a[i, :] + coef[i] * b[i, :]   (per-particle scalar; `per` entries per particle)
__global__ void __launch_bounds__(256) k_axpy_particle(const double *a_, const double *b_, const double *coef, double *y, long long per, long long tot) {
  for (long long k = blockIdx.x * 256ll + threadIdx.x; k < tot; k += (long long)gridDim.x * 256) y[k] = fma(coef[k / per], b_[k], a_[k]);
}

// y = a + alpha * b
__global__ void __launch_bounds__(256) k_step_to(const double *a_, const double *b_, double alpha, double *y, long long tot) {
  for (long long k = blockIdx.x * 256ll + threadIdx.x; k < tot; k += (long long)gridDim.x * 256) y[k] = fma(alpha, b_[k], a_[k]);
}

// U strictly inside its box: midpoint of a finite box, the caller's value pulled inside a half-infinite one (cold start of the barrier Newton)
__global__ void __launch_bounds__(256) k_interior(double *U, const double *lo, const double *hi, long long tot, double frac) {
  for (long long k = blockIdx.x * 256ll + threadIdx.x; k < tot; k += (long long)gridDim.x * 256) {
    const double l = lo[k], h = hi[k];
    double v = U[k];
    if (l > -1e300 && h < 1e300) {
      const double m = frac * (h - l);
      v = fmin(fmax(v, l + m), h - m);
    } else if (l > -1e300) v = fmax(v, l + frac * fmax(1.0, fabs(l)));
    else if (h < 1e300) v = fmin(v, h - frac * fmax(1.0, fabs(h)));
    U[k] = v;
  }
}

// consensus stages: every particle carries particle 0's controls (one shared value)
__global__ void __launch_bounds__(256) k_share_cons(double *U, int M, int N, int u, int Nc) {
  const long long tot = (long long)M * Nc * u;
  for (long long k = blockIdx.x * 256ll + threadIdx.x; k < tot; k += (long long)gridDim.x * 256) {
    const long long i = k / ((long long)Nc * u), r = k % ((long long)Nc * u);
    if (i > 0) U[i * (long long)N * u + r] = U[r];
  }
}

inline unsigned grid_for(long long tot, unsigned cap = 2048) {
  const long long b = (tot + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

void launch_bar_prep(const double *X, const double *U, const double *lx, const double *ux, const double *lu, const double *uu, double *Dx, double *wx,
                     double *Du, double *wu, double mu, long long nx, long long nu, int u, int N, int Nc, int owner, double *part_val, double *part_min,
                     double *out2, hipStream_t s, int mode, double beta) {
  const unsigned G = grid_for(nx > nu ? nx : nu, PMPC_RED_BLOCKS);
  hipLaunchKernelGGL(k_bar_prep, dim3(G), dim3(256), 0, s, X, U, lx, ux, lu, uu, Dx, wx, Du, wu, mu, nx, nu, u, N, Nc, owner, part_val, part_min, mode, beta);
  hipLaunchKernelGGL(k_bar_reduce, dim3(1), dim3(256), 0, s, (const double *)part_val, (const double *)part_min, (int)G, out2);
}
void launch_cost_dots(const LQArgs &a, const double *X, const double *U, const double *dX1, const double *dU1, const double *dX2, const double *dU2,
                      double *out, hipStream_t s, const double *bx, const double *bu) {
  const size_t lds = (size_t)3 * a.N * (a.x + a.u) * sizeof(double);  // (N (x + u) <= 2 300 inside the 64 KB next to the reduction arrays: N = 100 at x12 u4 is 1 600)
  hipLaunchKernelGGL(k_cost_dots, dim3(a.M), dim3(256), lds, s, a, X, U, dX1, dU1, dX2, dU2, out, bx, bu);
}
// mode 0: everything (one rank); 1: this rank's sums -> xch (returns their count, fail flag included, for the all-reduce); 2: solve from xch
int launch_epi_newton(const double *Hc_part, const double *gb, const double *ga, const double *dots, const double *sig, int M, int nc, double k_minus_summu,
                      double *coef, double *duc, int *fail, hipStream_t s, int mode, double *xch) {
#define X(NC)                                                                                                                                            \
  case NC:                                                                                                                                               \
    if (mode == 0) hipLaunchKernelGGL((k_epi_newton<NC, 0>), dim3(1), dim3(1024), 0, s, Hc_part, gb, ga, dots, sig, M, k_minus_summu, coef, duc, fail, xch);       \
    else if (mode == 1) hipLaunchKernelGGL((k_epi_newton<NC, 1>), dim3(1), dim3(1024), 0, s, Hc_part, gb, ga, dots, sig, M, k_minus_summu, coef, duc, fail, xch);  \
    else hipLaunchKernelGGL((k_epi_newton<NC, 2>), dim3(1), dim3(1024), 0, s, Hc_part, gb, ga, dots, sig, M, k_minus_summu, coef, duc, fail, xch);                 \
    return NC * (NC + 1) / 2 + 2 * NC + 2 + 1;
  switch (nc) { X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) default: return 0; }
#undef X
}
void launch_axpy_particle(const double *a_, const double *b_, const double *coef, double *y, long long per, long long tot, hipStream_t s) {
  hipLaunchKernelGGL(k_axpy_particle, dim3(grid_for(tot)), dim3(256), 0, s, a_, b_, coef, y, per, tot);
}
void launch_step_to(const double *a_, const double *b_, double alpha, double *y, long long tot, hipStream_t s) {
  hipLaunchKernelGGL(k_step_to, dim3(grid_for(tot)), dim3(256), 0, s, a_, b_, alpha, y, tot);
}
void launch_interior(double *U, const double *lo, const double *hi, long long tot, double frac, hipStream_t s) {
  hipLaunchKernelGGL(k_interior, dim3(grid_for(tot)), dim3(256), 0, s, U, lo, hi, tot, frac);
}
void launch_share_cons(double *U, int M, int N, int u, int Nc, hipStream_t s) {
  if (Nc <= 0) return;
  hipLaunchKernelGGL(k_share_cons, dim3(grid_for((long long)M * Nc * u)), dim3(256), 0, s, U, M, N, u, Nc);
}
