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
