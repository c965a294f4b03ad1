// kernels_slew.hip — slew penalties without cross terms: the control-increment form of the problem.
//
// The reference adds the slew penalties  1/2 s sum_{j>=1} |u_j - u_{j-1}|^2 + 1/2 s0 |u_0|^2 - s0 u_0'u_{-1}  to the control
// block of the joint QP as a block tridiagonal (PMPC.jl/src/lqp_utils.jl:17-102, linear term :165).  Stage-wise this couples
// u_j to u_{j-1}; the generic kernels carry the previous control in the stage state and keep the cross block.  The
// register-resident MFMA sweeps have no cross block, so for them the problem is restated in the increments
//
//     w_j = u_j - u_{j-1}  (u_{-1} := 0),   z_j = [x_j ; u_j]:
//
//   dynamics   z_j = [f_j ; Up_j] + [fx_j fu_j ; 0 I] (z_{j-1} - Zp_{j-1}) + [fu_j ; I] (w_j - Wp_j),  Wp_j = Up_j - Up_{j-1}
//   stage cost 1/2 (z - zr)' blkdiag(Q_j + dx I, R_j + du I) (z - zr) + 1/2 reg |z - Zp|^2,  reg = min(reg_x, reg_u),
//              dx = reg_x - reg, du = reg_u - reg                                                      (state block)
//              1/2 s |w_j|^2  (j >= 1),   1/2 s0 |w_0 - u_{-1}|^2  (j = 0)                               (control block)
//   boxes      the control boxes become boxes on the lower part of z_j (consensus stages: particle 0's, lqp_utils.jl:329-330)
//
// which is a problem of the plain kind with (xdim + udim, udim) and no slew: same minimiser, same consensus structure
// (u_j shared for j < Nc  <=>  w_j shared for j < Nc).  zr is the reference point of the merged quadratics
// 1/2 (y - yr)'B(y - yr) + 1/2 d |y - yp|^2 (the common regulariser reg |z - Zp|^2 supplies the rest).
// Not valid for N = 1 (there the reference's diagonal rule adds s for a successor that does not exist, lqp_utils.jl:31-39).
#include "pmpc_dev.h"

namespace {

struct SlewAugArgs {
  int x, u, N, M, Nc, has_xb, has_ub, has_um1;
  double dx, du;  // excess of reg_x / reg_u over the regulariser the restated problem is solved with (one of them is 0)
  const double *f, *fx, *fu, *Xp, *Up, *Q, *R, *Xr, *Ur, *lx, *ux, *lu, *uu, *cons_lo, *cons_hi, *slew, *slew0, *um1;
  double *af, *afx, *afu, *aXp, *aUp, *aQ, *aR, *aXr, *aUr, *alo, *ahi;
};

// m <- (B + d I)^-1 (B r + d p) for a column-major dim x dim block B (dim <= 12), d >= 0; d == 0: m = r
__device__ void merged_reference(const double *B, int dim, double d, const double *r, const double *p, double *m) {
  if (d == 0.0) {
    for (int k = 0; k < dim; k++) m[k] = r[k];
    return;
  }
  double A[12 * 12], b[12];
  for (int k = 0; k < dim; k++) {
    double acc = d * p[k];
    for (int c = 0; c < dim; c++) {
      A[k * 12 + c] = B[k + dim * c] + (k == c ? d : 0.0);
      acc += B[k + dim * c] * r[c];
    }
    b[k] = acc;
  }
  for (int k = 0; k < dim; k++) {  // Gaussian elimination with partial pivoting
    int pr = k;
    for (int q = k + 1; q < dim; q++)
      if (fabs(A[q * 12 + k]) > fabs(A[pr * 12 + k])) pr = q;
    if (pr != k) {
      for (int c = 0; c < dim; c++) { const double tmp = A[k * 12 + c]; A[k * 12 + c] = A[pr * 12 + c]; A[pr * 12 + c] = tmp; }
      const double tmp = b[k]; b[k] = b[pr]; b[pr] = tmp;
    }
    const double piv = A[k * 12 + k];
    for (int q = k + 1; q < dim; q++) {
      const double f = A[q * 12 + k] / piv;
      for (int c = k; c < dim; c++) A[q * 12 + c] -= f * A[k * 12 + c];
      b[q] -= f * b[k];
    }
  }
  for (int k = dim - 1; k >= 0; k--) {
    double acc = b[k];
    for (int c = k + 1; c < dim; c++) acc -= A[k * 12 + c] * m[c];
    m[k] = acc / A[k * 12 + k];
  }
}

// one thread per (particle, stage)
__global__ void __launch_bounds__(128) k_slew_augment(SlewAugArgs a) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)a.M * a.N) return;
  const int i = (int)(t / a.N), j = (int)(t % a.N);
  const int x = a.x, u = a.u, n = x + u;
  const double *f = a.f + t * x, *fx = a.fx + t * x * x, *fu = a.fu + t * x * u;
  const double *Xp = a.Xp + t * x, *Up = a.Up + t * u, *Q = a.Q + t * x * x, *R = a.R + t * u * u;
  const double *Xr = a.Xr + t * x, *Ur = a.Ur + t * u;
  double *af = a.af + t * n, *afx = a.afx + t * n * n, *afu = a.afu + t * n * u, *aXp = a.aXp + t * n, *aUp = a.aUp + t * u;
  double *aQ = a.aQ + t * n * n, *aR = a.aR + t * u * u, *aXr = a.aXr + t * n, *aUr = a.aUr + t * u;

  for (int r = 0; r < x; r++) { af[r] = f[r]; aXp[r] = Xp[r]; }
  for (int r = 0; r < u; r++) {
    af[x + r] = Up[r];
    aXp[x + r] = Up[r];
    aUp[r] = Up[r] - (j > 0 ? Up[r - u] : 0.0);
  }
  // column-major blocks: [fx fu ; 0 I] (n x n), [fu ; I] (n x u), blkdiag(Q + dx I, R + du I) (n x n)
  for (int c = 0; c < n; c++)
    for (int r = 0; r < n; r++) {
      double v = 0.0, q = 0.0;
      if (r < x) v = c < x ? fx[r + x * c] : fu[r + x * (c - x)];
      else if (c >= x) v = (r == c) ? 1.0 : 0.0;
      if (r < x && c < x) q = Q[r + x * c] + ((r == c) ? a.dx : 0.0);
      else if (r >= x && c >= x) q = R[(r - x) + u * (c - x)] + ((r == c) ? a.du : 0.0);
      afx[r + n * c] = v;
      aQ[r + n * c] = q;
    }
  for (int c = 0; c < u; c++)
    for (int r = 0; r < n; r++) afu[r + n * c] = r < x ? fu[r + x * c] : ((r - x == c) ? 1.0 : 0.0);
  // control block: s I (j >= 1) / s0 I with reference u_{-1} (j = 0; the linear term exists only with consensus stages, :165)
  const double sd = j == 0 ? a.slew0[i] : a.slew[i];
  for (int c = 0; c < u; c++)
    for (int r = 0; r < u; r++) aR[r + u * c] = (r == c) ? sd : 0.0;
  for (int r = 0; r < u; r++) aUr[r] = (j == 0 && a.has_um1) ? a.um1[(size_t)i * u + r] : 0.0;

  // reference points: the MFMA sweeps know ONE regulariser for the whole state; it is min(reg_x, reg_u), and the part of z
  // whose own regulariser is larger takes the excess d into its cost block: 1/2 (y - r)'B(y - r) + 1/2 d |y - p|^2 =
  // 1/2 (y - m)'(B + d I)(y - m) + const with (B + d I) m = B r + d p  (B + d I is positive definite for d > 0)
  merged_reference(Q, x, a.dx, Xr, Xp, aXr);
  merged_reference(R, u, a.du, Ur, Up, aXr + x);

  // boxes on z: state part as given (or free), control part from the control boxes
  if (a.alo) {
    double *alo = a.alo + t * n, *ahi = a.ahi + t * n;
    const double inf = __builtin_huge_val();
    for (int r = 0; r < x; r++) {
      alo[r] = a.has_xb ? a.lx[t * x + r] : -inf;
      ahi[r] = a.has_xb ? a.ux[t * x + r] : inf;
    }
    for (int r = 0; r < u; r++) {
      double lo = -inf, hi = inf;
      if (a.has_ub) {
        if (j < a.Nc) { lo = a.cons_lo[j * u + r]; hi = a.cons_hi[j * u + r]; }
        else { lo = a.lu[t * u + r]; hi = a.uu[t * u + r]; }
      }
      alo[x + r] = lo;
      ahi[x + r] = hi;
    }
  }
}

// (X, U) from the restated problem's state.  Consensus stages: the lower part of z differs between particles by rounding
// (each particle's forward sweep adds its own U_prev terms), but the reference's consensus controls are ONE variable — they are
// rebuilt as the running sum of the shared increments, the same operations in the same order for every particle of every rank.
__global__ void __launch_bounds__(256) k_slew_split(const double *Z, const double *W, double *X, double *U, long long rows, int x, int u,
                                                    int N, int Nc, const double *cons_lo, const double *cons_hi) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int n = x + u;
  if (t >= rows * n) return;
  const long long row = t / n;
  const int r = (int)(t % n), j = (int)(row % N);
  if (r < x) { X[row * x + r] = Z[t]; return; }
  double v = Z[t];
  if (j < Nc) {
    const double *w0 = W + (row - j) * u + (r - x);
    v = 0.0;
    for (int k = 0; k <= j; k++) v += w0[(long long)k * u];
    if (cons_lo) v = fmin(fmax(v, cons_lo[j * u + (r - x)]), cons_hi[j * u + (r - x)]);
  }
  U[row * u + (r - x)] = v;
}

}  // namespace

void launch_slew_augment(const SlewAug &g, hipStream_t s) {
  SlewAugArgs a;
  a.x = g.x; a.u = g.u; a.N = g.N; a.M = g.M; a.Nc = g.Nc; a.has_xb = g.has_xb; a.has_ub = g.has_ub; a.has_um1 = g.has_um1;
  a.dx = g.dx; a.du = g.du;
  a.f = g.f; a.fx = g.fx; a.fu = g.fu; a.Xp = g.Xp; a.Up = g.Up; a.Q = g.Q; a.R = g.R; a.Xr = g.Xr; a.Ur = g.Ur;
  a.lx = g.lx; a.ux = g.ux; a.lu = g.lu; a.uu = g.uu; a.cons_lo = g.cons_lo; a.cons_hi = g.cons_hi;
  a.slew = g.slew; a.slew0 = g.slew0; a.um1 = g.um1;
  a.af = g.af; a.afx = g.afx; a.afu = g.afu; a.aXp = g.aXp; a.aUp = g.aUp; a.aQ = g.aQ; a.aR = g.aR; a.aXr = g.aXr; a.aUr = g.aUr;
  a.alo = g.alo; a.ahi = g.ahi;
  const long long rows = (long long)g.M * g.N;
  hipLaunchKernelGGL(k_slew_augment, dim3((unsigned)((rows + 127) / 128)), dim3(128), 0, s, a);
}

void launch_slew_split(const double *Z, const double *W, double *X, double *U, long long rows, int x, int u, int N, int Nc,
                       const double *cons_lo, const double *cons_hi, hipStream_t s) {
  const long long tot = rows * (x + u);
  hipLaunchKernelGGL(k_slew_split, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, Z, W, X, U, rows, x, u, N, Nc, cons_lo, cons_hi);
}
