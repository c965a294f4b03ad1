// kernels_soc.hip — elementwise kernels of the stage-wise second-order-cone extension (config E's thrust cones).
//
// The reference reaches user cones only through its pyjulia-only `extra_cstrs` tuples over the joint variable vector
// (README.md:219-239, PMPC.jl/src/main.jl:293-316); they cannot cross its C ABI.  The extension
// here covers the structured case those tuples are used for in MPC practice — one cone per (particle, stage) on that
// stage's controls,   || W u + w0 ||_2 <= v'u + v0,   shared W, w0, v, v0 —  next to the control boxes.
//
// Method: feasible primal-dual path following with Nesterov-Todd scaling.  Per stage the constraints are
//     s = A u + c in K,   K = R+ (each finite box side: u - lo, hi - u)  x  Q^{q+1} (s = (v'u + v0, W u + w0)),
// with duals z in K and s o z = mu e.  Eliminating dz = sigma mu s^-1 - z - W_nt^-2 ds (ds = A du) from the Newton system
// leaves the SAME structured LQ solve with
//     control Hessian  += A' W_nt^-2 A            gradient shift  wu = -sigma mu A' s^-1,
// where for a box side W_nt^-2 = z/s and for the cone W_nt^-2 = (2 (J wb)(J wb)' - J) / eta^2 (wb the NT scaling point, eta^2 =
// sqrt(s'Js / z'Jz), J = diag(1, -I)).  Both ride on the Riccati kernels' interior-point inputs: the Hessian block goes in as
// a FULL u x u `Du` (LQArgs.du_full; the box path uses a diagonal there), the shift as `wu`.  (Folding the block into the cost
// block R instead would make the kernels form R_eff (u - u_ref) and cancel 1e12-sized terms against the shift.)  The slacks are variables of their own (s += alpha ds, residual rp = A u + c - s
// carried in the Newton system: ds = A du + rp, wu += A' W_nt^-2 rp): recomputing hi - u at slack ~ mu/z ~ 1e-11 would tie
// their positivity to the last bits of u and stall the iteration near mu = 1e-10.  The step length keeps s, z inside K.
// Each iteration is a Mehrotra predictor-corrector pair on ONE factorisation: the predictor (sigma = 0) gives the affine step,
// the step polynomial mu_aff(alpha) = (S0 + alpha S1 + alpha^2 S2)/deg and the second-order term — c = ds*dz for a box side,
// c = W^-1 (lambda \ ((W^-1 ds) o (W dz))) for the cone (lambda = W z = W^-1 s, o and \ the Jordan product and division) —
// and the corrector solves only for the DIFFERENCE step with the gradient  A'(-sigma mu s^-1 + c)  (vector sweeps).
// Consensus stages carry ONE shared control: its constraints are counted once, on the owner rank's particle 0.
#include "pmpc_dev.h"

namespace {

constexpr int TB = 256, UMAX = 8, QMAX = 4;

__device__ __forceinline__ bool soc_counts(const SocArgs &a, int i, int j) { return j >= a.Nc || (i == 0 && a.owner); }

// cone slack s = (v'u + v0, W u + w0) of one stage
__device__ __forceinline__ void cone_slack(const SocArgs &a, const double *uvec, double *s, int u, int q) {
  double s0 = a.v0;
  for (int r = 0; r < u; r++) s0 += a.v[r] * uvec[r];
  s[0] = s0;
  for (int p = 0; p < q; p++) {
    double bp = a.w0[p];
    for (int r = 0; r < u; r++) bp += a.W[p * u + r] * uvec[r];
    s[1 + p] = bp;
  }
}
__device__ __forceinline__ double jdot(const double *x, const double *y, int q) {  // x'Jy
  double d = x[0] * y[0];
  for (int p = 1; p <= q; p++) d -= x[p] * y[p];
  return d;
}
// largest t in (0, cap] with x + t d strictly inside Q (x inside)
__device__ __forceinline__ double cone_ratio(const double *x, const double *d, int q, double cap) {
  double al = cap;
  if (d[0] < 0.0) al = fmin(al, x[0] / -d[0]);
  const double A0 = jdot(x, x, q), A1 = 2.0 * jdot(x, d, q), A2 = jdot(d, d, q);
  if (A2 < 0.0 || A1 < 0.0) {
    const double disc = A1 * A1 - 4.0 * A2 * A0;
    if (disc >= 0.0) {
      const double sq = sqrt(disc), t1 = 2.0 * A0 / (-A1 + sq), t2 = 2.0 * A0 / (-A1 - sq);
      if (t1 > 0.0) al = fmin(al, t1);
      if (t2 > 0.0) al = fmin(al, t2);
    }
  }
  return al;
}

// NT scaling of one cone pair (s, z): wb, eta, and products with W, W^-1, W^-2
struct NtScal {
  double wb[1 + QMAX], eta, ieta2;
  __device__ void init(const double *s, const double *z, int q) {
    const double ss = jdot(s, s, q), zz = jdot(z, z, q), rs = 1.0 / sqrt(ss), rz = 1.0 / sqrt(zz);
    double dotb = 0.0;
    for (int p = 0; p <= q; p++) dotb += s[p] * z[p] * rs * rz;
    const double ig = 0.5 / sqrt(0.5 * (1.0 + dotb));
    wb[0] = (s[0] * rs + z[0] * rz) * ig;
    for (int p = 1; p <= q; p++) wb[p] = (s[p] * rs - z[p] * rz) * ig;
    ieta2 = sqrt(zz / ss);
    eta = sqrt(sqrt(ss / zz));
  }
  // W x = eta [wb0 x0 + wb1'x1 ; x0 wb1 + x1 + wb1 (wb1'x1)/(1 + wb0)],  W^-1 x: same with wb1 -> -wb1 and 1/eta
  __device__ void mulW(const double *x, double *y, int q, bool inverse) const {
    const double sg = inverse ? -1.0 : 1.0, sc = inverse ? 1.0 / eta : eta;
    double d = 0.0;
    for (int p = 1; p <= q; p++) d += wb[p] * x[p];
    y[0] = sc * (wb[0] * x[0] + sg * d);
    for (int p = 1; p <= q; p++) y[p] = sc * (sg * x[0] * wb[p] + x[p] + wb[p] * d / (1.0 + wb[0]));
  }
  __device__ void mulWm2(const double *x, double *y, int q) const {  // W^-2 x = ieta2 (2 (J wb)(J wb)'x - J x)
    double jw = wb[0] * x[0];
    for (int p = 1; p <= q; p++) jw -= wb[p] * x[p];
    y[0] = ieta2 * (2.0 * wb[0] * jw - x[0]);
    for (int p = 1; p <= q; p++) y[p] = ieta2 * (-2.0 * wb[p] * jw + x[p]);
  }
};

// MODE 0: s = A u + c, z = mu s^-1 (centred start);  MODE 2: s = A u + c, z kept (warm start from remembered duals);
// MODE 1: Newton-system inputs from (U, s, z) — predictor (a.corr == 0): Hadd = A'W^-2 A and wu = A'W^-2 rp (sigma = 0);
// corrector (a.corr == 1): wu = the DIFFERENCE of the gradient shifts, A'(-sigma mu s^-1 + c), Hadd left as it is.
// Sums s'z and the cone count in every mode.
// UDT / QT: compile-time udim / cone rows (0 = run-time sizes): with constant trip counts the per-stage arrays stay in
// registers; with run-time bounds they live in scratch memory and the pass is ~3x slower
template <int MODE, int UDT, int QT>
__global__ void __launch_bounds__(TB) k_soc_prepare(SocArgs a, double *part_sum, double *part_cnt) {
  __shared__ double sh[TB];
  const long long tot = (long long)a.M * a.N;
  const bool corr = MODE == 1 && a.corr;
  double comp = 0.0, cnt = 0.0;
  for (long long k = blockIdx.x * (long long)TB + threadIdx.x; k < tot; k += (long long)gridDim.x * TB) {
    const int i = (int)(k / a.N), j = (int)(k % a.N), u = UDT ? UDT : a.u, q = UDT ? QT : a.q;
    const double *U = a.U + k * u;
    double *Ha = a.Hadd + k * u * u, *wu = a.wu + k * u;
    if (!soc_counts(a, i, j)) {
      if (MODE == 1) {
        if (!corr)
          for (int e = 0; e < u * u; e++) Ha[e] = 0.0;
        for (int r = 0; r < u; r++) wu[r] = 0.0;
      }
      continue;
    }
    double uu[UMAX], g[UMAX], H[UMAX][UMAX];
    for (int r = 0; r < u; r++) {
      uu[r] = U[r];
      g[r] = 0.0;
      for (int t = 0; t < u; t++) H[r][t] = 0.0;
    }
    bool ok = true;
    if (a.lo) {
      for (int r = 0; r < u; r++) {
        const double lo = a.lo[k * u + r], hi = a.hi[k * u + r];
        if (isfinite(lo)) {
          if (MODE != 1) a.sl[k * u + r] = uu[r] - lo;
          if (MODE == 0) a.zl[k * u + r] = a.mu / (uu[r] - lo);
          const double s = a.sl[k * u + r], z = a.zl[k * u + r], rp = (uu[r] - lo) - s, d = z / s;
          ok &= (s > 0.0) && (z > 0.0);
          comp += s * z; cnt += 1.0;
          if (corr) g[r] += -(a.sigmu - a.cl[k * u + r]) / s;  // A = +e_r
          else { g[r] += d * rp; H[r][r] += d; }
        }
        if (isfinite(hi)) {
          if (MODE != 1) a.su[k * u + r] = hi - uu[r];
          if (MODE == 0) a.zu[k * u + r] = a.mu / (hi - uu[r]);
          const double s = a.su[k * u + r], z = a.zu[k * u + r], rp = (hi - uu[r]) - s, d = z / s;
          ok &= (s > 0.0) && (z > 0.0);
          comp += s * z; cnt += 1.0;
          if (corr) g[r] -= -(a.sigmu - a.cu[k * u + r]) / s;  // A = -e_r
          else { g[r] -= d * rp; H[r][r] += d; }
        }
      }
    }
    if (q > 0) {
      double s[1 + QMAX], z[1 + QMAX], rp[1 + QMAX], t[1 + QMAX];
      double *sc = a.sc + k * (q + 1), *zc = a.zc + k * (q + 1);
      cone_slack(a, uu, rp, u, q);  // A u + c
      if (MODE != 1)
        for (int p = 0; p <= q; p++) sc[p] = rp[p];  // s = A u + c
      if (MODE == 0) {  // z = mu s^-1 = mu J s / (s'Js)
        const double ss0 = jdot(rp, rp, q);
        zc[0] = a.mu * rp[0] / ss0;
        for (int p = 1; p <= q; p++) zc[p] = -a.mu * rp[p] / ss0;
      }
      for (int p = 0; p <= q; p++) { s[p] = sc[p]; rp[p] -= s[p]; z[p] = zc[p]; }
      const double ss = jdot(s, s, q), zz = jdot(z, z, q);
      ok &= (s[0] > 0.0) && (ss > 0.0) && (z[0] > 0.0) && (zz > 0.0);
      double sz = 0.0;
      for (int p = 0; p <= q; p++) sz += s[p] * z[p];
      comp += sz; cnt += 1.0;
      if (MODE == 1) {
        if (corr) {  // t = -sigma mu s^-1 + c,  s^-1 = J s / s'Js
          const double *cc = a.cc + k * (q + 1);
          t[0] = -a.sigmu * s[0] / ss + cc[0];
          for (int p = 1; p <= q; p++) t[p] = a.sigmu * s[p] / ss + cc[p];
        } else {
          NtScal nt;
          nt.init(s, z, q);
          nt.mulWm2(rp, t, q);  // t = W^-2 rp
          double Aw[UMAX];      // A'(J wb)
          for (int r = 0; r < u; r++) {
            double v = a.v[r] * nt.wb[0];
            for (int p = 1; p <= q; p++) v -= a.W[(p - 1) * u + r] * nt.wb[p];
            Aw[r] = v;
          }
          for (int r = 0; r < u; r++)
            for (int c2 = 0; c2 < u; c2++) {
              double AJA = a.v[r] * a.v[c2];  // A'JA
              for (int p = 1; p <= q; p++) AJA -= a.W[(p - 1) * u + r] * a.W[(p - 1) * u + c2];
              H[r][c2] += nt.ieta2 * (2.0 * Aw[r] * Aw[c2] - AJA);
            }
        }
        for (int r = 0; r < u; r++) {  // g += A't
          double v = a.v[r] * t[0];
          for (int p = 1; p <= q; p++) v += a.W[(p - 1) * u + r] * t[p];
          g[r] += v;
        }
      }
    }
    if (!ok) *a.fail = 3;
    if (MODE == 1) {
      for (int r = 0; r < u; r++) {
        wu[r] = g[r];
        if (!corr)
          for (int t = 0; t < u; t++) Ha[r + u * t] = H[r][t];
      }
    }
  }
  sh[threadIdx.x] = comp;
  __syncthreads();
  for (int o = TB / 2; o > 0; o >>= 1) { if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) part_sum[blockIdx.x] = sh[0];
  __syncthreads();
  sh[threadIdx.x] = cnt;
  __syncthreads();
  for (int o = TB / 2; o > 0; o >>= 1) { if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) part_cnt[blockIdx.x] = sh[0];
}

// Steps ds = A du + rp, dz = (sigma mu s^-1 - c) - z - W^-2 ds of every cone (du = dU, or dU + dU2 in the corrector), stored;
// largest alpha in (0, 2] keeping s and z inside K.  The predictor (a.corr == 0, sigma mu = 0, c = 0) also stores the
// second-order terms c and the partial sums S1 = sum (s'dz + z'ds), S2 = sum ds'dz of the step polynomial.
template <int UDT, int QT>
__global__ void __launch_bounds__(TB) k_soc_step(SocArgs a, unsigned long long *amin_bits, double *part_s1, double *part_s2) {
  __shared__ double sh[TB];
  const long long tot = (long long)a.M * a.N;
  const bool corr = a.corr != 0;
  double al = 2.0;  // (the host takes min(1, 0.99 * this): a step that would end ON a boundary just beyond 1 is shortened too)
  double s1 = 0.0, s2 = 0.0;
  for (long long k = blockIdx.x * (long long)TB + threadIdx.x; k < tot; k += (long long)gridDim.x * TB) {
    const int i = (int)(k / a.N), j = (int)(k % a.N), u = UDT ? UDT : a.u, q = UDT ? QT : a.q;
    if (!soc_counts(a, i, j)) continue;  // (the copies of a shared control take the same step: nothing to test)
    double uu[UMAX], du[UMAX];
    for (int r = 0; r < u; r++) {
      uu[r] = a.U[k * u + r];
      du[r] = a.dU[k * u + r] + (corr ? a.dU2[k * u + r] : 0.0);
    }
    if (a.lo) {
      for (int r = 0; r < u; r++) {
        const double lo = a.lo[k * u + r], hi = a.hi[k * u + r];
        if (isfinite(lo)) {
          const double s = a.sl[k * u + r], z = a.zl[k * u + r], ds = du[r] + ((uu[r] - lo) - s);
          const double dz = (a.sigmu - (corr ? a.cl[k * u + r] : 0.0)) / s - z - (z / s) * ds;
          a.dsl[k * u + r] = ds; a.dzl[k * u + r] = dz;
          if (!corr) { a.cl[k * u + r] = ds * dz; s1 += s * dz + z * ds; s2 += ds * dz; }
          if (ds < 0.0) al = fmin(al, s / -ds);
          if (dz < 0.0) al = fmin(al, z / -dz);
        }
        if (isfinite(hi)) {
          const double s = a.su[k * u + r], z = a.zu[k * u + r], ds = -du[r] + ((hi - uu[r]) - s);
          const double dz = (a.sigmu - (corr ? a.cu[k * u + r] : 0.0)) / s - z - (z / s) * ds;
          a.dsu[k * u + r] = ds; a.dzu[k * u + r] = dz;
          if (!corr) { a.cu[k * u + r] = ds * dz; s1 += s * dz + z * ds; s2 += ds * dz; }
          if (ds < 0.0) al = fmin(al, s / -ds);
          if (dz < 0.0) al = fmin(al, z / -dz);
        }
      }
    }
    if (q > 0) {
      double s[1 + QMAX], ds[1 + QMAX], z[1 + QMAX], dz[1 + QMAX], t[1 + QMAX];
      const double *sc = a.sc + k * (q + 1), *zc = a.zc + k * (q + 1);
      double *cc = a.cc + k * (q + 1);
      cone_slack(a, uu, ds, u, q);  // A u + c
      for (int p = 0; p <= q; p++) { s[p] = sc[p]; ds[p] -= s[p]; z[p] = zc[p]; }  // ds = rp so far
      for (int r = 0; r < u; r++) ds[0] += a.v[r] * du[r];
      for (int p = 1; p <= q; p++)
        for (int r = 0; r < u; r++) ds[p] += a.W[(p - 1) * u + r] * du[r];
      NtScal nt;
      nt.init(s, z, q);
      nt.mulWm2(ds, t, q);
      const double ss = jdot(s, s, q);
      dz[0] = a.sigmu * s[0] / ss - z[0] - t[0] - (corr ? cc[0] : 0.0);
      for (int p = 1; p <= q; p++) dz[p] = -a.sigmu * s[p] / ss - z[p] - t[p] - (corr ? cc[p] : 0.0);
      double *dsc = a.dsc + k * (q + 1), *dzc = a.dzc + k * (q + 1);
      for (int p = 0; p <= q; p++) { dsc[p] = ds[p]; dzc[p] = dz[p]; }
      if (!corr) {
        // c = W^-1 (lambda \ ((W^-1 ds) o (W dz))),  lambda = W z
        double lam[1 + QMAX], x[1 + QMAX], y[1 + QMAX], pr[1 + QMAX], dv[1 + QMAX];
        nt.mulW(z, lam, q, false);
        nt.mulW(ds, x, q, true);
        nt.mulW(dz, y, q, false);
        pr[0] = 0.0;
        for (int p = 0; p <= q; p++) pr[0] += x[p] * y[p];
        for (int p = 1; p <= q; p++) pr[p] = x[0] * y[p] + y[0] * x[p];
        double ll = 0.0, lr = 0.0;
        for (int p = 1; p <= q; p++) { ll += lam[p] * lam[p]; lr += lam[p] * pr[p]; }
        dv[0] = (lam[0] * pr[0] - lr) / (lam[0] * lam[0] - ll);
        for (int p = 1; p <= q; p++) dv[p] = (pr[p] - dv[0] * lam[p]) / lam[0];
        nt.mulW(dv, t, q, true);
        for (int p = 0; p <= q; p++) { cc[p] = t[p]; s1 += s[p] * dz[p] + z[p] * ds[p]; s2 += ds[p] * dz[p]; }
      }
      al = fmin(al, cone_ratio(s, ds, q, 2.0));
      al = fmin(al, cone_ratio(z, dz, q, 2.0));
    }
  }
  sh[threadIdx.x] = al;
  __syncthreads();
  for (int o = TB / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) sh[threadIdx.x] = fmin(sh[threadIdx.x], sh[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double v = sh[0];
    if (!(v >= 0.0)) v = 0.0;
    atomicMin(amin_bits, (unsigned long long)__double_as_longlong(v));
  }
  if (!corr) {
    __syncthreads();
    sh[threadIdx.x] = s1;
    __syncthreads();
    for (int o = TB / 2; o > 0; o >>= 1) { if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) part_s1[blockIdx.x] = sh[0];
    __syncthreads();
    sh[threadIdx.x] = s2;
    __syncthreads();
    for (int o = TB / 2; o > 0; o >>= 1) { if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) part_s2[blockIdx.x] = sh[0];
  }
}

// the whole update in one launch: X += alpha (dX + dX2), U += alpha (dU + dU2), every slack / dual += alpha * its step
__global__ void __launch_bounds__(TB) k_soc_update(SocArgs a, double alpha, double *X, const double *dX, const double *dX2, double *U,
                                                   long long nx, long long nu, long long ncz) {
  const long long stride = (long long)gridDim.x * TB, t0 = blockIdx.x * (long long)TB + threadIdx.x;
  for (long long k = t0; k < nx; k += stride) X[k] += alpha * (dX[k] + dX2[k]);
  for (long long k = t0; k < nu; k += stride) {
    U[k] += alpha * (a.dU[k] + a.dU2[k]);
    a.sl[k] += alpha * a.dsl[k]; a.su[k] += alpha * a.dsu[k];
    a.zl[k] += alpha * a.dzl[k]; a.zu[k] += alpha * a.dzu[k];
  }
  for (long long k = t0; k < ncz; k += stride) { a.sc[k] += alpha * a.dsc[k]; a.zc[k] += alpha * a.dzc[k]; }
}

__global__ void k_soc_fill_u(double *U, const double *u0, long long tot, int u) {
  for (long long k = blockIdx.x * (long long)TB + threadIdx.x; k < tot; k += (long long)gridDim.x * TB) U[k] = u0[k % u];
}

}  // namespace

static unsigned soc_grid(const SocArgs &a) {
  long long b = ((long long)a.M * a.N + TB - 1) / TB;
  if (b > PMPC_RED_BLOCKS) b = PMPC_RED_BLOCKS;
  return (unsigned)b;
}
// every launch runs exactly PMPC_RED_BLOCKS-bounded grids and fills part_sum / part_cnt [0, grid): returns the grid size
// compiled (udim, cone rows) pairs; anything else takes the run-time-size instance
#define PMPC_SOC_DIMS(X) X(4, 2) X(4, 3) X(4, 1) X(3, 2) X(3, 1) X(2, 1)
template <int UDT, int QT>
static void soc_prepare_t(const SocArgs &a, int mode, unsigned g, double *ps, double *pc, hipStream_t s) {
  if (mode == 0) hipLaunchKernelGGL((k_soc_prepare<0, UDT, QT>), dim3(g), dim3(TB), 0, s, a, ps, pc);
  else if (mode == 2) hipLaunchKernelGGL((k_soc_prepare<2, UDT, QT>), dim3(g), dim3(TB), 0, s, a, ps, pc);
  else hipLaunchKernelGGL((k_soc_prepare<1, UDT, QT>), dim3(g), dim3(TB), 0, s, a, ps, pc);
}
int launch_soc_prepare(const SocArgs &a, int mode, double *part_sum, double *part_cnt, hipStream_t s) {
  const unsigned g = soc_grid(a);
#define X(ud, qq) if (a.u == ud && a.q == qq) { soc_prepare_t<ud, qq>(a, mode, g, part_sum, part_cnt, s); return (int)g; }
  PMPC_SOC_DIMS(X)
#undef X
  soc_prepare_t<0, 0>(a, mode, g, part_sum, part_cnt, s);
  return (int)g;
}
int launch_soc_step(const SocArgs &a, unsigned long long *amin_bits, double *part_s1, double *part_s2, hipStream_t s) {
  const unsigned g = soc_grid(a);
#define X(ud, qq) if (a.u == ud && a.q == qq) { hipLaunchKernelGGL((k_soc_step<ud, qq>), dim3(g), dim3(TB), 0, s, a, amin_bits, part_s1, part_s2); return (int)g; }
  PMPC_SOC_DIMS(X)
#undef X
  hipLaunchKernelGGL((k_soc_step<0, 0>), dim3(g), dim3(TB), 0, s, a, amin_bits, part_s1, part_s2);
  return (int)g;
}
void launch_soc_update(const SocArgs &a, double alpha, double *X, const double *dX, const double *dX2, double *U, long long nx,
                       long long nu, long long ncz, hipStream_t s) {
  long long b = (nx + TB - 1) / TB;
  if (b > 4096) b = 4096;
  hipLaunchKernelGGL(k_soc_update, dim3((unsigned)b), dim3(TB), 0, s, a, alpha, X, dX, dX2, U, nx, nu, ncz);
}
void launch_soc_fill_u(double *U, const double *u0, long long tot, int u, hipStream_t s) {
  long long b = (tot + TB - 1) / TB;
  if (b > 4096) b = 4096;
  hipLaunchKernelGGL(k_soc_fill_u, dim3((unsigned)b), dim3(TB), 0, s, U, u0, tot, u);
}
