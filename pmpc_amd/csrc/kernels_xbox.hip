// State boxes inside the active-set rounds (pmpc_dev.h, XboxArgs).  gfx950 only.
//
// The reference hands lx <= x <= ux to its sparse QP solver with every other row (lqp_utils.jl:318-333, OSQP / ECOS / COSMO treat
// them alike).  On the Riccati path a state is not a free variable: the rounds of kernels_as.hip put a CONTROL on its bound and hold
// it there with a penalty on a zero step, which a state cannot follow — it is where the dynamics put it.  A binding state box is
// therefore a row of a semismooth Newton iteration on its natural map
//     s - max(0, s - z) = 0,   s = x - lo  (or hi - x),   z >= 0 its multiplier:
// while s - z >= 0 the row is off; else it is held by a penalty with a multiplier estimate,  rho/2 (s_b + ds)^2 - z ds,  whose
// minimiser over the whole stage-structured problem is what the factor sweep computes when rho sits on the diagonal of the stage's
// state cost and  +-(rho s_b - z)  in its gradient (k_bwd_as<.., XBOX>); afterwards  z+ = z - rho s_new.  The penalty is moderate
// (1e7 x the cost scale: a 1e30 penalty would wipe the cost-to-go out of H_uu = R + B'(S + rho e e')B), so a held row sits on
// its bound to ~z/rho after one round and to round-off after two: rows further off than `tol` keep the rounds going (`open`, like an
// unconverged stage cone).  One block per particle: it sums its own counters, no atomics.
#include "pmpc_dev.h"

namespace {

__global__ void __launch_bounds__(256) k_xbox_step(XboxArgs a) {
  if (a.done && *a.done) return;
  const int i = blockIdx.x, t = threadIdx.x, xd = a.x, n = a.N * xd;
  __shared__ double redd[256];
  __shared__ int redi[2][256];
  // penalty of this particle: the largest diagonal cost entry of its horizon sets the scale (found by the prepare call of the attempt)
  if (!a.finish || !a.qmax) {
    double m = 0.0;
    for (int k = t; k < n; k += 256) m = fmax(m, fabs(a.Q[((size_t)i * n + k) * xd + (k % xd)]));
    redd[t] = m;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if (t < w) redd[t] = fmax(redd[t], redd[t + w]);
      __syncthreads();
    }
    if (t == 0 && a.qmax) a.qmax[i] = redd[0];
  } else {
    if (t == 0) redd[0] = a.qmax[i];
    __syncthreads();
  }
  const double pw = a.pw ? a.pw[i] : 1.0;
  // (the scale follows the particle's cost weight, with a floor: a weight of zero — the cone objective's weighted QPs can hand one out —
  //  must not switch the rows of that particle off)
  const double pws = fmax(pw, 1e-6);
  double rho = a.rho_scale * pws * (redd[0] + a.reg_x);
  if (!(rho > 0.0)) rho = a.rho_scale * pws;
  const double margin0 = 10.0 * (a.ctl ? a.ctl->tol_l : 1e-11 * a.dual_scale);
  int changed = 0, open = 0, jm = -1;  // jm: 1 + the highest stage with a changed / open row of this thread
  const bool clamped = a.keep_on_clamp && a.finish && a.cnt[3 * i + 1] > 0;  // (read by every thread before thread 0 adds to it, two barriers further down)
  // pass 1: the largest violation among the rows that would be newly held (a round holds only those within act_frac of it: holding
  // every violated row of a window at once over-constrains the stage — the rows behind the first usually clear once it is held)
  double vmax = 0.0;
  if (a.act_frac > 0.0) {
    for (int k = t; k < n; k += 256) {
      const size_t idx = (size_t)i * n + k;
      const double xv = a.X[idx];
      if (a.st[idx] == 0 && k % xd < a.ctrl_from) vmax = fmax(vmax, fmax(a.lo[idx] - xv, xv - a.hi[idx]));
    }
    __syncthreads();
    redd[t] = vmax;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if (t < w) redd[t] = fmax(redd[t], redd[t + w]);
      __syncthreads();
    }
    vmax = redd[0];
  }
  const double vthr = a.act_frac * vmax;
  for (int k = t; k < n; k += 256) {
    const size_t idx = (size_t)i * n + k;
    const double xv = a.X[idx], lo = a.lo[idx], hi = a.hi[idx];
    const int st = a.st[idx];
    const double z = a.z[idx];
    const double sl = xv - lo, sh = hi - xv;
    // (z+ = z - rho s_new holds for the sweep's own Newton step: where the forward sweep clamped a control of this particle the states
    //  are those of the clamped controls, rho times that difference is no multiplier — the old estimate stands for this round)
    double zn = z;
    if (a.finish && !clamped) zn = st == 1 ? z - rho * sl : (st == 2 ? z - rho * sh : 0.0);
    if (a.finish && st == 0) zn = 0.0;
    // s - z < 0 holds the side; a status changes only with a margin above the round's multiplier tolerance (hysteresis against the
    // flip-flopping of a weakly active row, as for the box multipliers and the stage cones)
    const double mg = a.finish ? margin0 + 1e-13 * fmax(1.0, fabs(xv)) : 0.0;
    const double wl = sl - (st == 1 ? zn : 0.0), wh = sh - (st == 2 ? zn : 0.0);
    int nst = 0;
    if (wl < (st == 1 ? mg : -mg)) nst = 1;
    else if (wh < (st == 2 ? mg : -mg)) nst = 2;
    bool deferred = false;
    if (st == 0 && nst != 0 && k % xd < a.ctrl_from && fmax(-sl, -sh) < vthr) { nst = 0; deferred = true; }  // violated, but not among the worst: next round
    const double zo = (nst != 0 && nst == st) ? zn : 0.0;  // (a newly held side starts without an estimate)
    if (a.finish) {
      if (nst != st || deferred) { changed++; jm = k / xd + 1; }  // (k ascends)
      else if (nst != 0) {
        const double sv = nst == 1 ? sl : sh, bd = nst == 1 ? lo : hi;
        if (fabs(sv) > a.tol * fmax(1.0, fabs(bd)) || fabs(zo - z) > a.z_tol * fmax(a.dual_scale, fabs(zo))) { open++; jm = k / xd + 1; }
      }
    }
    a.st[idx] = nst;
    a.z[idx] = zo;
    double D = 0.0, g = 0.0;
    if (nst == 1) { D = rho; g = rho * sl - zo; }
    else if (nst == 2) { D = rho; g = -(rho * sh - zo); }
    a.D[idx] = D;
    a.g[idx] = g;
  }
  redi[0][t] = changed;
  redi[1][t] = open;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (t < w) { redi[0][t] += redi[0][t + w]; redi[1][t] += redi[1][t + w]; }
    __syncthreads();
  }
  const int nchanged = redi[0][0], nopen = redi[1][0];
  if (a.jhi && a.finish) {  // (uniform)
    __syncthreads();
    redi[0][t] = jm;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if (t < w) redi[0][t] = redi[0][t] > redi[0][t + w] ? redi[0][t] : redi[0][t + w];
      __syncthreads();
    }
    if (t == 0 && redi[0][0] > a.jhi[i]) a.jhi[i] = redi[0][0];
  }
  if (t == 0) {
    a.open[i] = nopen;
    if (a.finish) {
      a.cnt[3 * i + 1] += nchanged;
      if (nchanged || nopen) a.settled[i] = 0;
    }
  }
}

// first guess of the statuses and multipliers from an interior-point iterate (the finish of the interior-point iteration): the
// side whose multiplier exceeds its slack is held — the rule of k_as_setup for the control boxes
__global__ void __launch_bounds__(256) k_xbox_from_ipm(Slab s, int *st, double *z) {
  for (long long k = blockIdx.x * 256ll + threadIdx.x; k < s.count; k += (long long)gridDim.x * 256) {
    const double lo = s.lo[k], hi = s.hi[k], x = s.z[k], ll = s.ll[k], lu = s.lu[k];
    bool aL = isfinite(lo) && ll > x - lo, aU = isfinite(hi) && lu > hi - x;
    if (aL && aU) { aL = ll >= lu; aU = !aL; }
    st[k] = aL ? 1 : (aU ? 2 : 0);
    z[k] = aL ? ll : (aU ? lu : 0.0);
  }
}

}  // namespace

void launch_xbox_from_ipm(const Slab &sx, int *st, double *z, hipStream_t s) {
  const long long blocks = (sx.count + 255) / 256;
  hipLaunchKernelGGL(k_xbox_from_ipm, dim3((unsigned)(blocks < 1024 ? (blocks > 0 ? blocks : 1) : 1024)), dim3(256), 0, s, sx, st, z);
}
void launch_xbox_step(const XboxArgs &a, hipStream_t s) { hipLaunchKernelGGL(k_xbox_step, dim3(a.M), dim3(256), 0, s, a); }
