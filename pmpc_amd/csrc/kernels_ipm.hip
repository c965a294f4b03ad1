// kernels_ipm.hip — elementwise / reduction kernels of the box-constraint interior-point layer.
//
// The reference hands `l <= G z <= u` (PMPC.jl/src/lqp_utils.jl:306-393, G = selector rows) to
// OSQP; here the same constraints are driven to complementarity by a Mehrotra predictor-corrector
// whose Newton systems are the structured LQ solves of kernels_generic.hip / kernels_fast.hip.
// All kernels are HBM-streaming, grid-stride, 256 threads; reductions are two-stage and
// deterministic (block partials summed in index order by a single lane).
//
// Per bounded variable z with bounds [lo, hi] (either side may be infinite):
//   slacks t_l, t_u > 0, multipliers l_l, l_u > 0, residuals r_l = z - lo - t_l, r_u = hi - z - t_u
//   D = l_l/t_l + l_u/t_u
//   w_l = (sigma*mu - c_l - l_l r_l)/t_l,  w_u likewise;  Newton gradient shift w = -w_l + w_u
//   dt_l = dz + r_l, dt_u = -dz + r_u, dl_l = w_l - l_l - (l_l/t_l) dz, dl_u = w_u - l_u + (l_u/t_u) dz
// Consensus controls (stages j < Nc) are ONE variable stored M times (lqp_utils.jl:17-61); they are
// counted once, on the owner rank's particle 0, whose bounds they use (lqp_utils.jl:329-330).
#include "pmpc_dev.h"

namespace {

constexpr int TB = 256;

__device__ __forceinline__ double slab_weight(const Slab &s, long long idx) {
  if (!s.is_u || s.Nc == 0) return 1.0;
  int j = (int)((idx / s.d) % s.N);
  if (j >= s.Nc) return 1.0;
  long long i = idx / ((long long)s.d * s.N);
  return (i == 0 && s.owner) ? 1.0 : 0.0;
}

struct Elem {
  bool ml, mu;
  double lo, hi, tl, tu, ll, lu, rl, ru, wl, wu, dtl, dtu, dll, dlu;
};

// everything the Newton step needs for one variable; corrector terms (sigmu, cl, cu) optional
__device__ __forceinline__ Elem load_elem(const Slab &s, long long k, double sigmu, bool corrector, bool with_dz) {
  Elem e;
  e.lo = s.lo[k];
  e.hi = s.hi[k];
  e.ml = isfinite(e.lo);
  e.mu = isfinite(e.hi);
  const double z = s.z[k];
  e.tl = s.tl[k]; e.tu = s.tu[k]; e.ll = s.ll[k]; e.lu = s.lu[k];
  e.rl = e.ml ? z - e.lo - e.tl : 0.0;
  e.ru = e.mu ? e.hi - z - e.tu : 0.0;
  const double cl = corrector ? s.cl[k] : 0.0, cu = corrector ? s.cu[k] : 0.0;
  e.wl = e.ml ? (sigmu - cl - e.ll * e.rl) / e.tl : 0.0;
  e.wu = e.mu ? (sigmu - cu - e.lu * e.ru) / e.tu : 0.0;
  if (with_dz) {
    const double dz = s.dz[k] + (s.dz2 ? s.dz2[k] : 0.0);
    e.dtl = e.ml ? dz + e.rl : 0.0;
    e.dtu = e.mu ? -dz + e.ru : 0.0;
    e.dll = e.ml ? e.wl - e.ll - (e.ll / e.tl) * dz : 0.0;
    e.dlu = e.mu ? e.wu - e.lu + (e.lu / e.tu) * dz : 0.0;
  }
  return e;
}

__device__ __forceinline__ double block_sum(double v, double *sh) {
  const int t = threadIdx.x;
  sh[t] = v;
  __syncthreads();
  for (int o = TB / 2; o > 0; o >>= 1) {
    if (t < o) sh[t] += sh[t + o];
    __syncthreads();
  }
  double r = sh[0];
  __syncthreads();
  return r;
}
__device__ __forceinline__ double block_max(double v, double *sh) {
  const int t = threadIdx.x;
  sh[t] = v;
  __syncthreads();
  for (int o = TB / 2; o > 0; o >>= 1) {
    if (t < o) sh[t] = fmax(sh[t], sh[t + o]);
    __syncthreads();
  }
  double r = sh[0];
  __syncthreads();
  return r;
}
__device__ __forceinline__ double block_min(double v, double *sh) {
  const int t = threadIdx.x;
  sh[t] = v;
  __syncthreads();
  for (int o = TB / 2; o > 0; o >>= 1) {
    if (t < o) sh[t] = fmin(sh[t], sh[t + o]);
    __syncthreads();
  }
  double r = sh[0];
  __syncthreads();
  return r;
}

__global__ void __launch_bounds__(TB) k_axpy(double *y, const double *x, double alpha, long long n) {
  for (long long k = blockIdx.x * (long long)TB + threadIdx.x; k < n; k += (long long)gridDim.x * TB) y[k] += alpha * x[k];
}
__global__ void __launch_bounds__(TB) k_fill(double *y, double v, long long n) {
  for (long long k = blockIdx.x * (long long)TB + threadIdx.x; k < n; k += (long long)gridDim.x * TB) y[k] = v;
}

// consensus-control bounds come from particle 0 (lqp_utils.jl:329-330): replicate them
__global__ void __launch_bounds__(TB) k_cons_bounds(double *lo, double *hi, int M, int N, int u, int Nc) {
  const long long per = (long long)Nc * u, tot = per * (M - 1);
  for (long long k = blockIdx.x * (long long)TB + threadIdx.x; k < tot; k += (long long)gridDim.x * TB) {
    long long i = 1 + k / per, e = k % per;
    lo[i * (long long)N * u + e] = lo[e];
    hi[i * (long long)N * u + e] = hi[e];
  }
}

// dynamics-consistent base point: free controls = U_prev, consensus controls = 0
__global__ void __launch_bounds__(TB) k_init_base(double *U, const double *U_prev, int M, int N, int u, int Nc) {
  const long long tot = (long long)M * N * u;
  for (long long k = blockIdx.x * (long long)TB + threadIdx.x; k < tot; k += (long long)gridDim.x * TB) {
    int j = (int)((k / u) % N);
    U[k] = j < Nc ? 0.0 : U_prev[k];
  }
}

__global__ void __launch_bounds__(TB) k_violation(Slab s, double *part_max) {
  __shared__ double sh[TB];
  double v = 0.0;
  for (long long k = blockIdx.x * (long long)TB + threadIdx.x; k < s.count; k += (long long)gridDim.x * TB) {
    const double z = s.z[k], lo = s.lo[k], hi = s.hi[k];
    if (isfinite(lo)) v = fmax(v, lo - z);
    if (isfinite(hi)) v = fmax(v, z - hi);
    if (!(z == z)) v = INFINITY;
  }
  v = block_max(v, sh);
  if (threadIdx.x == 0) part_max[blockIdx.x] = v;
}

// the same for the candidate point za + zb (state boxes of an active-set candidate, before anything is overwritten)
__global__ void __launch_bounds__(TB) k_violation_sum(Slab s, const double *za, const double *zb, double *part_max) {
  __shared__ double sh[TB];
  double v = 0.0;
  for (long long k = blockIdx.x * (long long)TB + threadIdx.x; k < s.count; k += (long long)gridDim.x * TB) {
    const double z = za[k] + zb[k], lo = s.lo[k], hi = s.hi[k];
    if (isfinite(lo)) v = fmax(v, lo - z);
    if (isfinite(hi)) v = fmax(v, z - hi);
    if (!(z == z)) v = INFINITY;
  }
  v = block_max(v, sh);
  if (threadIdx.x == 0) part_max[blockIdx.x] = v;
}

// pull the controls strictly inside their box so that control slack residuals start at zero
__global__ void __launch_bounds__(TB) k_ipm_clip(Slab s) {
  for (long long k = blockIdx.x * (long long)TB + threadIdx.x; k < s.count; k += (long long)gridDim.x * TB) {
    const double lo = s.lo[k], hi = s.hi[k];
    const bool ml = isfinite(lo), mu = isfinite(hi);
    double wid = (ml && mu) ? hi - lo : fmax(1.0, ml ? fabs(lo) : (mu ? fabs(hi) : 1.0));
    double z = s.z[k];
    if (ml) z = fmax(z, lo + 0.1 * wid);
    if (mu) z = fmin(z, hi - 0.1 * wid);
    s.z[k] = z;
  }
}

__global__ void __launch_bounds__(TB) k_ipm_init_slack(Slab s, double mu0, double thr_frac) {
  // thr_frac: lower clamp of the slacks as a fraction of the box width (cold start 1e-2; a warm start keeps the
  // remembered iterate's own slacks, clamped only against exact zeros)
  for (long long k = blockIdx.x * (long long)TB + threadIdx.x; k < s.count; k += (long long)gridDim.x * TB) {
    const double lo = s.lo[k], hi = s.hi[k], z = s.z[k];
    const bool ml = isfinite(lo), mu = isfinite(hi);
    const double wid = (ml && mu) ? hi - lo : 1.0;
    const double thr = fmax(thr_frac * wid, 1e-2 * thr_frac);
    const double tl = ml ? fmax(z - lo, thr) : 1.0, tu = mu ? fmax(hi - z, thr) : 1.0;
    s.tl[k] = tl;
    s.tu[k] = tu;
    s.ll[k] = ml ? mu0 / tl : 0.0;
    s.lu[k] = mu ? mu0 / tu : 0.0;
    s.cl[k] = 0.0;
    s.cu[k] = 0.0;
  }
}

__global__ void __launch_bounds__(TB) k_ipm_prepare(Slab s, int corrector, const IpmScal *sc, double *part_sum,
                                                    double *part_cnt, double *part_max) {
  __shared__ double sh[TB];
  const double sigmu = corrector ? sc->sigmu : 0.0;
  double comp = 0.0, cnt = 0.0, res = 0.0;
  for (long long k = blockIdx.x * (long long)TB + threadIdx.x; k < s.count; k += (long long)gridDim.x * TB) {
    Elem e = load_elem(s, k, sigmu, corrector != 0, false);
    if (!corrector) {
      s.D[k] = (e.ml ? e.ll / e.tl : 0.0) + (e.mu ? e.lu / e.tu : 0.0);
      const double wgt = slab_weight(s, k);
      comp += wgt * ((e.ml ? e.tl * e.ll : 0.0) + (e.mu ? e.tu * e.lu : 0.0));
      cnt += wgt * ((e.ml ? 1.0 : 0.0) + (e.mu ? 1.0 : 0.0));
      res = fmax(res, fmax(fabs(e.rl), fabs(e.ru)));
    }
    if (!corrector) s.w[k] = -e.wl + e.wu;
    else  // difference between the corrector and the predictor gradient shifts
      s.w[k] = -(e.ml ? (sigmu - s.cl[k]) / e.tl : 0.0) + (e.mu ? (sigmu - s.cu[k]) / e.tu : 0.0);
  }
  if (!corrector) {
    comp = block_sum(comp, sh);
    cnt = block_sum(cnt, sh);
    res = block_max(res, sh);
    if (threadIdx.x == 0) {
      part_sum[blockIdx.x] = comp;
      part_cnt[blockIdx.x] = cnt;
      part_max[blockIdx.x] = res;
    }
  }
}

// ratio test of one Newton step.  Predictor also stores the second-order terms c = dt*dl and the partial
// sums of  S1 = sum w (t dl + l dt),  S2 = sum w dt dl:  mu_aff(alpha) = (S0 + alpha S1 + alpha^2 S2)/cnt
// with S0 = comp_sum, so no second pass over the slab is needed once alpha_aff is known.
__global__ void __launch_bounds__(TB) k_ipm_step(Slab s, int corrector, IpmScal *sc, double *part_s1, double *part_s2) {
  __shared__ double sh[TB];
  const double sigmu = corrector ? sc->sigmu : 0.0;
  // barrier mode: no second-order correction — its fixed point is t*l = mu_target - dt_aff*dl_aff, not the centre
  const double so = sc->mu_target > 0.0 ? 0.0 : 1.0;
  double a = 1.0, s1 = 0.0, s2 = 0.0;
  for (long long k = blockIdx.x * (long long)TB + threadIdx.x; k < s.count; k += (long long)gridDim.x * TB) {
    Elem e = load_elem(s, k, sigmu, corrector != 0, true);
    if (e.ml) {
      if (e.dtl < 0.0) a = fmin(a, -e.tl / e.dtl);
      if (e.dll < 0.0) a = fmin(a, -e.ll / e.dll);
    }
    if (e.mu) {
      if (e.dtu < 0.0) a = fmin(a, -e.tu / e.dtu);
      if (e.dlu < 0.0) a = fmin(a, -e.lu / e.dlu);
    }
    const double cl = e.dtl * e.dll, cu = e.dtu * e.dlu;
    if (!corrector) {
      s.cl[k] = so * cl;
      s.cu[k] = so * cu;
    }
    const double wgt = slab_weight(s, k);
    s1 += wgt * ((e.ml ? e.tl * e.dll + e.ll * e.dtl : 0.0) + (e.mu ? e.tu * e.dlu + e.lu * e.dtu : 0.0));
    s2 += wgt * ((e.ml ? cl : 0.0) + (e.mu ? cu : 0.0));
  }
  a = block_min(a, sh);
  s1 = block_sum(s1, sh);
  s2 = block_sum(s2, sh);
  if (threadIdx.x == 0) {
    if (!(a >= 0.0)) a = 0.0;  // NaN guard
    atomicMin(&sc->amin_bits, (unsigned long long)__double_as_longlong(a));
    part_s1[blockIdx.x] = s1;
    part_s2[blockIdx.x] = s2;
  }
}


// One pass per IPM iteration over BOTH slabs: (optionally) take the previous step z,t,l += alpha*d,
// then the predictor preparation of the new iterate (D, w, partial sums of complementarity / count /
// slack residual) and the gradient pre-pass arrays of the fast Riccati path
//   gm = z - ref,   gd = reg (z - prev) + w      (consensus stages: w only on the owner's particle 0).
__device__ __forceinline__ void advance_slab(const SlabEx &x, int do_update, double alpha, double sigmu, double mu_t, double &comp,
                                             double &cnt, double &res, double &dev) {
  const Slab &s = x.s;
  for (long long k = blockIdx.x * (long long)TB + threadIdx.x; k < s.count; k += (long long)gridDim.x * TB) {
    double z = s.z[k], w = 0.0;
    if (x.bounded) {
      const double lo = s.lo[k], hi = s.hi[k];
      const bool ml = isfinite(lo), mu = isfinite(hi);
      double tl = s.tl[k], tu = s.tu[k], ll = s.ll[k], lu = s.lu[k];
      if (do_update) {  // take the corrector step: z, t, lambda += alpha * d (d rebuilt from the stored step dz + dz2)
        const double dz = s.dz[k] + s.dz2[k];
        const double rl = ml ? z - lo - tl : 0.0, ru = mu ? hi - z - tu : 0.0;
        const double wl = ml ? (sigmu - s.cl[k] - ll * rl) / tl : 0.0, wu = mu ? (sigmu - s.cu[k] - lu * ru) / tu : 0.0;
        const double dtl = ml ? dz + rl : 0.0, dtu = mu ? -dz + ru : 0.0;
        const double dll = ml ? wl - ll - (ll / tl) * dz : 0.0, dlu = mu ? wu - lu + (lu / tu) * dz : 0.0;
        z += alpha * dz;
        if (ml) { tl += alpha * dtl; ll += alpha * dll; s.tl[k] = tl; s.ll[k] = ll; }
        if (mu) { tu += alpha * dtu; lu += alpha * dlu; s.tu[k] = tu; s.lu[k] = lu; }
        s.z[k] = z;
      }
      const double rl = ml ? z - lo - tl : 0.0, ru = mu ? hi - z - tu : 0.0;
      s.D[k] = (ml ? ll / tl : 0.0) + (mu ? lu / tu : 0.0);
      w = (ml ? ll * rl / tl : 0.0) - (mu ? lu * ru / tu : 0.0);  // -w_l + w_u with sigma*mu = 0, c = 0
      s.w[k] = w;
      const double wgt = slab_weight(s, k);
      comp += wgt * ((ml ? tl * ll : 0.0) + (mu ? tu * lu : 0.0));
      cnt += wgt * ((ml ? 1.0 : 0.0) + (mu ? 1.0 : 0.0));
      res = fmax(res, fmax(fabs(rl), fabs(ru)));
      if (mu_t > 0.0 && wgt > 0.0) dev = fmax(dev, fmax(ml ? fabs(tl * ll - mu_t) : 0.0, mu ? fabs(tu * lu - mu_t) : 0.0));
      if (s.is_u && s.Nc > 0 && slab_weight(s, k) == 0.0) w = 0.0;  // consensus shift counted once
    } else if (do_update) {
      z += alpha * (s.dz[k] + s.dz2[k]);
      s.z[k] = z;
    }
    if (x.gm) {
      const double pw = x.pw ? x.pw[k / x.per] : 1.0;
      x.gm[k] = pw * (z - x.ref[k]);
      x.gd[k] = pw * x.reg * (z - x.prev[k]) + w;
    }
  }
}

__global__ void __launch_bounds__(TB) k_ipm_advance(SlabEx X, SlabEx U, int do_update, const IpmScal *sc, double *part_sum,
                                                    double *part_cnt, double *part_max) {
  __shared__ double sh[TB];
  const double alpha = do_update ? sc->alpha : 0.0, sigmu = do_update ? sc->sigmu : 0.0;
  const int B = gridDim.x;
  const double mu_t = sc->mu_target;
  double *part_dev = sc->part_dev;
  double comp = 0.0, cnt = 0.0, res = 0.0, dev = 0.0;
  advance_slab(X, do_update, alpha, sigmu, mu_t, comp, cnt, res, dev);
  comp = block_sum(comp, sh); cnt = block_sum(cnt, sh); res = block_max(res, sh);
  if (mu_t > 0.0) dev = block_max(dev, sh);
  if (threadIdx.x == 0) {
    part_sum[blockIdx.x] = comp; part_cnt[blockIdx.x] = cnt; part_max[blockIdx.x] = res;
    if (mu_t > 0.0) part_dev[blockIdx.x] = dev;
  }
  comp = cnt = res = dev = 0.0;
  advance_slab(U, do_update, alpha, sigmu, mu_t, comp, cnt, res, dev);
  comp = block_sum(comp, sh); cnt = block_sum(cnt, sh); res = block_max(res, sh);
  if (mu_t > 0.0) dev = block_max(dev, sh);
  if (threadIdx.x == 0) {
    part_sum[B + blockIdx.x] = comp; part_cnt[B + blockIdx.x] = cnt; part_max[B + blockIdx.x] = res;
    if (mu_t > 0.0) part_dev[B + blockIdx.x] = dev;
  }
}

// 64-lane deterministic reductions of the block partials (fixed order: lane-strided, then butterfly)
__device__ __forceinline__ double wave_sum(const double *p, int nb) {  // block of TB threads, fixed order
  __shared__ double sh[TB];
  double v = 0.0;
  for (int b = threadIdx.x; b < nb; b += TB) v += p[b];
  return block_sum(v, sh);
}
__device__ __forceinline__ double wave_max(const double *p, int nb) {
  __shared__ double sh[TB];
  double v = 0.0;
  for (int b = threadIdx.x; b < nb; b += TB) v = fmax(v, p[b]);
  return block_max(v, sh);
}

// Scalar bookkeeping of the IPM with ONE cross-rank exchange per sync point.  PACK finalises this rank's
// block partials and writes its row of the exchange table xch[world][8] (other rows zeroed); the host
// all-reduces (sum) the table over RCCL — an all-gather in disguise, so min / max / sum quantities travel
// together; UNPACK combines the rows and derives the iteration scalars.  world == 1: both in one launch.
//   columns: 0 step ratio (min)  1 S1 (sum)  2 S2 (sum)  3 failure flag (max)  4 comp (sum)  5 count (sum)
//            6 slack residual (max)  7 bound violation (max)
// phases: 0 reset | 1 violation of the equality-only optimum | 2 IPM start (mu) |
//         3 predictor (alpha_aff, mu_aff polynomial, sigma) | 4 corrector (alpha, nu, next mu / residual)
__global__ void __launch_bounds__(TB) k_ipm_exchange(int phase, int do_pack, int do_unpack, IpmScal *sc, const int *fail,
                                                     double *xch, int rank, int world, const double *part_sum,
                                                     const double *part_cnt, const double *part_max, int nb, double mu_target,
                                                     double *part_dev, IpmScal *mirror, unsigned long long *mirror_seq,
                                                     unsigned long long seq) {
  const unsigned long long one_bits = (unsigned long long)__double_as_longlong(1.0);
  const bool l0 = threadIdx.x == 0;
  if (phase == 0) {
    if (l0) {
      sc->comp_sum = sc->cnt = sc->muaff_sum = sc->pad0 = 0.0;
      sc->res_max = sc->viol_max = 0.0;
      sc->amin_bits = one_bits;
      sc->mu = sc->sigma = sc->sigmu = 0.0;
      sc->alpha_aff = sc->alpha = 1.0;
      sc->nu = 1.0;
      sc->iter = 0;
      sc->status = 0;
      sc->mu_target = mu_target;
      sc->dev_max = 0.0;
      sc->part_dev = part_dev;
    }
    return;
  }
  if (do_pack) {
    double row[8] = {1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (phase == 1) {
      row[7] = wave_max(part_max, nb);
    } else if (phase == 2) {
      row[4] = wave_sum(part_sum, nb);
      row[5] = wave_sum(part_cnt, nb);
      row[6] = wave_max(part_max, nb);
      if (sc->mu_target > 0.0) row[7] = wave_max(sc->part_dev, nb);
    } else {
      row[1] = wave_sum(part_sum, nb);
      row[2] = wave_sum(part_cnt, nb);
    }
    row[0] = __longlong_as_double((long long)sc->amin_bits);
    row[3] = (double)(*fail);
    for (int k = threadIdx.x; k < world * 8; k += TB) {
      const int r = k >> 3, cidx = k & 7;
      double v = 0.0;
      if (r == rank) {
#pragma unroll
        for (int q = 0; q < 8; q++) v = (cidx == q) ? row[q] : v;
      }
      xch[k] = v;
    }
  }
  if (do_pack && do_unpack) __syncthreads();
  if (do_unpack && l0) {
    double amin = 1.0, s1 = 0.0, s2 = 0.0, fl = 0.0, comp = 0.0, cnt = 0.0, res = 0.0, viol = 0.0;
    for (int r = 0; r < world; r++) {  // fixed rank order: every rank computes bit-identical scalars
      const double *q = xch + r * 8;
      amin = fmin(amin, q[0]); s1 += q[1]; s2 += q[2]; fl = fmax(fl, q[3]);
      comp += q[4]; cnt += q[5]; res = fmax(res, q[6]); viol = fmax(viol, q[7]);
    }
    sc->status = (int)fl;
    sc->amin_bits = one_bits;
    if (phase == 1) {
      sc->viol_max = viol;
    } else if (phase == 2) {
      sc->comp_sum = comp; sc->cnt = cnt; sc->res_max = res;
      sc->mu = comp / fmax(cnt, 1.0);
      sc->dev_max = viol;
    } else if (phase == 3) {  // mu_aff(alpha_aff) = (S0 + a S1 + a^2 S2)/cnt, sigma = (mu_aff/mu)^3
      sc->alpha_aff = amin;
      const double mu_aff = (sc->comp_sum + amin * (s1 + amin * s2)) / fmax(sc->cnt, 1.0);
      const double r3 = mu_aff / sc->mu;
      sc->sigma = r3 * r3 * r3;
      sc->sigmu = fmax(sc->sigma * sc->mu, sc->mu_target);  // barrier mode: never aim below the target centrality
    } else {  // phase 4: step length of the corrector, then the scalars of the NEXT iterate by the same polynomial
      double a = amin;
      if (a < 1.0) a = fmin(1.0, fmax(0.99, 1.0 - sc->mu) * a);
      sc->alpha = a;
      sc->nu *= (1.0 - a);
      sc->iter += 1;
      sc->comp_sum = sc->comp_sum + a * (s1 + a * s2);
      sc->mu = sc->comp_sum / fmax(sc->cnt, 1.0);
      sc->res_max *= (1.0 - a);
    }
    if (mirror) {  // zero-copy publication to host-coherent memory: the host polls `seq` instead of a blit + stream sync
      *mirror = *sc;
      __threadfence_system();
      *(volatile unsigned long long *)mirror_seq = seq;
    }
  }
}

// ---- primal-dual active-set finish of the box interior-point iteration (control boxes) ---------------------------------
// Once mu is small the interior-point iterate has identified the active set: lower side if l_l > slack_l, upper side
// likewise.  The finish solves the equality-constrained QP on that set EXACTLY with one structured solve: the base point is
// the iterate with its active controls moved ONTO their bounds (and the states rolled out again), the active controls are
// held there by a quadratic penalty `big` on their step — du_b = -(grad L)_b / (H_bb + big), i.e. zero to ~1e-14, and
// -big du_b is the multiplier, to full relative precision (no cancellation: the penalty's target is du_b = 0).  The check
// pass verifies the KKT signs (multipliers of the active set >= 0, inactive controls inside their boxes) and applies the
// primal-dual active-set update where they fail; an unchanged set is the optimum of the QP (lqp_utils.jl:306-393 boxes
// included), with complementarity exactly zero.  act: 0 free, 1 at the lower bound, 2 at the upper bound.
__global__ void __launch_bounds__(TB) k_as_setup(Slab s, int from_ipm, int keep_base, int *act, double *ztry, double big) {
  for (long long k = blockIdx.x * (long long)TB + threadIdx.x; k < s.count; k += (long long)gridDim.x * TB) {
    const double lo = s.lo[k], hi = s.hi[k];
    // shared controls: ONE base value for every particle — particle 0's (a caller's U_prev may not have them equal)
    const long long per = (long long)s.N * s.d;
    const double z = (s.is_u && (k / s.d) % s.N < s.Nc) ? s.z[k % per] : s.z[k];
    int a;
    if (from_ipm == 2) {  // cold: the boxes the equality-only optimum violates
      a = z < lo ? 1 : (z > hi ? 2 : 0);
      act[k] = a;
    } else if (from_ipm) {
      const double ll = s.ll[k], lu = s.lu[k];
      bool aL = isfinite(lo) && ll > z - lo, aU = isfinite(hi) && lu > hi - z;
      if (aL && aU) { aL = ll >= lu; aU = !aL; }
      a = aL ? 1 : (aU ? 2 : 0);
      act[k] = a;
    } else {
      a = act[k];
    }
    if (!keep_base) ztry[k] = a == 1 ? lo : (a == 2 ? hi : fmin(fmax(z, lo), hi));  // (keep_base: only D changes)
    if (lo > hi) ztry[k] = NAN;  // an empty box: the NaN reaches the counters, the rounds stop and the solve fails as the
                                 // reference's does (NaN outputs, osqp_solver.jl:65-71)
    if (s.D) s.D[k] = a ? big : 0.0;  // (the fast-path rounds read the statuses themselves: no penalty / shift arrays)
    if (s.w) s.w[k] = 0.0;
  }
}

// counters[0] += variables released (negative multiplier), counters[1] += variables activated (outside their box),
// counters[2] |= a NaN step was seen; worst_bits = max over the changes of the multiplier's resp. the violation's magnitude
__global__ void __launch_bounds__(TB) k_as_check(Slab s, int *act, const double *ztry, double big, double tol_p, double tol_l,
                                                 int *counters, unsigned long long *worst_bits) {
  __shared__ double sh[TB];
  double rel = 0.0, add = 0.0, bad = 0.0, worst = 0.0;
  for (long long k = blockIdx.x * (long long)TB + threadIdx.x; k < s.count; k += (long long)gridDim.x * TB) {
    const int a = act[k];
    const double dz = s.dz[k];
    if (!(dz == dz)) { bad = 1.0; continue; }
    if (a) {
      const double lam = a == 1 ? -big * dz : big * dz;  // multiplier of the active side
      if (lam < -tol_l) { act[k] = 0; rel += 1.0; worst = fmax(worst, -lam); }
    } else {
      const double lo = s.lo[k], hi = s.hi[k], zt = ztry[k] + dz;
      if (zt < lo - tol_p * fmax(1.0, fabs(lo))) { act[k] = 1; add += 1.0; worst = fmax(worst, lo - zt); }
      else if (zt > hi + tol_p * fmax(1.0, fabs(hi))) { act[k] = 2; add += 1.0; worst = fmax(worst, zt - hi); }
    }
  }
  rel = block_sum(rel, sh);
  add = block_sum(add, sh);
  bad = block_max(bad, sh);
  worst = block_max(worst, sh);
  if (threadIdx.x == 0) {
    if (rel > 0.0) atomicAdd(&counters[0], (int)fmin(rel, 1e6));
    if (add > 0.0) atomicAdd(&counters[1], (int)fmin(add, 1e6));
    if (bad > 0.0) atomicMax(&counters[2], 1);
    if (worst > 0.0) atomicMax(worst_bits, (unsigned long long)__double_as_longlong(worst));
  }
}

// accepted active-set point in ONE pass: controls (bound on the active set, base + step kept inside the box elsewhere) and
// states (base + step), each written to the workspace (the next solve's warm start) AND to the caller's output
__global__ void __launch_bounds__(TB) k_as_accept_all(Slab s, const int *act, const double *ztry, double *Uout, const double *xtry,
                                                      const double *dx, long long nx, double *Xws, double *Xout) {
  const long long stride = (long long)gridDim.x * TB, t0 = blockIdx.x * (long long)TB + threadIdx.x;
  for (long long k = t0; k < s.count; k += stride) {
    const int a = act[k];
    const double lo = s.lo[k], hi = s.hi[k];
    const double z = a == 1 ? lo : (a == 2 ? hi : fmin(fmax(ztry[k] + s.dz[k], lo), hi));
    s.z[k] = z;
    Uout[k] = z;
  }
  for (long long k = t0; k < nx; k += stride) {
    const double x = xtry[k] + dx[k];
    Xws[k] = x;
    Xout[k] = x;
  }
}

// Input checks of the host-pointer ABI, on the device after the upload (the host would scan ~320 MB per call at config D):
// flags[0] |= 1 if lx or ux holds a NaN, flags[1] |= 1 if lu or uu does (sentinels of c_interface.jl:56-70),
// flags[2] |= 1 if some Q_j or R_j block is not exactly symmetric (then the generic kernels must be used).
__global__ void __launch_bounds__(TB) k_host_checks(const double *lx, const double *ux, long long nx, const double *lu, const double *uu,
                                                    long long nu, const double *Q, long long nq, int x, const double *R, long long nr,
                                                    int u, int *flags) {
  const long long stride = (long long)gridDim.x * TB, t0 = blockIdx.x * (long long)TB + threadIdx.x;
  int fx = 0, fu = 0, fs = 0;
  for (long long k = t0; k < nx; k += stride) fx |= (lx[k] != lx[k]) || (ux[k] != ux[k]);
  for (long long k = t0; k < nu; k += stride) fu |= (lu[k] != lu[k]) || (uu[k] != uu[k]);
  for (long long k = t0; k < nq; k += stride) {  // element (r, c) of block b against (c, r)
    const long long b = k / (x * x), e = k - b * (x * x);
    const int r = (int)(e % x), c = (int)(e / x);
    fs |= Q[k] != Q[b * (x * x) + c + (long long)x * r];
  }
  for (long long k = t0; k < nr; k += stride) {
    const long long b = k / (u * u), e = k - b * (u * u);
    const int r = (int)(e % u), c = (int)(e / u);
    fs |= R[k] != R[b * (u * u) + c + (long long)u * r];
  }
  if (fx) atomicOr(&flags[0], 1);
  if (fu) atomicOr(&flags[1], 1);
  if (fs) atomicOr(&flags[2], 1);
}

// out (column-major rows x cols blocks) <- in (row-major blocks); n = total element count
__global__ void __launch_bounds__(TB) k_block_transpose(const double *__restrict__ in, double *__restrict__ out, int rows, int cols,
                                                        long long n) {
  const int rc = rows * cols;
  for (long long k = blockIdx.x * (long long)TB + threadIdx.x; k < n; k += (long long)gridDim.x * TB) {
    const long long b = k / rc;
    const int e = (int)(k - b * rc), r = e % rows, c = e / rows;
    out[k] = in[b * rc + r * cols + c];
  }
}

inline int grid_for(long long n) {
  long long b = (n + TB - 1) / TB;
  if (b > PMPC_RED_BLOCKS) b = PMPC_RED_BLOCKS;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

void launch_block_transpose(const double *in, double *out, int rows, int cols, long long n, hipStream_t s) {
  hipLaunchKernelGGL(k_block_transpose, dim3(grid_for(n)), dim3(TB), 0, s, in, out, rows, cols, n);
}
void launch_host_checks(const double *lx, const double *ux, long long nx, const double *lu, const double *uu, long long nu,
                        const double *Q, long long nq, int x, const double *R, long long nr, int u, int *flags, hipStream_t s) {
  hipLaunchKernelGGL(k_host_checks, dim3(2048), dim3(TB), 0, s, lx, ux, nx, lu, uu, nu, Q, nq, x, R, nr, u, flags);
}
void launch_axpy(double *y, const double *xv, double alpha, long long n, hipStream_t s) {
  hipLaunchKernelGGL(k_axpy, dim3(grid_for(n) * 4), dim3(TB), 0, s, y, xv, alpha, n);
}
void launch_fill(double *y, double v, long long n, hipStream_t s) {
  hipLaunchKernelGGL(k_fill, dim3(grid_for(n) * 4), dim3(TB), 0, s, y, v, n);
}
void launch_cons_bounds(double *lo, double *hi, int M, int N, int u, int Nc, hipStream_t s) {
  if (M <= 1 || Nc <= 0) return;
  hipLaunchKernelGGL(k_cons_bounds, dim3(grid_for((long long)Nc * u * (M - 1))), dim3(TB), 0, s, lo, hi, M, N, u, Nc);
}
void launch_init_base(double *U, const double *U_prev, int M, int N, int u, int Nc, hipStream_t s) {
  hipLaunchKernelGGL(k_init_base, dim3(grid_for((long long)M * N * u)), dim3(TB), 0, s, U, U_prev, M, N, u, Nc);
}
// NOTE: every reduction kernel below runs exactly PMPC_RED_BLOCKS blocks so that unused
// partial slots never hold stale values.
void launch_violation(const Slab &sl, double *part_max, hipStream_t s) {
  hipLaunchKernelGGL(k_violation, dim3(PMPC_RED_BLOCKS), dim3(TB), 0, s, sl, part_max);
}
void launch_as_setup(const Slab &sl, int from_ipm, int keep_base, int *act, double *ztry, double big, hipStream_t s) {
  hipLaunchKernelGGL(k_as_setup, dim3(PMPC_RED_BLOCKS), dim3(TB), 0, s, sl, from_ipm, keep_base, act, ztry, big);
}
void launch_as_check(const Slab &sl, int *act, const double *ztry, double big, double tol_p, double tol_l, int *counters,
                     unsigned long long *worst_bits, hipStream_t s) {
  hipLaunchKernelGGL(k_as_check, dim3(PMPC_RED_BLOCKS), dim3(TB), 0, s, sl, act, ztry, big, tol_p, tol_l, counters, worst_bits);
}
void launch_as_accept_all(const Slab &sl, const int *act, const double *ztry, double *Uout, const double *xtry, const double *dx,
                          long long nx, double *Xws, double *Xout, hipStream_t s) {
  hipLaunchKernelGGL(k_as_accept_all, dim3(2 * PMPC_RED_BLOCKS), dim3(TB), 0, s, sl, act, ztry, Uout, xtry, dx, nx, Xws, Xout);
}
void launch_violation_sum(const Slab &sl, const double *za, const double *zb, double *part_max, hipStream_t s) {
  hipLaunchKernelGGL(k_violation_sum, dim3(PMPC_RED_BLOCKS), dim3(TB), 0, s, sl, za, zb, part_max);
}
void launch_ipm_clip(const Slab &sl, hipStream_t s) {
  hipLaunchKernelGGL(k_ipm_clip, dim3(grid_for(sl.count)), dim3(TB), 0, s, sl);
}
void launch_ipm_init_slack(const Slab &sl, double mu0, hipStream_t s, double thr_frac) {
  hipLaunchKernelGGL(k_ipm_init_slack, dim3(grid_for(sl.count)), dim3(TB), 0, s, sl, mu0, thr_frac);
}
void launch_ipm_prepare(const Slab &sl, int corrector, const IpmScal *sc, double *part_sum, double *part_cnt,
                        double *part_max, hipStream_t s) {
  hipLaunchKernelGGL(k_ipm_prepare, dim3(PMPC_RED_BLOCKS), dim3(TB), 0, s, sl, corrector, sc, part_sum, part_cnt, part_max);
}
void launch_ipm_step(const Slab &sl, int corrector, IpmScal *sc, double *part_s1, double *part_s2, hipStream_t s) {
  hipLaunchKernelGGL(k_ipm_step, dim3(PMPC_RED_BLOCKS), dim3(TB), 0, s, sl, corrector, sc, part_s1, part_s2);
}
void launch_ipm_advance(const SlabEx &X, const SlabEx &U, int do_update, const IpmScal *sc, double *part_sum, double *part_cnt,
                        double *part_max, hipStream_t s) {
  hipLaunchKernelGGL(k_ipm_advance, dim3(PMPC_RED_BLOCKS), dim3(TB), 0, s, X, U, do_update, sc, part_sum, part_cnt, part_max);
}
void launch_ipm_exchange(int phase, bool pack, bool unpack, IpmScal *sc, const int *fail, double *xch, int rank, int world,
                         const double *part_sum, const double *part_cnt, const double *part_max, int nblocks, hipStream_t s,
                         double mu_target, double *part_dev, IpmScal *mirror, unsigned long long *mirror_seq,
                         unsigned long long seq) {
  hipLaunchKernelGGL(k_ipm_exchange, dim3(1), dim3(TB), 0, s, phase, pack ? 1 : 0, unpack ? 1 : 0, sc, fail, xch, rank, world,
                     part_sum, part_cnt, part_max, nblocks, mu_target, part_dev, mirror, mirror_seq, seq);
}
