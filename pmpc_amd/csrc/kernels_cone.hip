// kernels_cone.hip — stage-wise second-order cones INSIDE the active-set rounds (gfx950).
//
// The reference reaches user cones through `extra_cstrs` tuples handed to ECOS (PMPC.jl/src/main.jl:293-316, README.md:219-239);
// here the structured case — one cone  s = A u + c in K = {(s0, sb) : |sb| <= s0},  A = [v'; W], c = (v0, w0)  on the controls of every
// (particle, stage), next to the control boxes — is solved by a semismooth Newton method on the natural map
//     Phi(u, z) = s - Proj_K(s - z) = 0,     grad_u J = A'z + box multipliers,
// whose Newton systems are exactly the structured solves of an active-set round (kernels_as.hip).  With w = s - z the generalised
// Jacobian of the projection has three cases, and each turns into terms the Riccati sweeps already know how to carry:
//   interior  (|wb| <=  w0):  z+ = 0, the cone is off;
//   polar     (|wb| <= -w0):  s+ = 0 (held at the apex): penalty rho/2 |s_b + A du|^2 with the multiplier estimate, z+ = zhat - rho s+;
//   otherwise: ONE equality e-' s+ = 0 along e- = (1, -wh)/sqrt2, wh = wb/|wb| (penalty + multiplier estimate nu = e-' z), and a
//              finite curvature (1 - theta)/theta, theta = (1 + w0/|wb|)/2, on the tangential part (I - wh wh') sb — no smoothing, and
//              no singularity at the apex: the curvature is measured through s - z, not through s alone.
// Per round ONE elementwise pass (thread = (particle, stage)) FINISHES the round just swept — multipliers from each stage's own
// Newton step before clamping (a multiplier update is valid for that step only; the clamped point decides the next case), change /
// open counters for the round control — and PREPARES the next: the (u x u) block cone_H and the vector cone_g the backward sweep
// adds to H_uu and to the control gradient.  The boxes stay with the primal-dual active-set rule of the forward sweep.
// Validated first as a numpy model against the sparse cone oracle (tools/proto/cone_ssn.py).
#include "pmpc_dev.h"

namespace {

template <int UD, int Q>
__global__ void __launch_bounds__(256) k_cone_step(ConeArgs a) {
  if (a.done && *a.done) return;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long rows = (long long)a.M * a.N;
  if (idx >= rows) return;
  const int i = (int)(idx / a.N), j = (int)(idx - (long long)i * a.N);
  constexpr int Q1 = Q + 1;
  double A[Q1][UD], cc[Q1];
#pragma unroll
  for (int r = 0; r < Q1; r++) {
    cc[r] = a.c[r];
#pragma unroll
    for (int k = 0; k < UD; k++) A[r][k] = a.A[r * UD + k];
  }
  auto s_of = [&](const double *u, double *s) {
#pragma unroll
    for (int r = 0; r < Q1; r++) {
      double v = cc[r];
#pragma unroll
      for (int k = 0; k < UD; k++) v = fma(A[r][k], u[k], v);
      s[r] = v;
    }
  };
  // classification of w = s - z: 0 interior, 1 boundary, 2 polar; wh = wb / |wb|
  auto classify = [&](const double *s, const double *z, double *wh, double &w0, double &nb) -> int {
    double wb[Q], n2 = 0.0;
    w0 = s[0] - z[0];
#pragma unroll
    for (int r = 0; r < Q; r++) { wb[r] = s[r + 1] - z[r + 1]; n2 = fma(wb[r], wb[r], n2); }
    nb = sqrt(n2);
    const double inv = nb > 0.0 ? 1.0 / nb : 0.0;
#pragma unroll
    for (int r = 0; r < Q; r++) wh[r] = wb[r] * inv;
    return nb <= w0 ? 0 : (nb <= -w0 ? 2 : 1);
  };

  double u[UD], s[Q1], z[Q1];
#pragma unroll
  for (int k = 0; k < UD; k++) u[k] = a.U[idx * UD + k];
  s_of(u, s);
  double *rec = a.rec + idx * PMPC_CONE_REC;
  bool bad = false;
#pragma unroll
  for (int r = 0; r < Q1; r++) bad |= !(s[r] == s[r]);
  // A cone that was off (no multiplier, zero terms) and is still off — most cones, most rounds — needs none of the rest: s alone
  // classifies it (w = s - 0), nothing counts, nothing is to be written.  Decided from the controls and ONE word of the record.
  if (a.finish && (int)rec[0] == 0 && !bad) {
    double zz[Q1], whx[Q], w0x, nbx;
#pragma unroll
    for (int r = 0; r < Q1; r++) zz[r] = 0.0;
    int cn = classify(s, zz, whx, w0x, nbx);
    if (cn != 0) {  // (the hysteresis of the general path below: a change counts only beyond the round's tolerance)
      const double m = cn == 2 ? -w0x - nbx : fmin(nbx - w0x, nbx + w0x);
      double ns0 = 0.0;
#pragma unroll
      for (int r = 0; r < Q1; r++) ns0 = fmax(ns0, fabs(s[r]));
      if (m <= 10.0 * (a.ctl ? a.ctl->tol_l : 1e-11 * a.dual_scale) + 1e-13 * fmax(1.0, ns0)) cn = 0;
    }
    if (cn == 0) return;
  }
#pragma unroll
  for (int r = 0; r < Q1; r++) z[r] = a.z[idx * Q1 + r];

  int changed = 0, open = 0, decided = -1;
  if (a.finish) {
    // ---- finish the round just swept -----------------------------------------------------------------------------------------
    const int case_old = (int)rec[0];
    const double rho = rec[1], curv = rec[2], nu_hat = rec[3];
    double wh_o[Q], sb_o[Q1], ur[UD], sr[Q1], zn[Q1];
#pragma unroll
    for (int r = 0; r < Q; r++) wh_o[r] = rec[4 + r];
#pragma unroll
    for (int r = 0; r < Q1; r++) sb_o[r] = rec[4 + Q + r];
#pragma unroll
    for (int k = 0; k < UD; k++) ur[k] = a.Uraw[idx * UD + k];
    s_of(ur, sr);
    if (case_old == 1) {
      // z+ = nu+ e-  -  curv (I - wh wh') sb_raw,   nu+ = nu_hat - rho e-' s_raw,   e- = (1, -wh) / sqrt2
      const double rs2 = 0.70710678118654752440;
      double es = sr[0], whs = 0.0;
#pragma unroll
      for (int r = 0; r < Q; r++) { es = fma(-wh_o[r], sr[r + 1], es); whs = fma(wh_o[r], sr[r + 1], whs); }
      es *= rs2;
      const double nu = nu_hat - rho * es;
      zn[0] = nu * rs2;
#pragma unroll
      for (int r = 0; r < Q; r++) zn[r + 1] = -nu * rs2 * wh_o[r] - curv * (sr[r + 1] - whs * wh_o[r]);
    } else if (case_old == 2) {
#pragma unroll
      for (int r = 0; r < Q1; r++) zn[r] = z[r] - rho * sr[r];
    } else {
#pragma unroll
      for (int r = 0; r < Q1; r++) zn[r] = 0.0;
    }
    double wh_n[Q], w0, nb;
    int case_new = classify(s, zn, wh_n, w0, nb);
    // hysteresis against flip-flopping of a (nearly) weakly active cone — the counterpart of the widening sign tolerance of the
    // box multipliers: a case change counts only if the new classification holds with a margin above the round's tolerance
    if (case_new != case_old) {
      const double m = case_new == 0 ? w0 - nb : (case_new == 2 ? -w0 - nb : fmin(nb - w0, nb + w0));
      double ns0 = 0.0;
#pragma unroll
      for (int r = 0; r < Q1; r++) ns0 = fmax(ns0, fabs(s[r]));
      if (m <= 10.0 * (a.ctl ? a.ctl->tol_l : 1e-11 * a.dual_scale) + 1e-13 * fmax(1.0, ns0)) case_new = case_old;
    }
    decided = case_new;
    if (case_new == 0) {
#pragma unroll
      for (int r = 0; r < Q1; r++) zn[r] = 0.0;
    }
    changed = case_new != case_old;
    if (changed) rec[11] += 1.0;  // case changes of this cone in the current attempt (diagnostic)
    // open: the cone's own Newton iteration has not converged (the rest of the system is linear given the cone terms)
    double ns = 0.0, nz = 0.0, ds = 0.0, dz = 0.0;
#pragma unroll
    for (int r = 0; r < Q1; r++) {
      ns = fmax(ns, fabs(s[r])); nz = fmax(nz, fabs(zn[r]));
      ds = fmax(ds, fabs(s[r] - sb_o[r])); dz = fmax(dz, fabs(zn[r] - z[r]));
    }
    if (!changed && case_new == 1) {
      const double lam = 0.5 * (w0 + nb);  // Proj_K(w) = lam (1, wh)
      double phi = fabs(s[0] - lam);
#pragma unroll
      for (int r = 0; r < Q; r++) phi = fmax(phi, fabs(s[r + 1] - lam * wh_n[r]));
      open = (ds > a.tol_step * fmax(1.0, ns) || dz > a.tol_step * fmax(a.dual_scale, nz) || phi > a.tol_phi * fmax(1.0, ns)) ? 1 : 0;
    } else if (!changed && case_new == 2) {
      open = (ns > a.tol_phi || dz > a.tol_step * fmax(a.dual_scale, nz)) ? 1 : 0;
    }
#pragma unroll
    for (int r = 0; r < Q1; r++) { z[r] = zn[r]; bad |= !(zn[r] == zn[r]); }
#pragma unroll
    for (int r = 0; r < Q1; r++) a.z[idx * Q1 + r] = zn[r];
  }

  // ---- prepare the next round: Newton terms at (u, z) ---------------------------------------------------------------------------
  double wh[Q], w0, nb;
  int cs = classify(s, z, wh, w0, nb);
  if (decided >= 0 && !(decided == 1 && !(nb > 0.0))) cs = decided;  // (the case the hysteresis kept)
  // a cone that was off and stays off (most of them, most rounds): its terms and record are zero already, nothing counts, nothing
  // to write — 280 of the ~600 bytes this pass moves per cone, and the strided read of R's diagonal
  if (a.finish && cs == 0 && (int)rec[0] == 0 && !bad) return;
  double tr = 0.0;
#pragma unroll
  for (int k = 0; k < UD; k++) tr += a.r32 ? (double)((const float *)a.R)[idx * UD * UD + k * (UD + 1)] : a.R[idx * UD * UD + k * (UD + 1)];
  const double rho = a.rho_scale * (tr / UD + a.reg_u);
  double H[UD][UD], g[UD];
#pragma unroll
  for (int p = 0; p < UD; p++) {
    g[p] = 0.0;
#pragma unroll
    for (int q = 0; q < UD; q++) H[p][q] = 0.0;
  }
  double curv = 0.0, nu_hat = 0.0;
  if (cs == 1) {
    const double rs2 = 0.70710678118654752440;
    const double theta = fmin(fmax(0.5 * (1.0 + w0 / nb), 1e-12), 1.0);  // (clamped: the hysteresis may keep this case a hair outside its region)
    curv = (1.0 - theta) / theta;
    double am[UD], es = s[0], whs = 0.0;
    nu_hat = z[0];
#pragma unroll
    for (int r = 0; r < Q; r++) { es = fma(-wh[r], s[r + 1], es); nu_hat = fma(-wh[r], z[r + 1], nu_hat); whs = fma(wh[r], s[r + 1], whs); }
    es *= rs2; nu_hat *= rs2;
#pragma unroll
    for (int k = 0; k < UD; k++) {
      double v = A[0][k];
#pragma unroll
      for (int r = 0; r < Q; r++) v = fma(-wh[r], A[r + 1][k], v);
      am[k] = v * rs2;
    }
    // tangential: T = (I - wh wh') Ab  (Q x UD);  H += curv T'T (the projector is idempotent), g += curv T' sb
    double T[Q][UD];
#pragma unroll
    for (int k = 0; k < UD; k++) {
      double wa = 0.0;
#pragma unroll
      for (int r = 0; r < Q; r++) wa = fma(wh[r], A[r + 1][k], wa);
#pragma unroll
      for (int r = 0; r < Q; r++) T[r][k] = A[r + 1][k] - wh[r] * wa;
    }
    const double gm = -nu_hat + rho * es;
#pragma unroll
    for (int p = 0; p < UD; p++) {
      double gt = 0.0;
#pragma unroll
      for (int r = 0; r < Q; r++) gt = fma(T[r][p], s[r + 1], gt);  // T'sb = Ab'(I - wh wh') sb
      g[p] = curv * gt + gm * am[p];
#pragma unroll
      for (int q = 0; q < UD; q++) {
        double tt = 0.0;
#pragma unroll
        for (int r = 0; r < Q; r++) tt = fma(T[r][p], T[r][q], tt);
        H[p][q] = curv * tt + rho * am[p] * am[q];
      }
    }
  } else if (cs == 2) {
#pragma unroll
    for (int p = 0; p < UD; p++) {
      double gv = 0.0;
#pragma unroll
      for (int r = 0; r < Q1; r++) gv = fma(A[r][p], -z[r] + rho * s[r], gv);
      g[p] = gv;
#pragma unroll
      for (int q = 0; q < UD; q++) {
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < Q1; r++) t = fma(A[r][p], A[r][q], t);
        H[p][q] = rho * t;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < UD; q++)
#pragma unroll
    for (int p = 0; p < UD; p++) a.H[idx * UD * UD + q * UD + p] = H[p][q];
#pragma unroll
  for (int p = 0; p < UD; p++) a.g[idx * UD + p] = g[p];
  rec[0] = (double)cs; rec[1] = rho; rec[2] = curv; rec[3] = nu_hat;
  if (!a.finish) rec[11] = 0.0;
#pragma unroll
  for (int r = 0; r < Q; r++) rec[4 + r] = wh[r];
#pragma unroll
  for (int r = 0; r < Q1; r++) rec[4 + Q + r] = s[r];

  // ---- counters (a shared control's cone is ONE cone: counted by the owner's particle 0) ----------------------------------------
  const bool counts = j >= a.Nc || (i == 0 && a.owner);
  if (a.finish && counts) {
    if (changed) atomicAdd(&a.cnt[3 * i + 1], 1);
    if (open) atomicAdd(&a.open[i], 1);
    if (changed || open) a.settled[i] = 0;
    if ((changed || open) && a.jhi) atomicMax(&a.jhi[i], j);
    if (bad) a.cnt[3 * i + 2] = 1;
  }
}

// General form: `ncones` cones per (particle, stage), cone k with qs[k] + 1 rows (qs[k] = 0: a linear row s >= 0, i.e. the cone R+,
// whose projection has the interior and the polar case only), rows stacked in A (rows x UD) / c (rows); per_stage: every
// (particle, stage) has its own (A, c).  Same arithmetic as k_cone_step, cone by cone; the Newton terms add up.
template <int UD>
__global__ void __launch_bounds__(256) k_cone_step_multi(ConeArgs a) {
  if (a.done && *a.done) return;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long nrows = (long long)a.M * a.N;
  if (idx >= nrows) return;
  const int i = (int)(idx / a.N), j = (int)(idx - (long long)i * a.N);
  constexpr int QM = 3, Q1M = QM + 1;
  const int R = a.rows;
  // shared controls carry ONE set of cones: particle 0's data on the consensus stages
  const long long didx = a.per_stage ? ((j < a.Nc) ? (long long)j : idx) : 0;
  const double *Ag = a.A + didx * R * UD, *cg = a.c + didx * R;
  double u[UD], ur[UD];
#pragma unroll
  for (int k = 0; k < UD; k++) { u[k] = a.U[idx * UD + k]; ur[k] = a.finish ? a.Uraw[idx * UD + k] : 0.0; }
  double tr = 0.0;
#pragma unroll
  for (int k = 0; k < UD; k++) tr += a.r32 ? (double)((const float *)a.R)[idx * UD * UD + k * (UD + 1)] : a.R[idx * UD * UD + k * (UD + 1)];
  const double rho = a.rho_scale * (tr / UD + a.reg_u);
  double H[UD][UD], g[UD];
#pragma unroll
  for (int p = 0; p < UD; p++) {
    g[p] = 0.0;
#pragma unroll
    for (int q = 0; q < UD; q++) H[p][q] = 0.0;
  }
  int changed = 0, open = 0, row0 = 0;
  bool bad = false;
  const double rs2 = 0.70710678118654752440;
  for (int kc = 0; kc < a.ncones; kc++) {
    const int Q = a.qs[kc], Q1 = Q + 1;
    double A[Q1M][UD], cc[Q1M], s[Q1M], z[Q1M];
#pragma unroll
    for (int r = 0; r < Q1M; r++) {
      const bool rv = r < Q1;
      cc[r] = rv ? cg[row0 + (rv ? r : 0)] : 0.0;
#pragma unroll
      for (int k = 0; k < UD; k++) A[r][k] = rv ? Ag[(row0 + (rv ? r : 0)) * UD + k] : 0.0;
      double v = cc[r];
#pragma unroll
      for (int k = 0; k < UD; k++) v = fma(A[r][k], u[k], v);
      s[r] = v;
      z[r] = rv ? a.z[idx * R + row0 + (rv ? r : 0)] : 0.0;
      bad |= !(v == v);
    }
    auto classify = [&](const double *sv, const double *zv, double *wh, double &w0, double &nb) -> int {
      double wb[QM], n2 = 0.0;
      w0 = sv[0] - zv[0];
#pragma unroll
      for (int r = 0; r < QM; r++) { wb[r] = r < Q ? sv[r + 1] - zv[r + 1] : 0.0; n2 = fma(wb[r], wb[r], n2); }
      nb = sqrt(n2);
      const double inv = nb > 0.0 ? 1.0 / nb : 0.0;
#pragma unroll
      for (int r = 0; r < QM; r++) wh[r] = wb[r] * inv;
      return nb <= w0 ? 0 : (nb <= -w0 ? 2 : 1);
    };
    double *rec = a.rec + (idx * a.ncones + kc) * PMPC_CONE_REC;
    int decided = -1;
    if (a.finish) {
      const int case_old = (int)rec[0];
      const double rho_o = rec[1], curv = rec[2], nu_hat = rec[3];
      double wh_o[QM], sb_o[Q1M], sr[Q1M], zn[Q1M];
#pragma unroll
      for (int r = 0; r < QM; r++) wh_o[r] = r < Q ? rec[4 + (r < Q ? r : 0)] : 0.0;
#pragma unroll
      for (int r = 0; r < Q1M; r++) {
        sb_o[r] = r < Q1 ? rec[4 + Q + (r < Q1 ? r : 0)] : 0.0;
        double v = cc[r];
#pragma unroll
        for (int k = 0; k < UD; k++) v = fma(A[r][k], ur[k], v);
        sr[r] = v;
      }
      if (case_old == 1) {
        double es = sr[0], whs = 0.0;
#pragma unroll
        for (int r = 0; r < QM; r++) { es = fma(-wh_o[r], sr[r + 1], es); whs = fma(wh_o[r], sr[r + 1], whs); }
        es *= rs2;
        const double nu = nu_hat - rho_o * es;
        zn[0] = nu * rs2;
#pragma unroll
        for (int r = 0; r < QM; r++) zn[r + 1] = r < Q ? -nu * rs2 * wh_o[r] - curv * (sr[r + 1] - whs * wh_o[r]) : 0.0;
      } else if (case_old == 2) {
#pragma unroll
        for (int r = 0; r < Q1M; r++) zn[r] = r < Q1 ? z[r] - rho_o * sr[r] : 0.0;
      } else {
#pragma unroll
        for (int r = 0; r < Q1M; r++) zn[r] = 0.0;
      }
      double wh_n[QM], w0, nb;
      int case_new = classify(s, zn, wh_n, w0, nb);
      if (case_new != case_old) {  // hysteresis, see k_cone_step
        const double m = case_new == 0 ? w0 - nb : (case_new == 2 ? -w0 - nb : fmin(nb - w0, nb + w0));
        double ns0 = 0.0;
#pragma unroll
        for (int r = 0; r < Q1M; r++) ns0 = fmax(ns0, fabs(s[r]));
        if (m <= 10.0 * (a.ctl ? a.ctl->tol_l : 1e-11 * a.dual_scale) + 1e-13 * fmax(1.0, ns0)) case_new = case_old;
      }
      decided = case_new;
      if (case_new == 0) {
#pragma unroll
        for (int r = 0; r < Q1M; r++) zn[r] = 0.0;
      }
      const int ch = case_new != case_old;
      changed += ch;
      double ns = 0.0, nz = 0.0, ds = 0.0, dz = 0.0;
#pragma unroll
      for (int r = 0; r < Q1M; r++) {
        ns = fmax(ns, fabs(s[r])); nz = fmax(nz, fabs(zn[r]));
        ds = fmax(ds, fabs(s[r] - sb_o[r])); dz = fmax(dz, fabs(zn[r] - z[r]));
      }
      if (!ch && case_new == 1) {
        const double lam = 0.5 * (w0 + nb);
        double phi = fabs(s[0] - lam);
#pragma unroll
        for (int r = 0; r < QM; r++) phi = fmax(phi, r < Q ? fabs(s[r + 1] - lam * wh_n[r]) : 0.0);
        open += (ds > a.tol_step * fmax(1.0, ns) || dz > a.tol_step * fmax(a.dual_scale, nz) || phi > a.tol_phi * fmax(1.0, ns)) ? 1 : 0;
      } else if (!ch && case_new == 2) {
        open += (ns > a.tol_phi || dz > a.tol_step * fmax(a.dual_scale, nz)) ? 1 : 0;
      }
#pragma unroll
      for (int r = 0; r < Q1M; r++) {
        z[r] = zn[r];
        bad |= !(zn[r] == zn[r]);
        if (r < Q1) a.z[idx * R + row0 + r] = zn[r];
      }
    }
    // ---- Newton terms of this cone at (u, z) -------------------------------------------------------------------------------------
    double wh[QM], w0, nb;
    int cs = classify(s, z, wh, w0, nb);
    if (decided >= 0 && !(decided == 1 && !(nb > 0.0))) cs = decided;
    double curv = 0.0, nu_hat = 0.0;
    if (cs == 1) {
      const double theta = fmin(fmax(0.5 * (1.0 + w0 / nb), 1e-12), 1.0);
      curv = (1.0 - theta) / theta;
      double am[UD], es = s[0], whs = 0.0;
      nu_hat = z[0];
#pragma unroll
      for (int r = 0; r < QM; r++) { es = fma(-wh[r], s[r + 1], es); nu_hat = fma(-wh[r], z[r + 1], nu_hat); whs = fma(wh[r], s[r + 1], whs); }
      es *= rs2; nu_hat *= rs2;
      double T[QM][UD];
#pragma unroll
      for (int k = 0; k < UD; k++) {
        double v = A[0][k], wa = 0.0;
#pragma unroll
        for (int r = 0; r < QM; r++) { v = fma(-wh[r], A[r + 1][k], v); wa = fma(wh[r], A[r + 1][k], wa); }
        am[k] = v * rs2;
#pragma unroll
        for (int r = 0; r < QM; r++) T[r][k] = r < Q ? A[r + 1][k] - wh[r] * wa : 0.0;
      }
      const double gm = -nu_hat + rho * es;
#pragma unroll
      for (int p = 0; p < UD; p++) {
        double gt = 0.0;
#pragma unroll
        for (int r = 0; r < QM; r++) gt = fma(T[r][p], s[r + 1], gt);
        g[p] += curv * gt + gm * am[p];
#pragma unroll
        for (int q = 0; q < UD; q++) {
          double tt = 0.0;
#pragma unroll
          for (int r = 0; r < QM; r++) tt = fma(T[r][p], T[r][q], tt);
          H[p][q] += curv * tt + rho * am[p] * am[q];
        }
      }
    } else if (cs == 2) {
#pragma unroll
      for (int p = 0; p < UD; p++) {
        double gv = 0.0;
#pragma unroll
        for (int r = 0; r < Q1M; r++) gv = fma(A[r][p], -z[r] + rho * s[r], gv);  // (rows beyond the cone are zero)
        g[p] += gv;
#pragma unroll
        for (int q = 0; q < UD; q++) {
          double t = 0.0;
#pragma unroll
          for (int r = 0; r < Q1M; r++) t = fma(A[r][p], A[r][q], t);
          H[p][q] += rho * t;
        }
      }
    }
    rec[0] = (double)cs; rec[1] = rho; rec[2] = curv; rec[3] = nu_hat;
    for (int r = 0; r < Q; r++) rec[4 + r] = wh[r];
    for (int r = 0; r < Q1; r++) rec[4 + Q + r] = s[r];
    row0 += Q1;
  }
#pragma unroll
  for (int q = 0; q < UD; q++)
#pragma unroll
    for (int p = 0; p < UD; p++) a.H[idx * UD * UD + q * UD + p] = H[p][q];
#pragma unroll
  for (int p = 0; p < UD; p++) a.g[idx * UD + p] = g[p];
  const bool counts = j >= a.Nc || (i == 0 && a.owner);
  if (a.finish && counts) {
    if (changed) atomicAdd(&a.cnt[3 * i + 1], changed);
    if (open) atomicAdd(&a.open[i], open);
    if (changed || open) a.settled[i] = 0;
    if ((changed || open) && a.jhi) atomicMax(&a.jhi[i], j);
    if (bad) a.cnt[3 * i + 2] = 1;
  }
}

// a lower box side that the cone's s0 >= 0 implies (v has ONE nonzero v_k > 0 and lo_k <= -v0 / v_k: thrust >= 0 next to the
// thrust cone) is dropped: at the apex the cone holds that control, and a box active on top of it would leave the multipliers
// without a unique split — the active-set rule would flip between the two for ever
__global__ void k_cone_drop_lo(double *lo, const double *A, const double *c, long long rows, int u) {
  int k = -1, nz = 0;
  for (int t = 0; t < u; t++)
    if (A[t] != 0.0) { k = t; nz++; }
  if (nz != 1 || !(A[k] > 0.0)) return;
  const double thr = -c[0] / A[k];
  for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (long long)gridDim.x * blockDim.x)
    if (lo[r * u + k] <= thr + 1e-15 * fmax(1.0, fabs(thr))) lo[r * u + k] = -INFINITY;
}

}  // namespace

bool cone_as_supported(int u, int q) { return u >= 2 && u <= 4 && q >= 0 && q <= 3; }

void launch_cone_step(const ConeArgs &a, hipStream_t s) {
  const long long rows = (long long)a.M * a.N;
  const dim3 grd((unsigned)((rows + 255) / 256)), blk(256);
  static const bool force_general = getenv("PMPC_CONE_GENERAL_KERNEL") && atoi(getenv("PMPC_CONE_GENERAL_KERNEL")) != 0;  // (tests: the general kernel on the single-cone cases)
  if (force_general || a.ncones != 1 || a.per_stage || a.qs[0] < 1) {  // the general form
    if (a.u == 4) hipLaunchKernelGGL((k_cone_step_multi<4>), grd, blk, 0, s, a);
    else if (a.u == 3) hipLaunchKernelGGL((k_cone_step_multi<3>), grd, blk, 0, s, a);
    else if (a.u == 2) hipLaunchKernelGGL((k_cone_step_multi<2>), grd, blk, 0, s, a);
    else abort();
    return;
  }
#define PMPC_CONE(UD, Q) if (a.u == UD && a.qs[0] == Q) { hipLaunchKernelGGL((k_cone_step<UD, Q>), grd, blk, 0, s, a); return; }
  PMPC_CONE(4, 2) PMPC_CONE(4, 1) PMPC_CONE(4, 3) PMPC_CONE(3, 2) PMPC_CONE(3, 1) PMPC_CONE(3, 3) PMPC_CONE(2, 1) PMPC_CONE(2, 2) PMPC_CONE(2, 3)
#undef PMPC_CONE
  abort();
}

void launch_cone_drop_redundant_lo(double *lo, const double *A, const double *c, int q, long long rows, int u, hipStream_t s) {
  (void)q;
  long long b = (rows + 255) / 256;
  if (b > 1024) b = 1024;
  hipLaunchKernelGGL(k_cone_drop_lo, dim3((unsigned)b), dim3(256), 0, s, lo, A, c, rows, u);
}
