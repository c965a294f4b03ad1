// solver_cone.hip — the cone objective of c_lcone_solve (PMPC.jl/src/main.jl:194-354) on the device solver: particle costs, hard boxes through the
// epigraph problem in the shared-control space, smoothed boxes (log barrier / squareplus) through the full-space Newton iteration, the
// free-particles path, the rank-based weighted-QP iteration as fallback.  Split out of solver.hip (r05); shared declarations: solver_internal.h.
#include "solver_internal.h"

extern "C" {

int pmpc_particle_costs_device(pmpc_ctx *c, const pmpc_problem *p0, const double *X, const double *U, double *J_out) {
  HIP_CHECK(hipSetDevice(c->device));
  Workspace &w = c->ws;
  // (the cost kernel reads Q, R as doubles: an fp32-storage problem's blocks are widened first — found by `bench.py --cone --fp32`,
  //  which read the float arrays as doubles, past their end)
  pmpc_problem pw_;
  const pmpc_problem *p = p0;
  if (p0->flags & PMPC_F32_MATRICES) {
    pw_ = widened_f32_problem(c, p0, /*jacobians=*/false);
    p = &pw_;
  }
  const size_t M = p->M, u = p->udim, D8 = sizeof(double);
  LQArgs a;
  memset(&a, 0, sizeof(a));
  a.x = (int)p->xdim; a.u = (int)u; a.N = (int)p->N; a.M = (int)M;
  a.reg_x = p->reg_x; a.reg_u = p->reg_u;
  a.Q = p->Q; a.R = p->R; a.X_prev = p->X_prev; a.U_prev = p->U_prev; a.X_ref = p->X_ref; a.U_ref = p->U_ref;
  const bool has_slew = p->flags & PMPC_HAS_SLEW, has_slew0 = p->flags & PMPC_HAS_SLEW0;
  if (w.zslew.bytes < M * D8 || w.zum1.bytes < M * u * D8) {
    w.zslew.ensure(M * D8); w.zslew0.ensure(M * D8); w.zum1.ensure(M * u * D8);
    HIP_CHECK(hipMemsetAsync(w.zslew.p, 0, M * D8, c->stream));
    HIP_CHECK(hipMemsetAsync(w.zslew0.p, 0, M * D8, c->stream));
    HIP_CHECK(hipMemsetAsync(w.zum1.p, 0, M * u * D8, c->stream));
  }
  a.slew = has_slew ? p->slew_reg : w.zslew.d();
  a.slew0 = has_slew0 ? p->slew_reg0 : w.zslew0.d();
  a.um1 = has_slew0 ? p->slew_um1 : w.zum1.d();
  launch_particle_cost(a, X, U, J_out, c->stream);
  return 0;
}

int pmpc_epigraph_solve_host(int M, int nc, const double *J, const double *H, const double *g, const unsigned char *held, double K, double cap,
                             double *lam_io, double *delta, double *t_out, int verbose);  // epigraph_host.hip
int pmpc_lcone_solve_device(pmpc_ctx *c, const pmpc_problem *p, double smooth_alpha, pmpc_info *info, int verbose) {
  try {
    return lcone_body(c, p, smooth_alpha, info, verbose);
  } catch (const PmpcHipError &) {
    return fail_after_error(c, p, info);
  }
}

// -------------------------------------------------------------------------------------------------
// cone objective WITH log-barrier smoothing (main.jl:246-262): proximal method of multipliers on the epigraph rows, damped Newton in
// the full space (kernels_epi.hip).  Any number of particles on the threshold — with the barrier a particle's optimum given the shared
// controls depends on its multiplier, so every particle whose cost range brackets the threshold carries a fractional multiplier.
//   F(z, t; lam) = (1 - eps) k t + sum_i psi(J_i(z_i) - t; lam_i) - mu_b sum log(slack),  psi(v; l) = max over m in [0, 1 + eps] of  m v - rho/2 (m - l)^2
// Newton step: per particle the Hessian  m_i grad^2 J_i + grad^2 B_i  (one Riccati factor sweep with cost weight m_i and the barrier
// diagonals) plus, for a row with m_i strictly inside, the rank-one term (1/rho) (grad J_i; -1)(grad J_i; -1)' coupling its variables
// to t: a second sweep with the right-hand side grad J_i gives v_i = K^-1 grad J_i, its condensed gradient and kappa_i = grad J_i' v_i,
// and Sherman-Morrison folds the term into the (Nc u + 1)-dimensional system of the shared controls and t, assembled on the host from
// per-particle scalars.  Returns -1 when the problem is outside what this path covers (the caller takes the weighted-QP iteration).
// -------------------------------------------------------------------------------------------------
static int lcone_smooth_body(pmpc_ctx *c, const pmpc_problem *p, double mu_b, pmpc_info *info, int verbose, int smode = 0, double sbeta = 1.0, bool phase1_start = false) {
  Workspace &w = c->ws;
  hipStream_t s = c->stream;
  // Sharded particles (equal contiguous blocks, as everywhere): the device work is local — every rank sweeps its own particles —, the host
  // side of the iteration is GLOBAL and identical on every rank: the per-particle scalars, condensed blocks and costs are gathered
  // (all-reduce(sum) of a zero-padded table, one per Newton step + one per trial point), so every rank assembles the same (Nc u + 1)
  // system, finds the same threshold and takes the same decisions.  Ml: particles here, M: particles in all.
  const int x = (int)p->xdim, u = (int)p->udim, N = (int)p->N, Ml = (int)p->M, world = c->multi() ? c->world : 1, M = Ml * world;
  const size_t off = (size_t)(c->multi() ? c->rank : 0) * Ml;
  const int Nc = p->Nc < 0 ? N : (int)std::min<long long>(p->Nc, (long long)N), nc = Nc * u;
  const bool has_xb = p->flags & PMPC_HAS_XBOUNDS, has_ub = p->flags & PMPC_HAS_UBOUNDS;
  // (several consensus stages: the condensed Hessians M (Nc u)^2 are gathered to the host every Newton step — bounded)
  // (M = 1: the epigraph row is degenerate — its multiplier is (1 - eps) k whatever t does — and the iteration is plain damped Newton on
  //  (1 - eps) J + the smoothing terms; taken for the squareplus hinge, which the interior-point iteration of the QP path does not know)
  if (p->weights || (p->flags & (PMPC_HAS_SLEW | PMPC_HAS_SLEW0 | PMPC_FORCE_GENERIC | PMPC_F32_MATRICES)) || M < (smode == 1 ? 1 : 2) ||
      !(has_xb || has_ub) || !(mu_b > 0.0) || (double)M * nc * nc > 2e7 || (size_t)3 * N * (x + u) * sizeof(double) > 56 * 1024 /* k_cost_dots keeps a particle's vectors in LDS */)
    return -1;
  const size_t nx = (size_t)Ml * N * x, nu = (size_t)Ml * N * u, D8 = sizeof(double);
  LQArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.u = u; a.N = N; a.M = Ml; a.Nc = Nc; a.w = 0; a.n = x;
  a.reg_x = p->reg_x; a.reg_u = p->reg_u;
  a.f = p->f; a.fx = p->fx; a.fu = p->fu; a.Q = p->Q; a.R = p->R;
  a.X_prev = p->X_prev; a.U_prev = p->U_prev; a.X_ref = p->X_ref; a.U_ref = p->U_ref;
  a.owner = (!c->multi() || c->rank == 0) ? 1 : 0; a.any_slew = 0; a.sym_cost = (p->flags & PMPC_SYMMETRIC_COST) ? 1 : 0;
  if (!lq_fast_supported(a)) return -1;
  const double eps = 1e-3, cap = 1.0 + eps;
  const double kk = (p->cone_k > 0 && p->cone_k < (long long)M) ? (double)p->cone_k : (double)M, K = (1.0 - eps) * kk;
  // workspace
  w.X.ensure(nx * D8); w.U.ensure(nu * D8); w.dX.ensure(nx * D8); w.dU.ensure(nu * D8); w.dX2.ensure(nx * D8); w.dU2.ensure(nu * D8);
  w.xm.ensure(nx * D8); w.xd.ensure(nx * D8); w.um.ensure(nu * D8); w.ud.ensure(nu * D8);
  w.es_xm.ensure(nx * D8); w.es_xd.ensure(nx * D8); w.es_um.ensure(nu * D8); w.es_ud.ensure(nu * D8);
  w.es_Dx.ensure(nx * D8); w.es_wx.ensure(nx * D8); w.es_Du.ensure(nu * D8); w.es_wu.ensure(nu * D8);
  w.es_Xt.ensure(nx * D8); w.es_Ut.ensure(nu * D8);
  w.K.ensure((size_t)Ml * N * 64 * D8); w.Hinv.ensure(nu * u * D8);
  w.kff.ensure(nu * D8); w.es_kff2.ensure(nu * D8); w.es_kff3.ensure(nu * D8);
  w.gc_part.ensure((size_t)Ml * std::max(nc, 1) * D8); w.es_gc2.ensure((size_t)Ml * std::max(nc, 1) * D8); w.Hc_part.ensure((size_t)Ml * std::max(nc * nc, 1) * D8);
  w.scratch.ensure((size_t)Ml * 3 * x * std::max(nc, 1) * D8);
  w.es_dots.ensure((size_t)3 * Ml * D8); w.pw.ensure((size_t)Ml * D8); w.Jc.ensure((size_t)Ml * D8);
  w.es_coef.ensure((size_t)2 * Ml * D8);  // [coefficients | sig (input of k_epi_newton)]
  w.es_out2.ensure(4 * D8);  // {step size | barrier value, smallest slack} + {barrier value, smallest slack} of the first trial point
  // gather table of the sharded runs: [H_i | g_i (b) | g_i (a) | dots | one flag per rank] for ALL particles
  const size_t nH = (size_t)nc * nc, tab_H = 0, tab_gb = tab_H + (size_t)M * nH, tab_ga = tab_gb + (size_t)M * nc, tab_dots = tab_ga + (size_t)M * nc, tab_rank = tab_dots + (size_t)3 * M,
               tab_tot = tab_rank + (size_t)4 * world;
  if (c->multi()) w.epi_gath.ensure(tab_tot * D8);
  w.part_sum.ensure(2 * PMPC_RED_BLOCKS * D8); w.part_max.ensure(2 * PMPC_RED_BLOCKS * D8);
  w.duc.ensure((size_t)std::max(nc, 1) * D8); w.fail.ensure(sizeof(int));
  if (w.es_zero.ensure((size_t)std::max(64, nc) * D8)) HIP_CHECK(hipMemsetAsync(w.es_zero.p, 0, w.es_zero.bytes, s));
  if (w.zeros.bytes == 0) {
    w.zeros.ensure(64 * D8);
    HIP_CHECK(hipMemsetAsync(w.zeros.p, 0, 64 * D8, s));
  }
  if (w.zslew.bytes < (size_t)Ml * D8 || w.zum1.bytes < (size_t)Ml * u * D8) {
    w.zslew.ensure((size_t)Ml * D8); w.zslew0.ensure((size_t)Ml * D8); w.zum1.ensure((size_t)Ml * u * D8);
    HIP_CHECK(hipMemsetAsync(w.zslew.p, 0, (size_t)Ml * D8, s));
    HIP_CHECK(hipMemsetAsync(w.zslew0.p, 0, (size_t)Ml * D8, s));
    HIP_CHECK(hipMemsetAsync(w.zum1.p, 0, (size_t)Ml * u * D8, s));
  }
  a.slew = w.zslew.d(); a.slew0 = w.zslew0.d(); a.um1 = w.zum1.d();
  a.zeros = w.zeros.d();
  a.K = w.K.d(); a.Hinv = w.Hinv.d(); a.kff = w.kff.d(); a.gc_part = w.gc_part.d(); a.Hc_part = w.Hc_part.d(); a.scratch = w.scratch.d();
  a.duc = w.es_zero.d(); a.dX = w.dX.d(); a.dU = w.dU.d(); a.fail = (int *)w.fail.p; a.X = w.X.d(); a.U = w.U.d();
  a.xm = w.xm.d(); a.xd = w.xd.d(); a.um = w.um.d(); a.ud = w.ud.d();
  a.pw = w.pw.d();
  a.Dx = has_xb ? w.es_Dx.d() : nullptr; a.wx = has_xb ? w.es_wx.d() : nullptr;
  a.Du = has_ub ? w.es_Du.d() : nullptr; a.wu = has_ub ? w.es_wu.d() : nullptr;
  w.as_key = -1;  // (the workspace's factor records and warm-start memories of the QP path are overwritten)
  w.warm_key = -1;
  pmpc_info inf;
  memset(&inf, 0, sizeof(inf));
  inf.fast_path = 1;
  auto finish = [&](int status) {
    inf.status = status;
    if (status == 0) {
      HIP_CHECK(hipMemcpyAsync(p->X_out, w.X.p, nx * D8, hipMemcpyDeviceToDevice, s));
      HIP_CHECK(hipMemcpyAsync(p->U_out, w.U.p, nu * D8, hipMemcpyDeviceToDevice, s));
    } else {
      fill_nan_outputs(c, p);
      w.es_key = -1;
      c->cone_lam_key = -1;
    }
    if (info) *info = inf;
    return status;
  };
  // ---- host pieces ---------------------------------------------------------------------------------------------------------
  std::vector<double> J(M), Jt(M), lam(M, std::log((K / (double)M) / (cap - K / (double)M))), mu(M), sig(M);
  // the arrays the device copies into / out of every Newton step: pinned, kept with the context
  struct Span {
    double *p;
    double &operator[](size_t i) const { return p[i]; }
    double *data() const { return p; }
  };
  const size_t n_H = (size_t)M * std::max(nc * nc, 1), n_g = (size_t)M * std::max(nc, 1), n_pin = n_H + 2 * n_g + (size_t)3 * M + (size_t)M;
  if (c->sm_pinned_bytes < n_pin * D8) {
    if (c->sm_pinned) HIP_WARN(hipHostFree(c->sm_pinned));
    c->sm_pinned = nullptr;
    c->sm_pinned_bytes = 0;
    HIP_CHECK(hipHostMalloc(&c->sm_pinned, n_pin * D8, hipHostMallocDefault));
    c->sm_pinned_bytes = n_pin * D8;
  }
  double *pin = (double *)c->sm_pinned;
  const Span Hh{pin}, gb{pin + n_H}, ga{pin + n_H + n_g}, dots{pin + n_H + 2 * n_g}, coef{pin + n_H + 2 * n_g + (size_t)3 * M};
  double rho = 1.0, out2[2] = {0.0, 0.0};
  std::vector<double> rk((size_t)4 * world);  // per-rank scalars as gathered
  auto combine_out2 = [&]() {  // rk = (barrier value, smallest slack) per rank -> out2, summed / minimised in rank order
    double v = 0.0, m_ = 1e300;
    for (int r = 0; r < world; r++) {
      v += rk[2 * r];
      m_ = (rk[2 * r + 1] < m_ || rk[2 * r + 1] != rk[2 * r + 1]) ? rk[2 * r + 1] : m_;
    }
    out2[0] = v;
    out2[1] = m_;
  };
  // Entropic proximal term (exponential method of multipliers for multipliers boxed in [0, cap]): with l_i = logit(lam_i / cap)
  //   m_i(v) = cap * sigmoid(l_i + v / rho),   psi(v; l_i) = rho cap [softplus(l_i + v / rho) - softplus(l_i)],   psi'' = m (cap - m) / (rho cap) > 0:
  // every row carries a rank-one term, F is smooth (the quadratic proximal term's clipping makes its second derivative jump between 0
  // and 1/rho — at config D the semismooth Newton iteration then flips a handful of rows in and out of the hinge for ever), and the
  // update l_i += v_i / rho sends the multiplier of a row below the threshold to zero geometrically.  `lam` holds the logits.
  // (beyond |w| = 40 the exponential is below the last bit of 1: no libm call.  At config D all but a few dozen of the 4096 rows are
  //  saturated at every evaluation, and the exponentials of solve_t / Fval were more than half of a Newton step's wall time)
  auto sigm = [](double w_) { return w_ > 40.0 ? 1.0 : (w_ < -40.0 ? std::exp(w_) : (w_ >= 0.0 ? 1.0 / (1.0 + std::exp(-w_)) : std::exp(w_) / (1.0 + std::exp(w_)))); };
  auto softplus = [](double w_) { return w_ > 40.0 ? w_ : (w_ < -40.0 ? std::exp(w_) : (w_ > 0.0 ? w_ + std::log1p(std::exp(-w_)) : std::log1p(std::exp(w_)))); };
  auto mult = [&](int i, const std::vector<double> &Jv, double t) { return cap * sigm(lam[i] + (Jv[i] - t) / rho); };
  double t_guess = std::numeric_limits<double>::quiet_NaN();
  auto solve_t = [&](const std::vector<double> &Jv) {  // sum_i m_i(J_i - t) = K  (smooth, strictly decreasing in t): safeguarded Newton
    double lo = 1e300, hi = -1e300;
    for (int i = 0; i < M; i++) { const double c0 = Jv[i] + rho * lam[i]; lo = std::min(lo, c0); hi = std::max(hi, c0); }
    lo -= 50.0 * rho + 1.0; hi += 50.0 * rho + 1.0;
    double tt = (t_guess == t_guess && t_guess > lo && t_guess < hi) ? t_guess : 0.5 * (lo + hi);
    for (int it = 0; it < 200; it++) {
      double sm = 0.0, ds = 0.0;
      for (int i = 0; i < M; i++) {
        const double m = mult(i, Jv, tt);
        sm += m;
        ds += m * (cap - m);
      }
      const double S = sm - K;
      if (S > 0.0) lo = tt; else hi = tt;
      if (std::fabs(S) <= 1e-13 * K || hi - lo <= 1e-15 * std::max(1.0, std::fabs(lo) + std::fabs(hi))) break;
      ds /= rho * cap;  // = -dS/dt
      double tn = ds > 0.0 ? tt + S / ds : 0.5 * (lo + hi);
      if (!(tn > lo && tn < hi)) tn = 0.5 * (lo + hi);
      tt = tn;
    }
    t_guess = tt;
    return tt;
  };
  auto Fval = [&](const std::vector<double> &Jv, double t, double bval) {
    double f = 0.0;
    for (int i = 0; i < M; i++) f += softplus(lam[i] + (Jv[i] - t) / rho) - softplus(lam[i]);
    return K * t + bval + rho * cap * f;
  };
  // barrier terms + particle costs at (Xe, Ue): -> Jv, out2 = {barrier value, smallest slack}
  auto eval_at = [&](const double *Xe, const double *Ue, std::vector<double> &Jv) {
    launch_bar_prep(Xe, Ue, has_xb ? p->lx : nullptr, has_xb ? p->ux : nullptr, has_ub ? p->lu : nullptr, has_ub ? p->uu : nullptr, w.es_Dx.d(), w.es_wx.d(),
                    w.es_Du.d(), w.es_wu.d(), mu_b, (long long)nx, (long long)nu, u, N, Nc, a.owner, w.part_sum.d(), w.part_max.d(), w.es_out2.d(), s, smode, sbeta);
    launch_particle_cost(a, Xe, Ue, w.Jc.d(), s);
    if (!c->multi()) {
      HIP_CHECK(hipMemcpyAsync(out2, w.es_out2.p, 2 * D8, hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipMemcpyAsync(Jv.data(), w.Jc.p, (size_t)M * D8, hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipStreamSynchronize(s));
      return;
    }
    // [J of every particle | (barrier value, smallest slack) of every rank]
    const size_t tot = (size_t)M + 2 * world;
    HIP_CHECK(hipMemsetAsync(w.epi_gath.p, 0, tot * D8, s));
    HIP_CHECK(hipMemcpyAsync(w.epi_gath.d() + off, w.Jc.p, (size_t)Ml * D8, hipMemcpyDeviceToDevice, s));
    HIP_CHECK(hipMemcpyAsync(w.epi_gath.d() + M + 2 * c->rank, w.es_out2.p, 2 * D8, hipMemcpyDeviceToDevice, s));
    allreduce(c, w.epi_gath.p, tot, ncclFloat64, ncclSum);
    HIP_CHECK(hipMemcpyAsync(Jv.data(), w.epi_gath.p, (size_t)M * D8, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipMemcpyAsync(rk.data(), w.epi_gath.d() + M, (size_t)2 * world * D8, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    combine_out2();
  };
  // ---- starting point: the previous smoothed solution of this shape (strictly inside the same boxes), else the caller's U_prev pulled inside ----
  const long long skey = ((((((long long)x * 131 + u) * 131 + N) * 1000003 + M) * 131 + Nc) * 4 + (has_xb ? 2 : 0) + (has_ub ? 1 : 0)) * 2 + smode;
  bool warm = (phase1_start || !(p->flags & PMPC_COLD_START)) && w.es_key == skey && w.es_U.bytes >= nu * D8;
  const bool lam_mem = c->opt[OPT_CONE_RANK_MEMORY] != 0.0 && !(p->flags & PMPC_COLD_START) && c->cone_lam_key == -(skey + 7) && (int)c->cone_lam.size() == M;
  if (lam_mem) lam = c->cone_lam;
  c->cone_lam_key = -1;
  for (int attempt = 0; attempt < 2; attempt++) {
    if (warm) {
      HIP_CHECK(hipMemcpyAsync(w.U.p, w.es_U.p, nu * D8, hipMemcpyDeviceToDevice, s));
    } else {
      HIP_CHECK(hipMemcpyAsync(w.U.p, p->U_prev, nu * D8, hipMemcpyDeviceToDevice, s));
      if (has_ub && smode == 0) launch_interior(w.U.d(), p->lu, p->uu, (long long)nu, 0.05, s);
    }
    if (c->multi() && nc > 0) broadcast(c, w.U.p, (size_t)nc, ncclFloat64, 0);  // (global particle 0's shared controls)
    launch_share_cons(w.U.d(), Ml, N, u, Nc, s);
    launch_rollout_fast(a, w.U.d(), w.X.d(), s);
    eval_at(w.X.d(), w.U.d(), J);
    if (out2[1] > 0.0 && out2[0] == out2[0]) break;
    if (warm) { warm = false; continue; }
    if (has_xb && !phase1_start) {
      // the caller's controls roll out to states outside their boxes (nothing pulls a STATE inside): phase 1 = the plain-sum problem with
      // the same barrier (the library's interior-point iteration starts anywhere); its controls are strictly inside everything
      pmpc_problem q1 = *p;
      q1.weights = nullptr;
      q1.barrier_mu = mu_b;
      q1.flags |= PMPC_COLD_START;
      pmpc_info i1;
      const int st1 = pmpc_lqp_solve_device(c, &q1, &i1, 0);
      if (verbose) printf("pmpc_hip: smoothed cone objective: start outside the state boxes (smallest slack %.3e); phase 1 (barrier QP) status %d\n", out2[1], st1);
      if (st1 == 0) {
        w.es_U.ensure(nu * D8);
        HIP_CHECK(hipMemcpyAsync(w.es_U.p, p->U_out, nu * D8, hipMemcpyDeviceToDevice, s));
        w.es_key = skey;
        return lcone_smooth_body(c, p, mu_b, info, verbose, smode, sbeta, true);  // (the QP solve may have moved workspace buffers: start over)
      }
    }
    if (verbose) printf("pmpc_hip: smoothed cone objective: no strictly feasible start (smallest slack %.3e)\n", out2[1]);
    return finish(1);
  }
  w.es_key = -1;
  if (lam_mem) {
    // remembered logits may belong to ANOTHER problem of this shape.  A multiplier saturated at the wrong end of [0, cap] takes
    // |l| rho / |J - t| updates to come back: where the costs at the start rank a particle clearly on the other side of the threshold
    // than its remembered multiplier says, the logit is pulled back to +-12 (inside an SCP loop the sides agree and nothing changes)
    std::vector<int> idx(M);
    for (int i = 0; i < M; i++) idx[i] = i;
    std::sort(idx.begin(), idx.end(), [&](int a_, int b_) { return J[a_] > J[b_]; });
    const int n_top = (int)(K / cap);  // about this many rows carry the full multiplier
    for (int r = 0; r < M; r++) {
      const int i = idx[r];
      if (lam[i] > 12.0 && r >= n_top + 2) lam[i] = 12.0;
      if (lam[i] < -12.0 && r < n_top - 1) lam[i] = -12.0;
    }
  }
  {
    double jlo = 1e300, jhi = -1e300;
    for (int i = 0; i < M; i++) { jlo = std::min(jlo, J[i]); jhi = std::max(jhi, J[i]); }
    rho = 1e-6 * std::max(1.0, std::max(std::fabs(jlo), std::fabs(jhi)));
  }
  // proximal parameter rho = width of the smoothed hinge in cost units.  The method converges for ANY rho (the multipliers are exact at
  // its fixed point); a narrow hinge makes one multiplier update nearly exact but its inner problem nearly non-smooth — with few rows
  // strictly inside, every Newton step crosses kinks and the line search cuts it to nothing (measured at config D: steps of 1e-4 below
  // rho ~ 1e-3 max|J|) —, a wide one gives inner problems that settle in a few full steps and more multiplier updates.  Adaptive: start
  // wide, narrow by 3 after an inner solve that took full steps, widen by 3 after one that was damped throughout.
  const double rho_min = rho, rho_max = 1e5 * rho;
  rho = lam_mem ? 3.0 * c->cone_rho : 1e4 * rho_min;
  if (!(rho >= rho_min && rho <= rho_max)) rho = 1e4 * rho_min;
  double bval = out2[0], t = solve_t(J), Fcur = Fval(J, t, bval);
  int newton = 0;
  bool converged = false;
  double dl_prev = 1e300, dl_last = 1e300, rho_floor = 0.0;
  std::vector<double> lam_before(M), dlam_prev(M, 0.0), dlam_cur(M, 0.0);
  bool have_prev_delta = false;
  double quad_c = 1e3;  // contraction constant of the Newton steps in their quadratic regime, |step_{n+1}| ~ quad_c |step_n|^2: the largest ratio seen in this solve
  static const bool quad_on = !(getenv("PMPC_SM_QUAD") && atoi(getenv("PMPC_SM_QUAD")) == 0);  // (A/B switch)
  double r_prev = -1.0, rho_at_prev_delta = -1.0;
  int since_extrap = 2;
  for (int outer = 0; outer < 500 && !converged; outer++) {
    bool inner_ok = false;
    const bool last_stage = true;
    const double step_tol = std::max(1e-9, std::min(1e-5, 1e-3 * dl_prev));  // (the inner problems are solved as sharply as the multipliers are known; looser — 1e-1 dl, cap 1e-3 — measured: no fewer Newton steps)  // (the inner problems are solved as sharply as the multipliers are known)
    int n_damped = 0, n_cut = 0;
    double s_prev = -1.0;  // the previous full Newton step of this inner solve
    for (int it = 0; it < 25; it++) {
      t = solve_t(J);
      Fcur = Fval(J, t, bval);
      double summu = 0.0;
      for (int i = 0; i < M; i++) {
        mu[i] = mult(i, J, t);
        sig[i] = mu[i] * (cap - mu[i]) / (rho * cap);
        summu += mu[i];
        coef[i] = std::max(mu[i], 1e-8);  // (cost weight of the sweeps: a particle of multiplier zero keeps a strictly convex sub-problem)
      }
      HIP_CHECK(hipMemcpyAsync(w.pw.p, coef.data() + off, (size_t)Ml * D8, hipMemcpyHostToDevice, s));
      HIP_CHECK(hipMemcpyAsync(w.es_coef.d() + Ml, sig.data() + off, (size_t)Ml * D8, hipMemcpyHostToDevice, s));
      HIP_CHECK(hipMemsetAsync(w.fail.p, 0, sizeof(int), s));
      // right-hand side b: gradient of sum m_i J_i + barrier (the barrier arrays were written by the last eval_at at this point)
      launch_grad_prep(a, s);
      launch_bwd_fast(a, true, s);
      if (Nc > 1) launch_cond_fast(a, s);  // off-diagonal blocks of the condensed Hessians
      // (the particles' own Newton steps p_b = -K^-1 b_i are not formed: all that is needed of them is pi_i = grad J_i . p_b = b_i . (-K^-1 grad J_i),
      //  a dot product with the second solve's direction — one forward sweep less per Newton step)
      // right-hand side a_i = grad J_i (unweighted, no barrier shift), same Hessian
      LQArgs ag = a;
      ag.pw = nullptr; ag.wx = ag.wu = nullptr;
      ag.xm = w.es_xm.d(); ag.xd = w.es_xd.d(); ag.um = w.es_um.d(); ag.ud = w.es_ud.d();
      launch_grad_prep(ag, s);
      LQArgs a2 = a;
      a2.xm = ag.xm; a2.xd = ag.xd; a2.um = ag.um; a2.ud = ag.ud;
      a2.kff = w.es_kff2.d(); a2.gc_part = w.es_gc2.d(); a2.dX = w.dX2.d(); a2.dU = w.dU2.d();
      launch_bwd_fast(a2, true, s);
      launch_fwd_fast(a2, s);  // -> -v_i = -K^-1 grad J_i in dX2 / dU2
      launch_cost_dots(a, w.X.d(), w.U.d(), w.dX2.d(), w.dU2.d(), w.dX2.d(), w.dU2.d(), w.es_dots.d(), s, a.wx, a.wu);  // (at least one of the two exists: the path needs boxes)
      // the (Nc u + 1) system: on the device where it applies (1 <= Nc u <= 8) — no read-back of the condensed blocks, no synchronisation
      // before the direction's sweep; sharded: every rank sums its own particles' terms, ONE all-reduce of those sums (<= 55 doubles,
      // the failure flags among them) instead of the table of every particle's blocks.  PMPC_SM_DEV=0: the host loop, for A/B
      static const bool dev_env = !(getenv("PMPC_SM_DEV") && atoi(getenv("PMPC_SM_DEV")) == 0);
      const bool dev_sys = dev_env && nc >= 1 && nc <= 8;
      int failflag = 0;
      if (dev_sys) {
        if (!c->multi()) {
          launch_epi_newton(w.Hc_part.d(), w.gc_part.d(), w.es_gc2.d(), w.es_dots.d(), w.es_coef.d() + Ml, Ml, nc, K - summu, w.es_coef.d(), w.duc.d(), (int *)w.fail.p, s);
        } else {
          const int n_x = launch_epi_newton(w.Hc_part.d(), w.gc_part.d(), w.es_gc2.d(), w.es_dots.d(), w.es_coef.d() + Ml, Ml, nc, K - summu, w.es_coef.d(), w.duc.d(),
                                            (int *)w.fail.p, s, 1, w.epi_gath.d());
          allreduce(c, w.epi_gath.p, (size_t)n_x, ncclFloat64, ncclSum);
          launch_epi_newton(w.Hc_part.d(), w.gc_part.d(), w.es_gc2.d(), w.es_dots.d(), w.es_coef.d() + Ml, Ml, nc, K - summu, w.es_coef.d(), w.duc.d(), (int *)w.fail.p, s, 2,
                            w.epi_gath.d());
        }
        inf.structured_solves += 2;
      } else {
        if (!c->multi()) {
          if (nc > 0) {
            HIP_CHECK(hipMemcpyAsync(Hh.data(), w.Hc_part.p, (size_t)M * nc * nc * D8, hipMemcpyDeviceToHost, s));
            HIP_CHECK(hipMemcpyAsync(gb.data(), w.gc_part.p, (size_t)M * nc * D8, hipMemcpyDeviceToHost, s));
            HIP_CHECK(hipMemcpyAsync(ga.data(), w.es_gc2.p, (size_t)M * nc * D8, hipMemcpyDeviceToHost, s));
          }
          HIP_CHECK(hipMemcpyAsync(dots.data(), w.es_dots.p, (size_t)3 * M * D8, hipMemcpyDeviceToHost, s));
          HIP_CHECK(hipMemcpyAsync(&failflag, w.fail.p, sizeof(int), hipMemcpyDeviceToHost, s));
          HIP_CHECK(hipStreamSynchronize(s));
        } else {
          // one table, one all-reduce: every rank gets every particle's condensed blocks and scalars (and every rank's failure flag)
          HIP_CHECK(hipMemcpyAsync(&failflag, w.fail.p, sizeof(int), hipMemcpyDeviceToHost, s));
          HIP_CHECK(hipMemsetAsync(w.epi_gath.p, 0, tab_tot * D8, s));
          if (nc > 0) {
            HIP_CHECK(hipMemcpyAsync(w.epi_gath.d() + tab_H + off * nH, w.Hc_part.p, (size_t)Ml * nH * D8, hipMemcpyDeviceToDevice, s));
            HIP_CHECK(hipMemcpyAsync(w.epi_gath.d() + tab_gb + off * nc, w.gc_part.p, (size_t)Ml * nc * D8, hipMemcpyDeviceToDevice, s));
            HIP_CHECK(hipMemcpyAsync(w.epi_gath.d() + tab_ga + off * nc, w.es_gc2.p, (size_t)Ml * nc * D8, hipMemcpyDeviceToDevice, s));
          }
          HIP_CHECK(hipMemcpyAsync(w.epi_gath.d() + tab_dots + 3 * off, w.es_dots.p, (size_t)3 * Ml * D8, hipMemcpyDeviceToDevice, s));
          HIP_CHECK(hipStreamSynchronize(s));  // (the flag is on the host)
          const double ff = (double)failflag;
          HIP_CHECK(hipMemcpyAsync(w.epi_gath.d() + tab_rank + 4 * c->rank, &ff, D8, hipMemcpyHostToDevice, s));
          allreduce(c, w.epi_gath.p, tab_tot, ncclFloat64, ncclSum);
          if (nc > 0) {
            HIP_CHECK(hipMemcpyAsync(Hh.data(), w.epi_gath.d() + tab_H, (size_t)M * nH * D8, hipMemcpyDeviceToHost, s));
            HIP_CHECK(hipMemcpyAsync(gb.data(), w.epi_gath.d() + tab_gb, (size_t)M * nc * D8, hipMemcpyDeviceToHost, s));
            HIP_CHECK(hipMemcpyAsync(ga.data(), w.epi_gath.d() + tab_ga, (size_t)M * nc * D8, hipMemcpyDeviceToHost, s));
          }
          HIP_CHECK(hipMemcpyAsync(dots.data(), w.epi_gath.d() + tab_dots, (size_t)3 * M * D8, hipMemcpyDeviceToHost, s));
          HIP_CHECK(hipMemcpyAsync(rk.data(), w.epi_gath.d() + tab_rank, (size_t)4 * world * D8, hipMemcpyDeviceToHost, s));
          HIP_CHECK(hipStreamSynchronize(s));
          for (int r = 0; r < world; r++) failflag = std::max(failflag, (int)rk[4 * r]);
        }
        inf.structured_solves += 2;
        if (failflag) {
          if (verbose) printf("pmpc_hip: smoothed cone objective: a factor sweep failed (flag %d)\n", failflag);
          return finish(2);
        }
        // (nc + 1) system of the shared controls and t
        const int n1 = nc + 1;
        std::vector<double> A((size_t)n1 * n1, 0.0), rhs(n1, 0.0), sol(n1, 0.0), sp(M), kap(M), pi_(M);
        double cc = 0.0;
        for (int i = 0; i < M; i++) {
          kap[i] = std::max(0.0, -dots[3 * i]);
          pi_[i] = dots[3 * i + 1];
          sp[i] = sig[i] / (1.0 + sig[i] * kap[i]);
          const double *Hi = &Hh[(size_t)i * nc * nc], *gbi = &gb[(size_t)i * nc], *gai = &ga[(size_t)i * nc];
          for (int r = 0; r < nc; r++) {
            for (int q_ = 0; q_ < nc; q_++) A[r + (size_t)n1 * q_] += Hi[(r <= q_ ? r : q_) + (size_t)nc * (r <= q_ ? q_ : r)] + sp[i] * gai[r] * gai[q_];
            A[r + (size_t)n1 * nc] -= sp[i] * gai[r];
            A[nc + (size_t)n1 * r] -= sp[i] * gai[r];
            rhs[r] -= gbi[r] + sp[i] * pi_[i] * gai[r];
          }
          cc += sp[i];
          rhs[nc] += sp[i] * pi_[i];
        }
        rhs[nc] -= K - summu;
        A[nc + (size_t)n1 * nc] = cc > 0.0 ? cc : 1.0;  // (no row strictly inside: t stays — it is re-optimised exactly at the next point)
        if (!(cc > 0.0)) {
          rhs[nc] = 0.0;
          for (int r = 0; r < nc; r++) A[r + (size_t)n1 * nc] = A[nc + (size_t)n1 * r] = 0.0;
        }
        {  // Gaussian elimination with partial pivoting (the matrix is positive definite; n1 = Nc u + 1)
          std::vector<double> Mx = A;
          sol = rhs;
          for (int k = 0; k < n1; k++) {
            int pv = k;
            for (int r = k + 1; r < n1; r++)
              if (std::fabs(Mx[r + (size_t)n1 * k]) > std::fabs(Mx[pv + (size_t)n1 * k])) pv = r;
            if (pv != k) {
              for (int q_ = 0; q_ < n1; q_++) std::swap(Mx[k + (size_t)n1 * q_], Mx[pv + (size_t)n1 * q_]);
              std::swap(sol[k], sol[pv]);
            }
            const double d = Mx[k + (size_t)n1 * k];
            if (!(std::fabs(d) > 0.0)) return finish(2);
            for (int r = k + 1; r < n1; r++) {
              const double fct = Mx[r + (size_t)n1 * k] / d;
              for (int q_ = k; q_ < n1; q_++) Mx[r + (size_t)n1 * q_] -= fct * Mx[k + (size_t)n1 * q_];
              sol[r] -= fct * sol[k];
            }
          }
          for (int k = n1 - 1; k >= 0; k--) {
            double v = sol[k];
            for (int q_ = k + 1; q_ < n1; q_++) v -= Mx[k + (size_t)n1 * q_] * sol[q_];
            sol[k] = v / Mx[k + (size_t)n1 * k];
          }
        }
        const double dt = sol[nc];
        for (int i = 0; i < M; i++) {
          double e_ = pi_[i] - dt;
          for (int r = 0; r < nc; r++) e_ += ga[(size_t)i * nc + r] * sol[r];
          coef[i] = sig[i] * e_ / (1.0 + sig[i] * kap[i]);
        }
        // total direction: feed-forward k_b + c_i k_a, shared step du_c
        HIP_CHECK(hipMemcpyAsync(w.es_coef.p, coef.data() + off, (size_t)Ml * D8, hipMemcpyHostToDevice, s));
        if (nc > 0) HIP_CHECK(hipMemcpyAsync(w.duc.p, sol.data(), (size_t)nc * D8, hipMemcpyHostToDevice, s));
      }
      launch_axpy_particle(w.kff.d(), w.es_kff2.d(), w.es_coef.d(), w.es_kff3.d(), (long long)N * u, (long long)nu, s);
      LQArgs a3 = a;
      a3.kff = w.es_kff3.d(); a3.duc = nc > 0 ? w.duc.d() : w.es_zero.d();
      launch_fwd_fast(a3, s);  // -> direction in dX / dU
      // along the step the particle costs are EXACT quadratics in the step length: J_i(al) = J_i + al s_i + al^2/2 q_i with s_i = grad J_i . d,
      // q_i = d' hess J_i d (one pass); only the barrier has to be evaluated at the trial points
      launch_cost_dots(a, w.X.d(), w.U.d(), w.dX.d(), w.dU.d(), w.dX.d(), w.dU.d(), w.es_dots.d(), s);
      launch_step_to(w.X.d(), w.dX.d(), 1.0, w.es_Xt.d(), (long long)nx, s);
      launch_step_to(w.U.d(), w.dU.d(), 1.0, w.es_Ut.d(), (long long)nu, s);
      launch_scp_residual(w.es_Xt.d(), w.X.d(), w.es_Ut.d(), w.U.d(), (long long)Ml * N, x, u, w.es_out2.d(), s, true);
      // (the barrier at the FULL step — the line search's first trial, nearly always the accepted one — rides in the same read-back)
      launch_bar_prep(w.es_Xt.d(), w.es_Ut.d(), has_xb ? p->lx : nullptr, has_xb ? p->ux : nullptr, has_ub ? p->lu : nullptr, has_ub ? p->uu : nullptr, w.es_Dx.d(), w.es_wx.d(),
                      w.es_Du.d(), w.es_wu.d(), mu_b, (long long)nx, (long long)nu, u, N, Nc, a.owner, w.part_sum.d(), w.part_max.d(), w.es_out2.d() + 2, s, smode, sbeta);
      double stepmax = 0.0, out_first[2] = {0.0, 0.0};
      if (!c->multi()) {
        HIP_CHECK(hipMemcpyAsync(dots.data(), w.es_dots.p, (size_t)3 * M * D8, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipMemcpyAsync(&stepmax, w.es_out2.p, D8, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipMemcpyAsync(out_first, w.es_out2.d() + 2, 2 * D8, hipMemcpyDeviceToHost, s));
        if (dev_sys) HIP_CHECK(hipMemcpyAsync(&failflag, w.fail.p, sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
      } else {  // [dots of every particle | largest step of every rank | (barrier value, smallest slack) of every rank at the full step]
        const size_t tot = (size_t)3 * M + 3 * world;
        HIP_CHECK(hipMemsetAsync(w.epi_gath.p, 0, tot * D8, s));
        HIP_CHECK(hipMemcpyAsync(w.epi_gath.d() + 3 * off, w.es_dots.p, (size_t)3 * Ml * D8, hipMemcpyDeviceToDevice, s));
        HIP_CHECK(hipMemcpyAsync(w.epi_gath.d() + 3 * M + c->rank, w.es_out2.p, D8, hipMemcpyDeviceToDevice, s));
        HIP_CHECK(hipMemcpyAsync(w.epi_gath.d() + 3 * M + world + 2 * c->rank, w.es_out2.d() + 2, 2 * D8, hipMemcpyDeviceToDevice, s));
        allreduce(c, w.epi_gath.p, tot, ncclFloat64, ncclSum);
        std::vector<double> rs(world);
        HIP_CHECK(hipMemcpyAsync(dots.data(), w.epi_gath.p, (size_t)3 * M * D8, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipMemcpyAsync(rs.data(), w.epi_gath.d() + 3 * M, (size_t)world * D8, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipMemcpyAsync(rk.data(), w.epi_gath.d() + 3 * M + world, (size_t)2 * world * D8, hipMemcpyDeviceToHost, s));
        if (dev_sys) HIP_CHECK(hipMemcpyAsync(&failflag, w.fail.p, sizeof(int), hipMemcpyDeviceToHost, s));  // (the same on every rank: the flags were summed)
        HIP_CHECK(hipStreamSynchronize(s));
        for (int r = 0; r < world; r++) stepmax = (rs[r] > stepmax || rs[r] != rs[r]) ? rs[r] : stepmax;
        combine_out2();
        out_first[0] = out2[0]; out_first[1] = out2[1];
      }
      if (dev_sys && failflag) {
        if (verbose) printf("pmpc_hip: smoothed cone objective: a factor sweep or the Newton system failed (flag %d)\n", failflag);
        return finish(2);
      }
      newton++;
      auto bar_at = [&](const double *Xe, const double *Ue) {  // barrier arrays + value + smallest slack at a point
        launch_bar_prep(Xe, Ue, has_xb ? p->lx : nullptr, has_xb ? p->ux : nullptr, has_ub ? p->lu : nullptr, has_ub ? p->uu : nullptr, w.es_Dx.d(), w.es_wx.d(),
                        w.es_Du.d(), w.es_wu.d(), mu_b, (long long)nx, (long long)nu, u, N, Nc, a.owner, w.part_sum.d(), w.part_max.d(), w.es_out2.d(), s, smode, sbeta);
        if (!c->multi()) {
          HIP_CHECK(hipMemcpyAsync(out2, w.es_out2.p, 2 * D8, hipMemcpyDeviceToHost, s));
          HIP_CHECK(hipStreamSynchronize(s));
          return;
        }
        HIP_CHECK(hipMemsetAsync(w.epi_gath.p, 0, (size_t)2 * world * D8, s));
        HIP_CHECK(hipMemcpyAsync(w.epi_gath.d() + 2 * c->rank, w.es_out2.p, 2 * D8, hipMemcpyDeviceToDevice, s));
        allreduce(c, w.epi_gath.p, (size_t)2 * world, ncclFloat64, ncclSum);
        HIP_CHECK(hipMemcpyAsync(rk.data(), w.epi_gath.p, (size_t)2 * world * D8, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        combine_out2();
      };
      // backtracking: strictly inside the boxes, no increase of F beyond its round-off (t re-optimised at every trial point); a step
      // that is already tiny is taken in full as soon as it is feasible — F cannot resolve it
      const double f_noise = 1e-13 * std::max(1.0, std::fabs(Fcur)) * std::sqrt((double)M);
      double al = 1.0, Ft = 0.0, tt = t, bt = 0.0, al_feas = 0.0;  // al_feas: the first trial step strictly inside the boxes
      bool accepted = false;
      for (int ls = 0; ls < 30; ls++) {
        if (ls > 0) {
          launch_step_to(w.X.d(), w.dX.d(), al, w.es_Xt.d(), (long long)nx, s);
          launch_step_to(w.U.d(), w.dU.d(), al, w.es_Ut.d(), (long long)nu, s);
        }
        if (ls == 0) { out2[0] = out_first[0]; out2[1] = out_first[1]; }
        else bar_at(w.es_Xt.d(), w.es_Ut.d());
        if (out2[1] > 0.0 && out2[0] == out2[0]) {
          if (al_feas == 0.0) al_feas = al;
          for (int i = 0; i < M; i++) Jt[i] = J[i] + al * (dots[3 * i] + 0.5 * al * dots[3 * i + 2]);
          bt = out2[0];
          tt = solve_t(Jt);
          Ft = Fval(Jt, tt, bt);
          if (Ft <= Fcur + f_noise || stepmax * al <= 1e-7) { accepted = true; break; }
        }
        al *= 0.5;
      }
      if (!accepted) {
        bar_at(w.X.d(), w.U.d());  // (restores the barrier arrays of the current point)
        if (verbose) printf("pmpc_hip: smoothed cone objective: line search found no decrease (Newton step %d, step size %.3e)\n", newton, stepmax);
        inner_ok = stepmax <= 1e-6;
        break;
      }
      const double dF = Fcur - Ft;
      HIP_CHECK(hipMemcpyAsync(w.X.p, w.es_Xt.p, nx * D8, hipMemcpyDeviceToDevice, s));
      HIP_CHECK(hipMemcpyAsync(w.U.p, w.es_Ut.p, nu * D8, hipMemcpyDeviceToDevice, s));
      J = Jt; bval = bt; t = tt; Fcur = Ft;
      if (verbose) printf("pmpc_hip: smoothed cone objective: outer %d (rho %.1e) Newton %d  alpha %.4f  step %.3e  F %.12e  decrease %.3e  t %.9e  rows inside %d\n", outer + 1, rho,
                          newton, al, stepmax * al, Fcur, dF, t, (int)std::count_if(mu.begin(), mu.end(), [&](double v) { return v > 1e-6 && v < cap - 1e-6; }));
      if (al < 0.2) n_damped++;
      if (al == 1.0 && stepmax <= step_tol) { inner_ok = true; break; }
      // Full steps in the quadratic regime: after a step s the iterate is off by ~ quad_c s^2.  Once that is a tenth of the tolerance the
      // step that would only confirm it (1e-9 .. 1e-14 in the traces of config D: a third of all Newton steps, two factor sweeps each) is
      // not taken.  quad_c: measured on this solve's own consecutive full steps (the largest ratio seen; 1e3 until there is one).
      if (quad_on && al == 1.0) {
        if (s_prev > 0.0 && s_prev < 1e-2 && stepmax < s_prev) quad_c = std::max(quad_c == 1e3 ? 1.0 : quad_c, std::min(1e6, stepmax / (s_prev * s_prev)));
        if (stepmax <= 1e-3 && 10.0 * quad_c * stepmax * stepmax <= step_tol) { inner_ok = true; break; }
        s_prev = stepmax;
      } else {
        s_prev = -1.0;
      }
      // a hinge too narrow for this start shows at once: step after step cut to a few per cent (every Newton step crosses kinks).  Four of
      // those in a row end the attempt — the remaining twenty would be spent the same way (measured at config D: 50 such steps before the
      // width was right) — and the hinge widens from where the iterate is now
      // (a step cut by the BOXES — the first feasible trial accepted, or nearly — is the barrier doing its work from a start near a
      //  bound, and recovers; only cuts the decrease test demanded beyond feasibility count)
      n_cut = (al < 0.1 && al <= 0.25 * al_feas) ? n_cut + 1 : 0;
      if (n_cut >= 4 && rho < rho_max) break;
    }
    if (!inner_ok) {  // the inner problem was not solved: the multipliers stay, the hinge widens (a smoother inner problem from the same point)
      if (verbose) printf("pmpc_hip: smoothed cone objective: outer %d: inner iteration limit at rho %.1e, widening\n", outer + 1, rho);
      if (rho >= rho_max) break;
      rho_floor = std::max(rho_floor, 3.0 * rho);
      rho = std::min(rho_max, 10.0 * rho);
      continue;
    }
    // multiplier update of the proximal method
    t = solve_t(J);
    double dl = 0.0;
    lam_before = lam;
    for (int i = 0; i < M; i++) {
      const double m_old = cap * sigm(lam[i]), m_new = mult(i, J, t);
      dl = std::max(dl, std::fabs(m_new - m_old));
      lam[i] = std::min(700.0, std::max(-700.0, lam[i] + (J[i] - t) / rho));
    }
    // The multiplier iteration is a fixed-point map that contracts LINEARLY at a fixed hinge width (rate ~ rho / (rho + curvature)): once
    // the width sits on its floor and two successive updates of the rows INSIDE the hinge point the same way with a steady ratio r, the
    // remaining geometric series is summed at once (Aitken):  l += r / (1 - r) * delta.  (Measured at config D: 50 updates at rate 0.74.)
    // Rows at either end of [0, cap] drift to +-infinity at constant speed in the logits anyway and are left alone.
    {
      double num = 0.0, den = 0.0;
      int nin = 0;
      for (int i = 0; i < M; i++) {
        const double m_new = cap * sigm(lam[i]);
        const bool inside = m_new > 1e-4 * cap && m_new < (1.0 - 1e-4) * cap;
        const double d = inside ? lam[i] - lam_before[i] : 0.0;
        if (inside && have_prev_delta) { num += d * dlam_prev[i]; den += dlam_prev[i] * dlam_prev[i]; nin++; }
        dlam_cur[i] = d;
      }
      const double r = den > 0.0 ? num / den : 0.0;
      const bool same_map = rho == rho_at_prev_delta;
      if (have_prev_delta && same_map && nin > 0 && r > 0.3 && r < 0.985 && std::fabs(r - r_prev) <= 0.1 * r && since_extrap >= 2) {
        const double gain = std::min(r / (1.0 - r), 50.0);
        for (int i = 0; i < M; i++)
          if (dlam_cur[i] != 0.0) lam[i] = std::min(700.0, std::max(-700.0, lam[i] + gain * dlam_cur[i]));
        if (verbose) printf("pmpc_hip: smoothed cone objective: outer %d: multiplier updates contract at %.3f: extrapolated (x %.1f, %d rows inside)\n", outer + 1, r, gain, nin);
        have_prev_delta = false;
        since_extrap = 0;
        r_prev = -1.0;
      } else {
        r_prev = (have_prev_delta && same_map) ? r : -1.0;
        dlam_prev.swap(dlam_cur);
        have_prev_delta = true;
        rho_at_prev_delta = rho;
        since_extrap++;
      }
    }
    if (verbose) printf("pmpc_hip: smoothed cone objective: outer %d done (%d Newton steps so far), multiplier change %.3e%s\n", outer + 1, newton, dl, inner_ok ? "" : " (inner limit)");
    (void)last_stage;
    // converged: the multipliers stand still AND they are the multipliers of these costs — complementarity of every epigraph row at the
    // threshold t (a multiplier saturated at the wrong end of [0, cap] also "stands still": its logit moves, its value does not; found
    // by tools/fuzz/fuzz_sequence.py with multipliers remembered from another problem of the same shape)
    double comp = 0.0;
    for (int i = 0; i < M; i++) {
      const double m_i = cap * sigm(lam[i]), v = J[i] - t;
      comp = std::max(comp, v > 0.0 ? v * (cap - m_i) : -v * m_i);
    }
    const bool kkt_ok = comp <= 1e-8 * cap * std::max(1.0, std::fabs(t));
    if (inner_ok && step_tol <= 1e-9 * 1.0000001 && dl <= 1e-7 * cap && kkt_ok) converged = true;
    dl_prev = dl;
    // (a sharper hinge makes the multiplier updates contract faster — wanted while they are far off or contracting slowly; never below a
    //  width whose inner problem this solve has already failed to solve)
    // (the floor is what a failed inner solve from a FAR start taught; next to the fixed point a sharper hinge is solvable again — Newton
    //  starts inside its basin — so the floor decays while the inner solves take full steps)
    if (inner_ok && n_damped == 0) rho_floor *= 0.5;
    if (inner_ok && n_damped == 0 && (dl > 1e-2 * cap || dl > 0.3 * dl_last)) rho = std::max(std::max(rho_min, rho_floor), rho / 3.0);
    else if (n_damped >= 3) rho = std::min(rho_max, 3.0 * rho);
    dl_last = dl;
    Fcur = Fval(J, solve_t(J), bval);
  }
  inf.ipm_iters = newton;
  inf.outer_solves = newton;
  inf.mu = mu_b;
  if (!converged) {
    if (verbose) printf("pmpc_hip: smoothed cone objective: not converged\n");
    return finish(1);
  }
  w.es_U.ensure(nu * D8);
  HIP_CHECK(hipMemcpyAsync(w.es_U.p, w.U.p, nu * D8, hipMemcpyDeviceToDevice, s));
  w.es_key = skey;
  c->cone_lam = lam;
  c->cone_lam_key = -(skey + 7);
  c->cone_rho = rho;
  return finish(0);
}

// Cone objective with hard boxes when NO inequality of a particle's own is active at the answer (always so with every control shared,
// Nc = N — the reference's default consensus horizon — and no state boxes; often so with loose boxes): each particle's cost, minimised over
// its own free controls, is then an exact quadratic of the shared controls,
//     V_i(u_c) = V_i(base) + g_i'(u_c - base) + 1/2 (u_c - base)' H_i (u_c - base),
// with (H_i, g_i) from ONE unweighted factor sweep (+ condensing) at any base point.  The reference's epigraph program (main.jl:204-239)
// is a problem in the Nc u + 1 unknowns (u_c, t) with M quadratic rows and the box on u_c: solved on the host by a primal active-set
// loop over that box around the epigraph solver of epigraph_host.hip — any number of costs on the threshold, one pass, no iteration on
// rankings (which knows two-way ties only and, for k < M, met four-way ones in tools/fuzz/fuzz_cone.py) and no dependence on how the
// sub-problem solve ended (the epigraph path below needs the state of the active-set rounds: an equality-only optimum has none).
// The particles' own boxes (free controls, states) are CHECKED at the answer: if one is violated the assumption was wrong, nothing
// is returned and the caller goes on with the general paths.  Returns -1 where it does not apply, else the status.
static int lcone_free_particles_body(pmpc_ctx *c, const pmpc_problem *p, pmpc_info *info, int verbose) {
  Workspace &w = c->ws;
  hipStream_t s = c->stream;
  const int x = (int)p->xdim, u = (int)p->udim, N = (int)p->N, M = (int)p->M;
  const int Nc = p->Nc < 0 ? N : (int)std::min<long long>(p->Nc, (long long)N), nc = Nc * u;
  const bool has_ub = p->flags & PMPC_HAS_UBOUNDS, has_xb = p->flags & PMPC_HAS_XBOUNDS;
  if (nc < 1 || c->multi() || c->world != 1 || p->weights || M < 2 || (double)M * nc * nc > 2e7 ||
      (p->flags & (PMPC_HAS_SLEW | PMPC_HAS_SLEW0 | PMPC_FORCE_GENERIC | PMPC_F32_MATRICES)) || p->cone_count > 0 || p->soc_W)
    return -1;
  const size_t nx = (size_t)M * N * x, nu = (size_t)M * N * u, D8 = sizeof(double);
  LQArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.u = u; a.N = N; a.M = M; a.Nc = Nc; a.w = 0; a.n = x;
  a.reg_x = p->reg_x; a.reg_u = p->reg_u;
  a.f = p->f; a.fx = p->fx; a.fu = p->fu; a.Q = p->Q; a.R = p->R;
  a.X_prev = p->X_prev; a.U_prev = p->U_prev; a.X_ref = p->X_ref; a.U_ref = p->U_ref;
  a.owner = 1; a.any_slew = 0; a.sym_cost = (p->flags & PMPC_SYMMETRIC_COST) ? 1 : 0;
  if (!lq_fast_supported(a)) return -1;
  const double eps = 1e-3, cap = 1.0 + eps;
  const double kk = (p->cone_k > 0 && p->cone_k < (long long)M) ? (double)p->cone_k : (double)M, K = (1.0 - eps) * kk;
  w.X.ensure(nx * D8); w.U.ensure(nu * D8); w.dX.ensure(nx * D8); w.dU.ensure(nu * D8);
  w.xm.ensure(nx * D8); w.xd.ensure(nx * D8); w.um.ensure(nu * D8); w.ud.ensure(nu * D8);
  w.es_Xt.ensure(nx * D8); w.es_Ut.ensure(nu * D8);
  w.K.ensure((size_t)M * N * 64 * D8); w.Hinv.ensure(nu * u * D8); w.kff.ensure(nu * D8);
  w.gc_part.ensure((size_t)M * nc * D8); w.Hc_part.ensure((size_t)M * nc * nc * D8);
  w.scratch.ensure((size_t)M * 3 * x * nc * D8); w.Jc.ensure((size_t)M * D8); w.duc.ensure((size_t)nc * D8); w.fail.ensure(sizeof(int));
  w.part_max.ensure(2 * PMPC_RED_BLOCKS * D8);
  if (w.es_zero.ensure((size_t)std::max(64, nc) * D8)) HIP_CHECK(hipMemsetAsync(w.es_zero.p, 0, w.es_zero.bytes, s));
  if (w.zeros.bytes == 0) {
    w.zeros.ensure(64 * D8);
    HIP_CHECK(hipMemsetAsync(w.zeros.p, 0, 64 * D8, s));
  }
  if (w.zslew.bytes < (size_t)M * D8 || w.zum1.bytes < (size_t)M * u * D8) {
    w.zslew.ensure((size_t)M * D8); w.zslew0.ensure((size_t)M * D8); w.zum1.ensure((size_t)M * u * D8);
    HIP_CHECK(hipMemsetAsync(w.zslew.p, 0, (size_t)M * D8, s));
    HIP_CHECK(hipMemsetAsync(w.zslew0.p, 0, (size_t)M * D8, s));
    HIP_CHECK(hipMemsetAsync(w.zum1.p, 0, (size_t)M * u * D8, s));
  }
  a.slew = w.zslew.d(); a.slew0 = w.zslew0.d(); a.um1 = w.zum1.d(); a.zeros = w.zeros.d();
  a.K = w.K.d(); a.Hinv = w.Hinv.d(); a.kff = w.kff.d(); a.gc_part = w.gc_part.d(); a.Hc_part = w.Hc_part.d(); a.scratch = w.scratch.d();
  a.duc = w.es_zero.d(); a.dX = w.dX.d(); a.dU = w.dU.d(); a.fail = (int *)w.fail.p; a.X = w.X.d(); a.U = w.U.d();
  a.xm = w.xm.d(); a.xd = w.xd.d(); a.um = w.um.d(); a.ud = w.ud.d();
  w.as_key = -1; w.warm_key = -1; w.es_key = -1;  // (the workspace's factor records and warm-start memories are overwritten)
  pmpc_info inf;
  memset(&inf, 0, sizeof(inf));
  inf.fast_path = 1;
  auto finish = [&](int status) {
    inf.status = status;
    if (status != 0) fill_nan_outputs(c, p);
    if (info) *info = inf;
    return status;
  };
  // base point: the previous controls, the shared ones = particle 0's inside their box (the joint problem takes particle 0's bounds on
  // a shared control, lqp_utils.jl:329-330); states by rollout
  std::vector<double> ub(nc), ub0(nc), lo(nc, -1e300), hi(nc, 1e300);
  HIP_CHECK(hipMemcpyAsync(ub.data(), p->U_prev, (size_t)nc * D8, hipMemcpyDeviceToHost, s));
  if (has_ub) {
    HIP_CHECK(hipMemcpyAsync(lo.data(), p->lu, (size_t)nc * D8, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipMemcpyAsync(hi.data(), p->uu, (size_t)nc * D8, hipMemcpyDeviceToHost, s));
  }
  HIP_CHECK(hipStreamSynchronize(s));
  for (int r = 0; r < nc; r++) {
    if (!(lo[r] == lo[r])) lo[r] = -1e300;  // (NaN = no bound)
    if (!(hi[r] == hi[r])) hi[r] = 1e300;
    if (!(lo[r] <= hi[r])) return -1;  // (an empty box: the general path reports it as the reference does)
    if (!(ub[r] == ub[r])) return -1;
    ub[r] = std::min(std::max(ub[r], lo[r]), hi[r]);
  }
  ub0 = ub;
  HIP_CHECK(hipMemcpyAsync(w.U.p, p->U_prev, nu * D8, hipMemcpyDeviceToDevice, s));
  HIP_CHECK(hipMemcpyAsync(w.U.p, ub.data(), (size_t)nc * D8, hipMemcpyHostToDevice, s));
  launch_share_cons(w.U.d(), M, N, u, Nc, s);
  launch_rollout_fast(a, w.U.d(), w.X.d(), s);
  HIP_CHECK(hipMemsetAsync(w.fail.p, 0, sizeof(int), s));
  launch_grad_prep(a, s);
  launch_bwd_fast(a, true, s);
  if (Nc > 1) launch_cond_fast(a, s);
  launch_fwd_fast(a, s);  // (shared step zero: every particle's own optimal response to the base shared controls, in dX / dU)
  launch_step_to(w.X.d(), w.dX.d(), 1.0, w.es_Xt.d(), (long long)nx, s);
  launch_step_to(w.U.d(), w.dU.d(), 1.0, w.es_Ut.d(), (long long)nu, s);
  launch_particle_cost(a, w.es_Xt.d(), w.es_Ut.d(), w.Jc.d(), s);  // V_i(base)
  std::vector<double> J(M), Hh((size_t)M * nc * nc), gh((size_t)M * nc);
  int failflag = 0;
  HIP_CHECK(hipMemcpyAsync(J.data(), w.Jc.p, (size_t)M * D8, hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipMemcpyAsync(Hh.data(), w.Hc_part.p, Hh.size() * D8, hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipMemcpyAsync(gh.data(), w.gc_part.p, gh.size() * D8, hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipMemcpyAsync(&failflag, w.fail.p, sizeof(int), hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  inf.structured_solves = 1;
  if (failflag) return -1;
  for (int i = 0; i < M; i++) {  // (the off-diagonal blocks live in the upper triangle)
    double *Hi = &Hh[(size_t)i * nc * nc];
    for (int r = 0; r < nc; r++)
      for (int q_ = r + 1; q_ < nc; q_++) Hi[q_ + (size_t)nc * r] = Hi[r + (size_t)nc * q_];
  }
  // primal active-set loop over the box of the shared controls around the epigraph solve
  std::vector<unsigned char> held(nc, 0);
  for (int r = 0; r < nc; r++) held[r] = (ub[r] <= lo[r] || ub[r] >= hi[r]) ? 1 : 0;
  std::vector<double> lam(M, K / (double)M), delta(nc, 0.0), Hd(nc);
  double t = 0.0, gmax = 0.0;
  for (double v : gh) gmax = std::max(gmax, std::fabs(v));
  const double tolg = 1e-9 * std::max(1.0, gmax);
  bool done = false;
  int it = 0;
  for (; it < 20 * nc + 50 && !done; it++) {
    const int est = pmpc_epigraph_solve_host(M, nc, J.data(), Hh.data(), gh.data(), held.data(), K, cap, lam.data(), delta.data(), &t, verbose > 1);
    if (est != 0 && verbose) printf("pmpc_hip: cone objective, free particles: the host solve stopped short of its tolerance\n");
    double al = 1.0;
    int blocking = -1;
    for (int r = 0; r < nc; r++) {
      if (held[r]) { delta[r] = 0.0; continue; }
      const double un = ub[r] + delta[r];
      if (un > hi[r] && delta[r] > 0.0) { const double v = (hi[r] - ub[r]) / delta[r]; if (v < al) { al = v; blocking = r; } }
      if (un < lo[r] && delta[r] < 0.0) { const double v = (lo[r] - ub[r]) / delta[r]; if (v < al) { al = v; blocking = r; } }
    }
    if (blocking >= 0) {  // move the base to the first bound met on the way and hold it there
      al = std::max(0.0, al);
      for (int i = 0; i < M; i++) {
        const double *Hi = &Hh[(size_t)i * nc * nc];
        double *gi = &gh[(size_t)i * nc];
        double gd = 0.0, dHd = 0.0;
        for (int r = 0; r < nc; r++) {
          double acc = 0.0;
          for (int q_ = 0; q_ < nc; q_++) acc += Hi[r + (size_t)nc * q_] * delta[q_];
          Hd[r] = acc;
          gd += gi[r] * delta[r];
          dHd += delta[r] * acc;
        }
        J[i] += al * gd + 0.5 * al * al * dHd;
        for (int r = 0; r < nc; r++) gi[r] += al * Hd[r];
      }
      for (int r = 0; r < nc; r++) ub[r] += al * delta[r];
      ub[blocking] = delta[blocking] > 0.0 ? hi[blocking] : lo[blocking];
      held[blocking] = 1;
      if (verbose) printf("pmpc_hip: cone objective, free particles: shared control %d meets its bound (step fraction %.3e)\n", blocking, al);
      continue;
    }
    // the full step stays inside the box: multipliers of the held bounds = gradient of sum lam_i V_i at base + delta
    int worst = -1;
    double wv = tolg;
    for (int r = 0; r < nc; r++) {
      if (!held[r]) continue;
      double gr = 0.0;
      for (int i = 0; i < M; i++) {
        const double *Hi = &Hh[(size_t)i * nc * nc];
        double acc = gh[(size_t)i * nc + r];
        for (int q_ = 0; q_ < nc; q_++) acc += Hi[r + (size_t)nc * q_] * delta[q_];
        gr += lam[i] * acc;
      }
      const bool at_lo = ub[r] <= lo[r], at_hi = ub[r] >= hi[r];
      const double viol = (at_lo && at_hi) ? 0.0 : (at_lo ? -gr : (at_hi ? gr : std::fabs(gr)));  // (lo == hi: the control stays where it is)
      if (viol > wv) { wv = viol; worst = r; }
    }
    if (worst >= 0) {
      held[worst] = 0;
      if (verbose) printf("pmpc_hip: cone objective, free particles: shared control %d leaves its bound (multiplier %.3e of the wrong sign)\n", worst, wv);
      continue;
    }
    if (est != 0) {
      // never an unverified iterate: the host solve stopped short of ITS tolerance — with many costs on the threshold its
      // multiplier updates crawl while the point is long exact — so the KKT conditions of the epigraph rows are checked on what it returned:
      // sum lam = K; lam_i = cap above the threshold cost, 0 below, anything on it; stationarity of sum lam_i (g_i + H_i delta) on the free
      // shared controls; t at the breakpoint the multipliers name.
      double slam = 0.0, jscale = 1.0;
      std::vector<double> Jn(M), gs(nc, 0.0);
      for (int i = 0; i < M; i++) {
        const double *Hi = &Hh[(size_t)i * nc * nc], *gi = &gh[(size_t)i * nc];
        double lin = 0.0, quad = 0.0;
        for (int r = 0; r < nc; r++) {
          double acc = 0.0;
          for (int q_ = 0; q_ < nc; q_++) acc += Hi[r + (size_t)nc * q_] * delta[q_];
          lin += gi[r] * delta[r];
          quad += delta[r] * acc;
          gs[r] += lam[i] * (gi[r] + acc);
        }
        Jn[i] = J[i] + lin + 0.5 * quad;
        jscale = std::max(jscale, std::fabs(Jn[i]));
        slam += lam[i];
      }
      const double jt = 1e-9 * jscale, lt = 1e-9 * cap;
      bool ok = std::fabs(slam - K) <= 1e-9 * std::max(1.0, K);
      for (int i = 0; i < M && ok; i++) {
        if (lam[i] < -lt || lam[i] > cap + lt) ok = false;
        else if (Jn[i] > t + jt && lam[i] < cap - lt) ok = false;   // above the threshold: full multiplier
        else if (Jn[i] < t - jt && lam[i] > lt) ok = false;         // below: none
      }
      for (int r = 0; r < nc && ok; r++)
        if (!held[r] && std::fabs(gs[r]) > tolg * std::max(1.0, K)) ok = false;
      if (!ok) {
        if (verbose) printf("pmpc_hip: cone objective, free particles: the last host solve did not converge and its point fails the KKT check: no answer from this path\n");
        return -1;
      }
    }
    done = true;
  }
  if (!done) {
    if (verbose) printf("pmpc_hip: cone objective, free particles: the box active set of the shared controls did not settle\n");
    return -1;
  }
  // total shared step from the DEVICE's base point; the forward sweep adds every particle's own response to it
  std::vector<double> dtot(nc);
  for (int r = 0; r < nc; r++) {
    ub[r] = std::min(std::max(ub[r] + delta[r], lo[r]), hi[r]);
    dtot[r] = ub[r] - ub0[r];
  }
  HIP_CHECK(hipMemcpyAsync(w.duc.p, dtot.data(), (size_t)nc * D8, hipMemcpyHostToDevice, s));
  LQArgs a3 = a;
  a3.duc = w.duc.d();
  launch_fwd_fast(a3, s);
  launch_step_to(w.X.d(), w.dX.d(), 1.0, w.es_Xt.d(), (long long)nx, s);
  launch_step_to(w.U.d(), w.dU.d(), 1.0, w.es_Ut.d(), (long long)nu, s);
  HIP_CHECK(hipMemcpyAsync(w.es_Ut.p, ub.data(), (size_t)nc * D8, hipMemcpyHostToDevice, s));  // (a held shared control sits exactly on its bound)
  launch_share_cons(w.es_Ut.d(), M, N, u, Nc, s);
  // the assumption: no box of a particle's own is violated at this point
  const int B = PMPC_RED_BLOCKS;
  std::vector<double> pm(2 * B, 0.0);
  HIP_CHECK(hipMemsetAsync(w.part_max.p, 0, 2 * B * D8, s));
  Slab sl;
  memset(&sl, 0, sizeof(sl));
  if (has_xb) { sl.count = (long long)nx; sl.lo = p->lx; sl.hi = p->ux; sl.z = w.es_Xt.d(); launch_violation(sl, w.part_max.d(), s); }
  if (has_ub) { sl.count = (long long)nu; sl.lo = p->lu; sl.hi = p->uu; sl.z = w.es_Ut.d(); launch_violation(sl, w.part_max.d() + B, s); }
  HIP_CHECK(hipMemcpyAsync(pm.data(), w.part_max.p, 2 * B * D8, hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  double viol = 0.0;
  for (double v : pm) viol = std::max(viol, v == v ? v : 1e300);
  if (verbose) printf("pmpc_hip: cone objective, free particles: threshold cost %.9e after %d host solves; largest violation of a particle's own box %.3e\n", t, it, viol);
  if (viol > 1e-9) return -1;
  HIP_CHECK(hipMemcpyAsync(p->X_out, w.es_Xt.p, nx * D8, hipMemcpyDeviceToDevice, s));
  HIP_CHECK(hipMemcpyAsync(p->U_out, w.es_Ut.p, nu * D8, hipMemcpyDeviceToDevice, s));
  HIP_CHECK(hipStreamSynchronize(s));
  c->cone_lam_key = -1;
  inf.outer_solves = it;
  return finish(0);
}

int lcone_body(pmpc_ctx *c, const pmpc_problem *p0, double smooth_alpha, pmpc_info *info, int verbose) {
  HIP_CHECK(hipSetDevice(c->device));
  Workspace &w = c->ws;
  hipStream_t s = c->stream;
  // (an fp32-storage problem runs the cone objective on widened copies: its weighted QPs and the particle costs read doubles)
  pmpc_problem pwide;
  const pmpc_problem *p = p0;
  if (p0->flags & PMPC_F32_MATRICES) {
    pwide = widened_f32_problem(c, p0);
    p = &pwide;
  }
  // particles are sharded in equal contiguous blocks (bench.py's layout): the ranking of the particle costs is global, so
  // every rank gathers all costs (all-reduce(sum) of a zero-padded vector) and takes the same decisions
  const size_t Ml = p->M, M = Ml * (size_t)c->world, off = (size_t)c->rank * Ml, D8 = sizeof(double);
  const double eps = 1e-3;  // COST_ANCHOR_EPS, main.jl:223
  w.pw.ensure(Ml * D8); w.Jc.ensure(Ml * D8); w.Jg.ensure(std::max<size_t>(M, 2) * D8);
  std::vector<double> user(M, 1.0), pw(Ml), J(M), loc(Ml);
  auto gather = [&](const double *local_dev, std::vector<double> &global) {  // global[rank*Ml + i] = local[i] on every rank
    if (!c->multi()) {
      HIP_CHECK(hipMemcpyAsync(global.data(), local_dev, Ml * D8, hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipStreamSynchronize(s));
      return;
    }
    HIP_CHECK(hipMemsetAsync(w.Jg.p, 0, M * D8, s));
    HIP_CHECK(hipMemcpyAsync(w.Jg.d() + off, local_dev, Ml * D8, hipMemcpyDeviceToDevice, s));
    allreduce(c, w.Jg.p, M, ncclFloat64, ncclSum);
    HIP_CHECK(hipMemcpyAsync(global.data(), w.Jg.p, M * D8, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
  };
  if (c->multi()) {  // equal shards are assumed by the offsets above
    double cnt[2] = {(double)Ml, -(double)Ml};
    HIP_CHECK(hipMemcpyAsync(w.Jg.p, cnt, 2 * D8, hipMemcpyHostToDevice, s));
    allreduce(c, w.Jg.p, 2, ncclFloat64, ncclMax);
    HIP_CHECK(hipMemcpyAsync(cnt, w.Jg.p, 2 * D8, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    if (cnt[0] != -cnt[1]) {
      fprintf(stderr, "pmpc_hip: pmpc_lcone_solve_device needs the same number of particles on every rank\n");
      fill_nan_outputs(c, p);
      if (info) {
        memset(info, 0, sizeof(*info));
        info->status = 2;
      }
      return 2;
    }
  }
  if (p->weights) gather(p->weights, user);
  pmpc_problem q = *p;
  q.weights = w.pw.d();
  // smooth_cstr = "logbarrier" (main.jl:246-262): -1/alpha sum log(alpha slack) replaces the hard boxes
  q.barrier_mu = (smooth_alpha == smooth_alpha && smooth_alpha > 0.0) ? 1.0 / smooth_alpha : 0.0;
  if (q.barrier_mu > 0.0 && !(p->flags & (PMPC_HAS_XBOUNDS | PMPC_HAS_UBOUNDS))) q.barrier_mu = 0.0;  // (no boxes: nothing to smooth, main.jl:248)
  if (p->smooth_cstr == 1 && q.barrier_mu > 0.0) {
    // smooth_cstr = "squareplus" (main.jl:265-279): soft boxes, tau(v) = beta/2 (v + sqrt(v^2 + 1/alpha^2)) per side; only the
    // full-space Newton path has it (mu carries 1/alpha)
    const double sbeta = p->smooth_beta > 0.0 ? p->smooth_beta : 1.0;
    int st_s = -1;
    if (M == 1 && !c->multi() && (p->weights || (p->flags & (PMPC_HAS_SLEW | PMPC_HAS_SLEW0)))) {
      // ONE particle (the shape of nearly every reference example): the epigraph row is degenerate, the problem is
      //   min (1 - eps) w J(z) + sum of hinges   <=>   min (1 - eps) J(z) + sum of hinges of slope beta / w      (scale_probs_cost!, main.jl:96-112)
      // and slew penalties go through the increment form of the QP path (state [x; u], control increments: the control boxes — and with
      // them their hinges — become state boxes; one particle, so no shared control is counted twice).
      double wgt = 1.0;
      if (p->weights) {
        HIP_CHECK(hipMemcpyAsync(&wgt, p->weights, sizeof(double), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
      }
      if (wgt > 0.0) {
        pmpc_problem p1 = *p;
        p1.weights = nullptr;
        // (the cone program takes each particle's cost from qp_repr_Pq, cone_utils.jl:64-95, which keeps the first-step slew term
        //  slew_reg0 |u_0 - u_{-1}|^2 whatever Nc is — the joint QP assembly loses it at Nc = 0, lqp_utils.jl:165; with ONE particle a
        //  consensus horizon of one stage is the same problem and carries the term)
        if (p1.Nc == 0) p1.Nc = 1;
        if (p->flags & (PMPC_HAS_SLEW | PMPC_HAS_SLEW0)) {
          LQArgs t;
          memset(&t, 0, sizeof(t));
          t.x = (int)(p->xdim + p->udim); t.u = (int)p->udim; t.N = (int)p->N; t.M = 1; t.sym_cost = 1;
          if (p->N >= 2 && (p->flags & PMPC_SYMMETRIC_COST) && !(p->flags & PMPC_FORCE_GENERIC) && lq_fast_supported(t)) {
            pmpc_problem qa;
            SlewAug g;
            build_slew_increment_problem(c, &p1, qa, g);
            qa.weights = nullptr;
            st_s = lcone_smooth_body(c, &qa, q.barrier_mu, info, verbose, 1, sbeta / wgt);
            const int Nc1 = p->Nc < 0 ? (int)p->N : (int)p->Nc;
            if (st_s == 0) launch_slew_split(w.sa_Xo.d(), w.sa_Uo.d(), p->X_out, p->U_out, (long long)p->N, (int)p->xdim, (int)p->udim, (int)p->N, 0, g.cons_lo, g.cons_hi, s);
            else if (st_s > 0) fill_nan_outputs(c, p);
            (void)Nc1;
          }
        } else {
          st_s = lcone_smooth_body(c, &p1, q.barrier_mu, info, verbose, 1, sbeta / wgt);
        }
      }
    } else {
      st_s = lcone_smooth_body(c, p, q.barrier_mu, info, verbose, 1, sbeta);
    }
    if (st_s >= 0) return st_s;
    fprintf(stderr, "pmpc_hip: smooth_cstr = \"squareplus\" needs boxes to smooth, M (Nc u)^2 <= 2e7, no fp32 storage, a compiled (xdim, udim) pair and — with slew penalties or particle weights — M = 1 (slew: symmetric costs, N >= 2, a compiled (xdim + udim, udim) pair)\n");
    fill_nan_outputs(c, p);
    if (info) { memset(info, 0, sizeof(*info)); info->status = 2; }
    return 2;
  }
  if (q.barrier_mu > 0.0 && c->opt[OPT_CONE_EPIGRAPH] != 0.0 && M > 1) {
    const int st_s = lcone_smooth_body(c, p, q.barrier_mu, info, verbose);
    if (st_s >= 0) return st_s;
  }
  // The free-particles path (lcone_free_particles_body) is exact for any tie pattern but host-heavy (it gathers every particle's
  // condensed Hessian): it goes FIRST only on a shape where it was needed and worked before; otherwise it is the last resort behind the
  // ranking iteration (measured, config B with Nc = N: 4200 it/s through the ranking, 280 through this path).
  const long long fkey = (((((((long long)M * 1000003 + (long long)p->N) * 131 + (long long)p->xdim) * 131 + (long long)p->udim) * 131 + p->Nc + 2) * 1000003 + (long long)p->cone_k + 1) * 4 +
                          ((p->flags & PMPC_HAS_XBOUNDS) ? 2 : 0) + ((p->flags & PMPC_HAS_UBOUNDS) ? 1 : 0));
  // (option cone_path: which body answers does not have to depend on what the context saw before — 1: the free-particles body first,
  //  2: never, 3: the rank-based iteration alone; tests/test_cone_ties_gpu.py runs the same problems through each)
  const int forced = (int)c->opt[OPT_CONE_PATH];
  const bool fp_applies = !(q.barrier_mu > 0.0) && c->opt[OPT_CONE_EPIGRAPH] != 0.0 && M > 1 && forced != 2 && forced != 3;
  if ((p->flags & PMPC_COLD_START) || c->fp_key != fkey) { c->fp_key = fkey; c->fp_ok = -1; }
  if (forced == 1) c->fp_ok = 1;
  if (fp_applies && c->fp_ok == 1) {
    const int st_f = lcone_free_particles_body(c, p, info, verbose);
    if (st_f >= 0) return st_f;
    c->fp_ok = 0;
  }
  pmpc_info inf, last;
  memset(&last, 0, sizeof(last));
  int outer = 0, solves_total = 0, ipm_total = 0;
  // threshold rank of the piecewise-linear epigraph cost in t: the m*-th cheapest particle, m* = ceil(2 eps M / (1+eps))
  // (general k, main.jl:204-227: multipliers lambda_i in [0, 1+eps] of the cone rows sum to (1-eps) k, so the
  // n_hi = floor((1-eps) k / (1+eps)) costliest particles carry 1+eps, the next one the remainder, the rest nothing)
  const double kk = (p->cone_k > 0 && p->cone_k < (long long)M) ? (double)p->cone_k : (double)M;
  const long long n_hi = (long long)std::floor((1.0 - eps) * kk / (1.0 + eps) + 1e-12);
  const long long mstar = std::max<long long>(1, (long long)M - n_hi);
  // floor weight of the weightless particles: they shift the shared controls by O(w_floor (m* - 1) / n_hi) — 1e-4 is harmless for
  // the one-in-500 of k = M, the worst-k objective (half the particles weightless at k = M / 2) needs it smaller
  const double w_hi = 1.0 + eps, w_thr = (1.0 - eps) * kk - (1.0 + eps) * (double)n_hi, w_floor = kk < (double)M ? 1e-9 : 1e-4;

  auto solve_with = [&](const std::vector<double> &rankw) -> int {
    for (size_t i = 0; i < Ml; i++) pw[i] = user[off + i] * rankw[off + i];
    HIP_CHECK(hipMemcpyAsync(w.pw.p, pw.data(), Ml * D8, hipMemcpyHostToDevice, s));
    const int st = pmpc_lqp_solve_device(c, &q, &inf, verbose > 1);
    // the later weighted QPs of this call start from THIS solve's set and solution (kept in the workspace), not from the
    // caller's X_prev / U_prev: the caller's promise covers the first one only
    q.flags &= ~(unsigned)PMPC_PREV_IS_LAST_SOLUTION;
    outer++;
    solves_total += inf.structured_solves;
    ipm_total += inf.ipm_iters;
    last = inf;
    if (st != 0) return st;
    pmpc_particle_costs_device(c, p, p->X_out, p->U_out, w.Jc.d());
    gather(w.Jc.d(), J);
    for (size_t i = 0; i < M; i++) J[i] *= user[i];  // scale_probs_cost! (main.jl:96-112) acts on the costs themselves
    return 0;
  };
  auto finish = [&](int status) {
    last.status = status;
    last.outer_solves = outer;
    last.structured_solves = solves_total;
    last.ipm_iters = ipm_total;
    if (status != 0) fill_nan_outputs(c, p);
    if (info) *info = last;
    return status;
  };
  // the (m* - 1) cheapest particles carry no weight in the reference's objective (their trajectories are then not
  // unique); they keep w_floor here so that every particle's sub-problem stays strictly convex
  auto rank_weights = [&](const std::vector<size_t> &low, std::vector<double> &rw) {
    std::fill(rw.begin(), rw.end(), w_hi);
    for (size_t k = 0; k + 1 < low.size(); k++) rw[low[k]] = w_floor;
    rw[low.back()] = std::max(w_thr, w_floor);
  };
  auto cheapest = [&](std::vector<size_t> &low) {  // indices of the m* cheapest particles, ascending cost
    std::vector<size_t> idx(M);
    for (size_t i = 0; i < M; i++) idx[i] = i;
    std::partial_sort(idx.begin(), idx.begin() + mstar, idx.end(), [&](size_t a_, size_t b_) { return J[a_] < J[b_] || (J[a_] == J[b_] && a_ < b_); });
    low.assign(idx.begin(), idx.begin() + mstar);
  };

  std::vector<double> rw(M, w_hi);
  if (M == 1) {  // one particle: weight 1 - eps, same minimiser as the QP (k = 1)
    rw[0] = 1.0 - eps;
    return finish(solve_with(rw));
  }
  // ---- hard boxes: the epigraph problem in the space of the shared controls (epigraph_host.hip) ------------------------------------
  // Scaling a particle's whole cost changes neither its gains nor its active set nor its optimum GIVEN the shared controls, so the
  // sweeps run unweighted (a particle of weight zero takes the minimum-cost completion: the limit of the floor weight of the
  // weighted-QP iteration below, without the floor) and the multipliers lam_i of the M epigraph rows enter only where the particles
  // meet: the consensus system sum_i lam_i (H_i, g_i) (LQArgs::cons_w).  Each sub-problem solve leaves the particles' condensed
  // quadratics behind; when its costs J_i and the multipliers violate the KKT conditions of the epigraph problem (lam = 1 + eps above
  // the threshold cost, 0 below, anything on it), the host solves that problem on those quadratics — ties among any number of
  // particles are ordinary degenerate rows there — and the next solve applies the result: one more round if no box changes status.
  const int Ncc = p->Nc < 0 ? (int)p->N : (int)std::min<long long>(p->Nc, (long long)p->N), ncv = Ncc * (int)p->udim;
  // (sharded: every rank holds the multipliers of ALL particles, checks the gathered costs on the host and solves the same epigraph problem
  //  on the all-gathered quadratics: identical decisions everywhere, as for the rank-based iteration)
  const bool epi_multi = c->multi();
  const bool epi_on = c->opt[OPT_CONE_EPIGRAPH] != 0.0 && forced != 3 && !(q.barrier_mu > 0.0) && (double)M * ncv * ncv <= 2e7 && !(p->flags & PMPC_FORCE_GENERIC);
  if (epi_on) {
    pmpc_problem qq = *p;
    qq.weights = nullptr;
    qq.barrier_mu = 0.0;
    const double Ksum = (1.0 - eps) * kk, cap = 1.0 + eps;
    std::vector<double> lam(M, Ksum / (double)M), cw(Ml), lam_loc(Ml);
    // KKT check of the epigraph rows on the device (k_epi_check): costs of the accepted point, threshold cost, violation — 32 bytes back.
    // Inside pmpc_scp_loop_device both kernels go in BEHIND the first batch of rounds, ahead of the speculative follow-up work (residual,
    // next linearisation), so the answer is there when the host has seen the rounds end.
    if (w.epi_lam.ensure(Ml * D8)) c->epi_lam_host.clear();  // (fresh allocations hold nothing of what the host mirrors remember)
    if (w.cons_w.ensure(Ml * D8)) c->cons_w_host.clear();
    w.epi_out.ensure(4 * D8);
    double chk[4] = {0.0, 0.0, 0.0, 0.0};
    auto enqueue_check = [&]() {
      pmpc_particle_costs_device(c, p, p->X_out, p->U_out, w.Jc.d());
      if (!epi_multi) launch_epi_check(w.epi_lam.d(), w.Jc.d(), p->weights, (int)Ml, cap, w.epi_out.d(), s, c->mirror_dev->epi, &c->mirror_dev->epi_seq, ++c->epi_seq);
    };
    int n_solves = 0, hook_solve = -1;
    bool costs_on_host = false;
    auto solve_cons = [&](bool weighted) -> int {
      bool need = p->weights != nullptr || weighted;
      if (need) {
        for (size_t i = 0; i < Ml; i++) cw[i] = (weighted ? lam[off + i] : 1.0) * user[off + i];
        if (c->cons_w_host != cw) {  // (unchanged since the last upload — the steady state of an SCP loop: nothing to send)
          HIP_CHECK(hipMemcpyAsync(w.cons_w.p, cw.data(), Ml * D8, hipMemcpyHostToDevice, s));
          c->cons_w_host = cw;
        }
      }
      for (size_t i = 0; i < Ml; i++) lam_loc[i] = lam[off + i];
      if (!epi_multi && c->epi_lam_host != lam_loc) {
        HIP_CHECK(hipMemcpyAsync(w.epi_lam.p, lam_loc.data(), Ml * D8, hipMemcpyHostToDevice, s));
        c->epi_lam_host = lam_loc;
      }
      c->cons_w_active = (need && ncv > 0) ? w.cons_w.d() : nullptr;
      bool fired_here = false;
      std::function<void()> orig;
      if (c->post_batch) {
        orig.swap(c->post_batch);
        c->post_batch = [&]() {
          fired_here = true;
          enqueue_check();
          orig();
        };
      }
      int st_;
      try {
        st_ = pmpc_lqp_solve_device(c, &qq, &inf, verbose > 1);
      } catch (...) {
        c->cons_w_active = nullptr;
        c->post_batch = nullptr;
        throw;
      }
      c->cons_w_active = nullptr;
      if (c->post_batch) {  // not fired (the solve did not go through a first batch of rounds): the caller's hook stays for a later solve
        c->post_batch = nullptr;
        c->post_batch.swap(orig);
      }
      if (fired_here) hook_solve = n_solves;
      n_solves++;
      qq.flags &= ~(unsigned)PMPC_PREV_IS_LAST_SOLUTION;  // (later solves of this call start from the workspace's set and solution)
      outer++;
      solves_total += inf.structured_solves;
      ipm_total += inf.ipm_iters;
      last = inf;
      if (st_ != 0) return st_;
      if (!(fired_here && c->spec_ok)) enqueue_check();  // (what the hook computed saw unfinished outputs, or there was no hook)
      if (epi_multi) {
        // sharded: all costs to every rank (one all-reduce of a zero-padded vector), the same check on the host everywhere
        gather(w.Jc.d(), J);
        for (size_t i = 0; i < M; i++) J[i] *= user[i];
        costs_on_host = true;
        double tsum = 0.0, jmin_full = 1e300, jmax_zero = -1e300, viol = 0.0;
        size_t nfr = 0;
        for (size_t i = 0; i < M; i++) {
          if (lam[i] > 1e-12 && lam[i] < cap - 1e-12) { tsum += J[i]; nfr++; }
          else if (lam[i] >= cap - 1e-12) jmin_full = std::min(jmin_full, J[i]);
          else jmax_zero = std::max(jmax_zero, J[i]);
        }
        const double th = nfr ? tsum / (double)nfr : ((jmin_full < 1e300 && jmax_zero > -1e300) ? 0.5 * (jmin_full + jmax_zero) : (jmin_full < 1e300 ? jmin_full : jmax_zero));
        for (size_t i = 0; i < M; i++) {
          if (!(J[i] == J[i])) viol = 1e300;
          else if (lam[i] > 1e-12 && lam[i] < cap - 1e-12) viol = std::max(viol, std::fabs(J[i] - th));
          else if (lam[i] >= cap - 1e-12) viol = std::max(viol, th - J[i]);
          else viol = std::max(viol, J[i] - th);
        }
        chk[0] = viol; chk[1] = th; chk[2] = (double)nfr;
        return 0;
      }
      costs_on_host = false;
      // (polled from the host-coherent mirror: the stream — which may hold the next linearisation behind the check — is not drained)
      wait_published(c, &c->mirror->epi_seq, c->epi_seq);
      memcpy(chk, (const void *)c->mirror->epi, 4 * D8);
      return 0;
    };
    auto fetch_costs = [&]() {
      if (costs_on_host) return;
      gather(w.Jc.d(), J);
      for (size_t i = 0; i < M; i++) J[i] *= user[i];
    };
    const long long lkey = (((((((long long)M * 1000003 + (long long)p->N) * 131 + (long long)p->xdim) * 131 + (long long)p->udim) * 131 + p->Nc + 2) * 1000003 + (long long)kk)) * 64 + c->world;
    int st_ = 0;
    if (ncv == 0) return finish(solve_cons(false));  // no shared controls: every particle minimises its own cost, whatever its multiplier
    const bool remembered = c->opt[OPT_CONE_RANK_MEMORY] != 0.0 && !(p->flags & PMPC_COLD_START) && c->cone_lam_key == lkey && c->cone_lam.size() == M;
    if (remembered) lam = c->cone_lam;
    c->cone_lam_key = -1;
    st_ = solve_cons(remembered);
    if (st_ != 0) return finish(st_);
    std::vector<double> Hh, gh, dl(ncv), delta(ncv);
    std::vector<int> act0(ncv);
    std::vector<unsigned char> held(ncv);
    bool settled = false;
    for (int it = 0; it < 30; it++) {
      // KKT of the epigraph rows at (lam, J), from the device: {violation, threshold cost, rows on the threshold}
      const double viol = chk[0], tthr = chk[1];
      if (verbose) printf("pmpc_hip: cone epigraph outer %d: threshold cost %.9e, %d rows on it, KKT violation %.3e\n", it, tthr, (int)chk[2], viol);
      if (viol <= 1e-9 * std::max(1.0, std::fabs(tthr))) { settled = true; break; }
      fetch_costs();
      // the particles' quadratics around the accepted point: left by the LAST round of the active-set rounds (gradient at that round's
      // base point, the consensus step it applied).  A solve that ended elsewhere (equality-only optimum, interior-point iteration) is
      // repeated warm: one round that changes nothing.
      for (int rep = 0; rep < 2 && !(last.fast_path && last.ipm_iters == 0 && last.active_set_rounds >= 1); rep++) {
        st_ = solve_cons(true);
        if (st_ != 0) return finish(st_);
      }
      if (!(last.fast_path && last.ipm_iters == 0 && last.active_set_rounds >= 1)) {
        if (verbose) printf("pmpc_hip: cone epigraph: the sub-problem does not end in the active-set rounds; weighted-QP iteration instead\n");
        break;
      }
      Hh.resize(M * (size_t)ncv * ncv); gh.resize(M * (size_t)ncv);
      if (!epi_multi) {
        HIP_CHECK(hipMemcpyAsync(Hh.data(), w.Hc_part.p, Hh.size() * D8, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipMemcpyAsync(gh.data(), w.gc_part.p, gh.size() * D8, hipMemcpyDeviceToHost, s));
      } else {  // all-gather through one all-reduce of a zero-padded buffer [H of every particle | g of every particle]
        const size_t nH = (size_t)ncv * ncv, tot = M * (nH + ncv);
        w.epi_gath.ensure(tot * D8);
        HIP_CHECK(hipMemsetAsync(w.epi_gath.p, 0, tot * D8, s));
        HIP_CHECK(hipMemcpyAsync(w.epi_gath.d() + off * nH, w.Hc_part.p, Ml * nH * D8, hipMemcpyDeviceToDevice, s));
        HIP_CHECK(hipMemcpyAsync(w.epi_gath.d() + M * nH + off * ncv, w.gc_part.p, Ml * (size_t)ncv * D8, hipMemcpyDeviceToDevice, s));
        allreduce(c, w.epi_gath.p, tot, ncclFloat64, ncclSum);
        HIP_CHECK(hipMemcpyAsync(Hh.data(), w.epi_gath.p, M * nH * D8, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipMemcpyAsync(gh.data(), w.epi_gath.d() + M * nH, M * (size_t)ncv * D8, hipMemcpyDeviceToHost, s));
      }
      HIP_CHECK(hipMemcpyAsync(dl.data(), w.as_delta.p, ncv * D8, hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipMemcpyAsync(act0.data(), w.as_act.p, ncv * sizeof(int), hipMemcpyDeviceToHost, s));  // (particle 0's stages < Nc come first)
      HIP_CHECK(hipStreamSynchronize(s));
      for (int r = 0; r < ncv; r++) held[r] = act0[r] != 0 ? 1 : 0;
      for (size_t i = 0; i < M; i++) {
        double *Hi = &Hh[i * (size_t)ncv * ncv], *gi = &gh[i * (size_t)ncv];
        for (int r = 0; r < ncv; r++)
          for (int cc = r + 1; cc < ncv; cc++) Hi[cc + (size_t)ncv * r] = Hi[r + (size_t)ncv * cc];  // (off-diagonal blocks live in the upper triangle)
        if (i == 0)
          for (int r = 0; r < ncv; r++)
            if (held[r]) Hi[r + (size_t)ncv * r] = 1.0;  // (the 1e30 penalty of a held shared control is not part of the cost; the step there is zero)
        for (int r = 0; r < ncv; r++) {  // gradient at the accepted point = gradient at the last round's base + H_i (applied step)
          double acc = 0.0;
          for (int cc = 0; cc < ncv; cc++) acc += (held[cc] ? 0.0 : Hi[r + (size_t)ncv * cc] * dl[cc]);
          gi[r] += acc;
        }
        if (user[i] != 1.0) {
          for (size_t e_ = 0; e_ < (size_t)ncv * ncv; e_++) Hi[e_] *= user[i];
          for (int r = 0; r < ncv; r++) gi[r] *= user[i];
        }
      }
      double tpred = 0.0;
      const int est = pmpc_epigraph_solve_host((int)M, ncv, J.data(), Hh.data(), gh.data(), held.data(), Ksum, cap, lam.data(), delta.data(), &tpred, verbose);
      if (est != 0 && verbose) printf("pmpc_hip: cone epigraph: the host solve stopped short of its tolerance (the next check decides)\n");
      st_ = solve_cons(true);
      if (st_ != 0) return finish(st_);
    }
    // work the caller enqueued behind the first batch of rounds (pmpc_scp_loop_device) saw the final outputs only if that solve was the last
    if (c->spec_fired && hook_solve != n_solves - 1) c->spec_ok = false;
    if (settled) {
      c->cone_lam = lam;
      c->cone_lam_key = lkey;
      return finish(0);
    }
    if (verbose) printf("pmpc_hip: cone epigraph: not settled; weighted-QP iteration\n");
    q.flags &= ~(unsigned)PMPC_PREV_IS_LAST_SOLUTION;
  }
  // Inside an SCP loop the ranking of the particle costs rarely changes between iterations: the assignment the previous solve of this
  // shape settled on is tried FIRST — if the ranking at its optimum reproduces it, that is the fixed point (the same consistency test
  // as below: weights = multipliers of the epigraph rows, KKT of the reference's problem) after ONE weighted QP instead of two.
  const long long rwkey = ((((((long long)M * 1000003 + (long long)p->N) * 131 + (long long)p->xdim) * 131 + (long long)p->udim) * 131 + p->Nc + 2) * 1000003 + (long long)kk) * 64 + c->world;
  int st = 0;
  std::vector<size_t> low;
  std::vector<double> rw1(M), rw2(M), rw_prev;
  if (c->opt[OPT_CONE_RANK_MEMORY] != 0.0 && !(p->flags & PMPC_COLD_START) && c->cone_rw_key == rwkey && c->cone_rw.size() == M) {
    rw1 = c->cone_rw;
    low.assign(1, 0);  // (only its last entry is read before the first ranking, for the verbose line)
  } else {
    st = solve_with(rw);  // uniform weights: the QP optimum ranks the particles
    if (st != 0) return finish(st);
    cheapest(low);
    rank_weights(low, rw1);
  }
  c->cone_rw_key = -1;
  // fixed-point iteration on the WEIGHT assignment the ranking implies (the order inside the floor-weight group is irrelevant)
  bool settled = false;
  for (int it = 0; it < 12 && !settled; it++) {
    rw = rw1;
    st = solve_with(rw);
    if (st != 0) return finish(st);
    const size_t thr_old = low.back();
    cheapest(low);
    rank_weights(low, rw2);
    if (verbose) printf("pmpc_hip: cone outer %d  threshold particle %zu -> %zu  J_thr %.9e\n", it + 1, thr_old, low.back(), J[low.back()]);
    if (rw2 == rw1) {
      settled = true;
      c->cone_rw = rw1;  // (a kink's interpolated weights are not remembered: they are no assignment by rank)
      c->cone_rw_key = rwkey;
      break;
    }
    if (!rw_prev.empty() && rw2 == rw_prev) {
      // 2-cycle between two rankings: the optimum sits on a kink J_a = J_b between the particles a, b whose weights differ
      // most between the two assignments; on the segment rw(theta) = theta rw1 + (1 - theta) rw2 the gap J_a - J_b is
      // monotone in theta (a loses weight as theta grows, so its cost rises relative to b's): bisection.  mstar = 1: a and
      // b share the deficit 2 eps M; mstar > 1: they swap the threshold / floor weights.
      size_t a_ = 0, b_ = 0;
      double da = 0.0, db = 0.0;
      for (size_t i = 0; i < M; i++) {
        const double dlt = rw1[i] - rw2[i];
        if (dlt < da) { da = dlt; a_ = i; }  // lighter under assignment 1
        if (dlt > db) { db = dlt; b_ = i; }  // lighter under assignment 2
      }
      if (!(da < 0.0 && db > 0.0)) break;  // (cannot happen: the assignments differ)
      double lo = 0.0, hi = 1.0;  // theta = 1: assignment 1 (a light, then J_a > J_b), theta = 0: assignment 2
      for (int bis = 0; bis < 60; bis++) {
        const double th = 0.5 * (lo + hi);
        for (size_t i = 0; i < M; i++) rw[i] = th * rw1[i] + (1.0 - th) * rw2[i];
        st = solve_with(rw);
        if (st != 0) return finish(st);
        const double gap = J[a_] - J[b_];
        if (verbose) printf("pmpc_hip: cone kink bisection %2d  theta %.12f  J_a - J_b %+.3e\n", bis, th, gap);
        if (std::fabs(gap) <= 1e-11 * std::max(1.0, std::fabs(J[a_]))) break;
        if (gap > 0.0) hi = th; else lo = th;
      }
      // accept if the weights are consistent with the ranking at the kink: every particle lighter than the threshold pair is
      // cheaper than it, every full-weight particle costlier (KKT of the epigraph problem, multipliers lambda_i = w_i)
      const double jk = 0.5 * (J[a_] + J[b_]), tolj = 1e-9 * std::max(1.0, std::fabs(jk));
      settled = std::fabs(J[a_] - J[b_]) <= 1e-8 * std::max(1.0, std::fabs(jk));
      for (size_t i = 0; i < M && (settled || verbose); i++) {
        if (i == a_ || i == b_) continue;
        const bool dips = rw[i] >= w_hi && J[i] < jk - tolj;  // a full-weight particle dips below the threshold cost
        const bool rises = rw[i] < w_hi && J[i] > jk + tolj;  // a down-weighted particle rises above it
        // a multiplier strictly between 0 (the floor) and 1 + eps belongs to a row ON the threshold: a third particle that carries the
        // threshold remainder while its cost sits below the kink's is no KKT point (found by tools/fuzz/fuzz_cone.py with k < M: the
        // kink of the two COSTLIEST particles was accepted with the remainder weight on the third)
        const bool off = rw[i] > 2.0 * w_floor && rw[i] < w_hi * (1.0 - 1e-12) && std::fabs(J[i] - jk) > tolj;
        if (dips || rises || off) {
          settled = false;
          if (verbose) printf("pmpc_hip: cone kink: particle %zu (weight %.3e) is on the wrong side of the threshold cost by %.3e (J_thr %.9e)\n", i, rw[i], J[i] - jk, jk);
        }
      }
      break;
    }
    rw_prev = rw1;
    rw1 = rw2;
  }
  if (!settled) {
    // no consistent threshold set within the outer iteration limit (more than two costs on the threshold, typically): the exact path for
    // particles without an active inequality of their own, if that is what they are; else a failed solve (NaN outputs), never an
    // unverified iterate
    if (verbose) printf("pmpc_hip: cone objective: the threshold set did not settle\n");
    if (fp_applies && c->fp_ok != 0) {
      const int st_f = lcone_free_particles_body(c, p, info, verbose);
      c->fp_ok = st_f == 0 ? 1 : 0;
      if (st_f >= 0) return st_f;
    }
    return finish(1);
  }
  return finish(0);
}

}  // extern "C"
