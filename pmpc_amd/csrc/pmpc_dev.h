// pmpc_dev.h — shared declarations of the device solver (gfx950 only).
//
// Data layout in HBM (everything fp64, the C-ABI layout of include/pmpc_abi.h):
//   vectors  v[(i*N + j)*d + r]                  particle i, stage j, component r
//   matrices m[((i*N + j)*cols + t)*rows + r]    column-major (rows x cols) block per (i, j)
// so one particle's horizon is one contiguous slab and a GPU shard is a contiguous slice.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

// A failed HIP / RCCL call on a solve path must not take the host process down: the drop-in ABI's failure convention is
// NaN outputs (PMPC.jl/src/osqp_solver.jl:65-71 -> (None, None, None) in pmpc/scp_mpc.py:391-394).  HIP_CHECK reports and
// throws; every extern "C" entry point catches, fills its outputs with NaN where it still can and returns status 2.
struct PmpcHipError {
  int code;
  const char *what, *file;
  int line;
};
#define HIP_CHECK(expr)                                                                       \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      fprintf(stderr, "pmpc_hip: HIP error %s at %s:%d (%s)\n", hipGetErrorString(_e), __FILE__, \
              __LINE__, #expr);                                                               \
      throw PmpcHipError{(int)_e, #expr, __FILE__, __LINE__};                                 \
    }                                                                                         \
  } while (0)
// destructors and clean-up paths: report, never throw
#define HIP_WARN(expr)                                                                        \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess)                                                                     \
      fprintf(stderr, "pmpc_hip: HIP error %s at %s:%d (%s)\n", hipGetErrorString(_e), __FILE__, __LINE__, #expr); \
  } while (0)

// Device-resident control block of the active-set rounds (kernels_as.hip, k_as_ctl): the decisions the host used to take
// after reading the change counters back are taken on the device, so that several rounds can be enqueued ahead.
struct AsCtl {
  int done;          // no further round has work: every kernel of a later round returns at once
  int status;        // 0 accepted (set unchanged: the optimum), 1 not settled (stalled / round limit), 2 numerical failure
  int round;         // rounds completed
  int max_rounds, last_changes, stalls;
  int stall_limit;   // rounds whose changes may fail to halve before the attempt is given up (2; stage cones: 8 — their case changes
                     // thin out slowly while the boundary cones' Newton iteration converges)
  int open;          // stage cones whose Newton iteration has not converged yet (no case change, but |ds|, |dz| or |Phi| above
                     // tolerance), all ranks: the set counts as settled only when this is zero as well
  int cnt[4];        // {released, activated, bad (NaN / empty box / broken promise), failure flag} of the last round (all ranks)
  int hist[16][2];   // per round: released, activated
  double tol_l;      // sign tolerance of the multipliers for the NEXT round
  double dual_scale;
  double worst[16];  // per round: largest KKT violation among the changed controls — how far outside its box a control that got
                     // clamped stood (relative to max(1, |bound|)), how negative a released multiplier was (relative to dual_scale)
};

// the factor sweeps of the active-set rounds checkpoint their cost-to-go at the top of the stages FIRST << k, k = 0, 1, .. (LQArgs::as_ck):
// a geometric ladder — status changes sit near the start of the horizon (config D: none above stage 7, 85 % at stages <= 4), each
// checkpoint costs the sweep five stores, and a restart anywhere on the ladder redoes at most twice the stages an ideal one would.
// (first rung at 4 against 8, same box: restarted sweeps 0.036 -> 0.033 ms per step at 512 particles, 0.257 -> 0.221 at config E)
#ifndef PMPC_AS_CK_LOG
#define PMPC_AS_CK_LOG 2
#endif
#define PMPC_AS_CK_FIRST (1 << PMPC_AS_CK_LOG)

// Arguments of the structured LQ kernels (Riccati factor / vector sweeps / forward sweep).
struct LQArgs {
  int x, u, N, M, Nc;
  int w;  // u if slew penalties are active (stage state augmented with the previous control) else 0
  int n;  // x + w
  double reg_x, reg_u;
  const double *pw;  // per-particle cost weights (null = 1): J = sum_i pw_i J_i (cone path, `weights` setting)
  const double *cons_w;  // per-particle CONSENSUS weights (null = 1): sum_i cons_w_i (H_i, g_i) at the shared controls, sweeps unweighted (cone objective)
  // ABI inputs
  const double *f, *fx, *fu, *Q, *R, *X_prev, *U_prev, *X_ref, *U_ref;
  const double *slew, *slew0, *um1;  // per particle, never null (zeros when absent)
  // current iterate (dynamics-consistent)
  const double *X, *U;
  // IPM terms: extra Hessian diagonals / gradient shifts (null when absent)
  const double *Dx, *Du, *wx, *wu;
  int du_full;  // Du holds full (u x u) column-major blocks per (particle, stage) instead of diagonals (stage-cone extension)
  // fast path only: gradient pre-pass outputs (launch_grad_prep), same shapes as X / U
  //   xm = pw (X - X_ref), xd = pw reg_x (X - X_prev) + wx, um = pw (U - U_ref), ud = pw reg_u (U - U_prev) + wu (free stages;
  //   consensus stages: wu only on the owner's particle 0)
  double *xm, *xd, *um, *ud;
  const double *zeros;  // >= 64 readable zero doubles (lanes without an entry load from here, stride 0)
  // factor storage
  double *K;     // [M][N][u*n]  col-major u x n
  double *Hinv;  // [M][N][u*u]
  double *kff;   // [M][N][u]
  // consensus condensing
  double *gc_part;  // [M][nc]
  double *Hc_part;  // [M][nc*nc]
  int cost_lds;     // k_particle_cost: doubles of dynamic LDS the launch was given (0: none)
  double *Hc_grp;   // [groups][nc*nc] or null: the condensing kernel sums the particles of a workgroup before storing (k_cond_fast_grouped)
  double *scratch;  // [M][3*n*nc]
  const double *duc;  // [nc] consensus step
  // outputs of the forward sweep
  double *dX, *dU;
  // active-set mode of the fast forward sweep (null = off): per control status (0 free / 1 at lower / 2 at upper bound), its
  // box, and per-particle counters {released, activated, NaN seen}; the base controls are `U`
  int *as_act;
  const double *as_lo, *as_hi;
  int *as_cnt;
  double *as_viol;  // per particle: largest KKT violation among its changed controls (see AsCtl::worst); null = not recorded
  // first round of a warm start inside an SCP loop: the base point (X_prev, snapped U_prev) is NOT rolled out — its dynamics
  // defect r_j = f_j - X_prev_j (elementwise, since base == linearisation point) rides through the sweeps instead:
  // backward s_j += S_j r_j, forward dx_j += r_j.  Non-null selects the DEFECT kernel variants.
  const double *defect;
  // settled particles (no status change in the previous round, so their factors, condensed Hessian and feed-forward are
  // still valid): the factor sweep / condensing skip them (as_settled_in), the forward sweep marks them (as_settled_out)
  const int *as_settled_in;
  int *as_settled_out;
  // checkpointed restart of the later rounds' factor sweeps (kernels_as.hip): every factor sweep leaves the cost-to-go (S, s) and
  // the base state it was expanded around at the top of the stages j = PMPC_AS_CK_FIRST << k, k = 0 .. ck_slots - 1; the forward sweep records
  // the highest stage whose status changed (as_jhi, -1: none; the cone / state-row passes raise it to the highest stage whose
  // Newton terms changed); an unsettled particle's next factor sweep starts at the lowest checkpoint at or above that stage
  // — the recursion above it is unchanged, s follows the base point exactly: s + S (x_base_now - x_base_then).  Null = off.
  const int *as_perm;  // later rounds: particle of workgroup b (unsettled ones first, k_as_perm); null = identity
  double *as_ck;
  int *as_jhi;
  int ck_slots;
  unsigned long long *ck_stat;  // {restarted sweeps, stages they ran, sweeps of unsettled particles from the terminal cost, stages they ran} (pmpc_restart_stats)
  double as_big, as_tol_p, as_tol_l;
  // stage-cone rounds with one consensus stage: once the step of every FREE shared control is below as_freeze_tol (relative), it is taken
  // as zero by every particle — a convergence tolerance on the shared controls — and the settled particles leave the forward sweep at
  // once (measured at config E: the last 2-3 rounds of 7 serve 4-25 particles while moving the shared control by 1e-9 .. 1e-14)
  double as_freeze_tol;
  // active-set sweeps of kernels_as.hip: base point in (Xb, Ub) — ignored in the first round of a no-rollout warm start, whose
  // base is (X_prev, U_prev) —, new base point out (Xo, Uo: may alias Xb, Ub), the consensus step as applied (as_delta, nc
  // doubles, written by particle 0), the control block (tolerance of the round; null = as_tol_l) and the early-exit flag
  const double *Xb, *Ub;
  double *Xo, *Uo;
  double *as_delta;
  const AsCtl *as_ctl;
  const int *done;
  // sharded active-set rounds with a consensus horizon: the previous round's change counters ride behind [Hc | gc] in this
  // round's consensus all-reduce (0 off, 1 first round of an attempt: nothing to carry yet, 2 later rounds)
  int as_merge;
  // single rank, Nc = 1, active-set rounds: the consensus system is summed over cons_G block partials and solved by every
  // wave of the forward sweep itself (kernels_as.hip) — one launch less per round.  cons_G = 0: a.duc holds the step.
  const double *cons_tH, *cons_tg;
  int cons_G;
  // stage cones on the active-set sweeps (kernels_cone.hip prepares them per round): Newton terms of every (particle, stage) — a full
  // (u x u) block added to H_uu and a vector added to the control gradient (consensus stages: once, by the owner's particle 0) — and
  // the forward sweep's record of each stage's own Newton step u_b + du BEFORE clamping (the cone multiplier updates are valid for
  // that step only).  Null = no cones (the CONE instantiations are not launched).
  const double *cone_H, *cone_g;
  double *as_uraw;
  int *as_open;    // per particle: open stage cones (zeroed by the forward sweep, counted by the cone pass)
  double *as_T;    // forward sweep, one consensus stage: sensitivity records [M][N][64] (null: off) — see k_fwd_as<.., SENS>
  // state boxes on the active-set sweeps (kernels_xbox.hip prepares them per round): a penalty on the diagonal of the stage's state
  // cost and a gradient term, per state entry (M,N,x).  Null = none (the XBOX instantiations are not launched).
  const double *xb_D, *xb_g;
  int mat32;       // fx, fu, Q, R (and the factor record a.K of the active-set sweeps) are FLOAT arrays (fp32-storage mode; kernels_as.hip only)
  int owner;       // this rank holds global particle 0 (whose bounds the consensus controls use)
  int any_slew;    // slew_reg or slew_reg0 present
  int sym_cost;    // caller guarantees Q_j = Q_j', R_j = R_j' (else OSQP's triu semantics need the generic path)
  int *fail;
};

// One "slab" of box-constrained variables for the elementwise IPM kernels.
struct Slab {
  long long count;  // M*N*d
  int d, N, Nc;
  int is_u;   // consensus duplicates exist for stages j < Nc
  int owner;
  const double *lo, *hi;
  double *z, *dz;
  const double *dz2;  // corrector difference step (null during the predictor): the step is dz + dz2
  double *tl, *tu, *ll, *lu, *cl, *cu, *D, *w;
};

// A slab plus what the fused per-iteration pass needs: bounded or not, and the gradient pre-pass outputs
struct SlabEx {
  Slab s;
  int bounded;
  const double *ref, *prev;  // X_ref / X_prev (resp. U_ref / U_prev)
  double *gm, *gd;           // fast-path gradient arrays (null on the generic path)
  double reg;
  const double *pw;          // per-particle cost weights (null = 1)
  long long per;             // entries per particle (N * d)
};

// Device-resident IPM scalars.
struct IpmScal {
  double comp_sum;   // sum w * (t_l l_l + t_u l_u)           (all-reduced: sum)
  double cnt;        // number of finite bounds (weighted)    (all-reduced: sum)
  double muaff_sum;  // S1 = sum w (t dl + l dt)                 (all-reduced: sum)
  double pad0;       // S2 = sum w dt dl                         (all-reduced: sum)
  double res_max;    // max |slack residual|                  (all-reduced: max)
  double viol_max;   // max bound violation of the unconstrained optimum (all-reduced: max)
  unsigned long long amin_bits;  // min step ratio as u64 bits (all-reduced: min)
  unsigned long long pad1;
  double mu, sigma, sigmu, alpha_aff, alpha, nu;
  int iter, status;
  // log-barrier smoothing (cone path, smooth_alpha): stop AT centrality mu_target instead of driving mu to 0
  double mu_target;  // 0 = hard constraints
  double dev_max;    // max_k |t_k l_k - mu_target| of the current iterate (all-reduced: max)
  double *part_dev;  // [2 * PMPC_RED_BLOCKS] block partials of dev_max
};

#define PMPC_RED_BLOCKS 1024

// ---- kernels_generic.hip ------------------------------------------------------------------------
size_t lq_generic_lds_bytes(const LQArgs &a);
void launch_rollout(const LQArgs &a, const double *U, double *X, hipStream_t s);
void launch_bwd_generic(const LQArgs &a, bool factor, hipStream_t s);
void launch_fwd_generic(const LQArgs &a, hipStream_t s);
void launch_reduce_particles_hg(const double *Hc_part, const double *gc_part, double *tmp, double *Hg, int M, int nc, hipStream_t s, int MH = -1);  // MH: slabs of H (default M)
void launch_reduce_particles(const double *src, double *tmp, double *dst, int M, int E, hipStream_t s);
// a round-control call (k_as_ctl's arguments) that rides in the next consensus-partials launch instead of a launch of its own
struct AsCtlCall {
  AsCtl *ctl;  // null: nothing pending
  const int *cnt_part;
  int M;
  const int *fail;
  AsCtl *mirror;
  unsigned long long *mirror_seq;
  unsigned long long seq;
  const double *viol;
  const int *open_part;  // per-particle open stage cones (null: none)
  const int *settled;    // with perm: the launch order of the round this call rides in (as_perm_block; null: none)
  int *perm;
};
int launch_cons_partials(const double *Hc_part, const double *gc_part, int M, int nc, double *tmp, const AsCtlCall &pend,
                         hipStream_t s);  // -> number of partials
void launch_cons_small(const double *Hc_part, const double *gc_part, int M, int nc, bool with_H, double *Hg, double *tmp, bool solve_now,
                       double *Lc, double *duc, int *fail, hipStream_t s);
void launch_cons_solve(double *Hc, double *Lc, const double *gc, double *duc, int nc, bool factor, int *fail,
                       hipStream_t s);
// KKT check of the cone objective's epigraph rows on the device: out = {violation, threshold cost, rows on the threshold, 0}
void launch_epi_check(const double *lam, const double *J, const double *user, int M, double cap, double *out, hipStream_t s, double *mirror = nullptr,
                      unsigned long long *mirror_seq = nullptr, unsigned long long seq = 0);
// lambda_i (H_i, g_i) for the reductions (cone objective: consensus weights, see kernels_generic.hip)
void launch_cons_scale(const double *Hc_part, const double *gc_part, const double *w, int M, int nc, bool with_H, double *outH, double *outg,
                       const int *as_act, double as_big, const double *Du, const double *wu, int u, int owner, hipStream_t s);

// ---- kernels_fast.hip ---------------------------------------------------------------------------
bool lq_fast_supported(const LQArgs &a);
void launch_bwd_fast(const LQArgs &a, bool factor, hipStream_t s);
void launch_fwd_fast(const LQArgs &a, hipStream_t s);
int cond_fast_groups(int M);  // slabs of LQArgs::Hc_grp
void launch_cond_fast(const LQArgs &a, hipStream_t s);  // off-diagonal blocks of the condensed consensus Hessian (Nc > 1)
void launch_rollout_fast(const LQArgs &a, const double *U, double *X, hipStream_t s);
void launch_grad_prep(const LQArgs &a, hipStream_t s);
// generic-path active-set rounds: counters[3] <- *fail (fail != null) and / or publication of counters[0..3] (mirror_cnt != null)
void launch_as_publish(int *counters, const int *fail, int *mirror_cnt, unsigned long long *mirror_seq, unsigned long long seq, hipStream_t s);
// J[i] = 1/2 z_i' P_i z_i + q_i' z_i + r_i of PMPC.jl/src/qp_utils.jl:60-162 at (X, U) (unweighted), any dims / slew
void launch_particle_cost(const LQArgs &a, const double *X, const double *U, double *J, hipStream_t s);

// ---- kernels_as.hip -----------------------------------------------------------------------------
// sweeps of one active-set round on the control boxes (a.as_act etc. set; a.defect != null: first round of a no-rollout warm start;
// a.as_settled_in != null: skip the settled particles, which only refresh g_i += H_i as_delta)
void launch_bwd_as(const LQArgs &a, hipStream_t s);
void launch_fwd_as(const LQArgs &a, hipStream_t s);
// round control: reduce the per-particle counters into ctl->cnt (+ failure flag) and / or decide (done, status, next tolerance);
// publishes ctl to the host-coherent mirror with sequence number `seq` when the rounds are over or the batch ends
void launch_as_perm(const int *settled, int M, int *perm, const int *done, hipStream_t s);  // kernels_as.hip: unsettled particles first
void launch_as_begin(AsCtl *ctl, int *fail, int max_rounds, double dual_scale, hipStream_t s, int stall_limit = 2);  // fresh control block of an attempt, *fail = 0
// `tail`: 5 doubles {released, activated, bad, failure, open cones} (sharded runs)
void launch_as_ctl(AsCtl *ctl, const int *cnt_part, int M, const int *fail, int reduce, int decide, int last_of_batch, AsCtl *mirror,
                   unsigned long long *mirror_seq, unsigned long long seq, hipStream_t s, double *tail = nullptr, const double *viol = nullptr,
                   const int *open_part = nullptr);

// ---- kernels_cone.hip (stage-wise second-order cones inside the active-set rounds) --------------------------------------------
// One elementwise pass per round, thread = (particle, stage): FINISH the round just swept (cone multipliers from each stage's own
// Newton step, new case of the projection's generalised Jacobian, change / open counters) and PREPARE the next one (Newton terms
// cone_H, cone_g from the new base point and multipliers).  Semismooth Newton on  s - Proj_K(s - z) = 0,  s = A u + c:
//   interior (s - z in K): cone off;   polar: s -> 0 (apex) by penalty + multiplier;   else: ONE equality along e- = (1, -wh)/sqrt2
//   (penalty + multiplier) and the finite curvature (1 - theta)/theta on the tangential directions.
struct ConeArgs {
  int M, N, u, q, Nc, owner;
  const double *U, *Uraw;     // new base controls / each stage's raw Newton step (finish); U alone for the first preparation
  const double *A, *c;        // cone data, device: A = [v'; W] ((q+1) x u, row-major), c = (v0, w0) — general form: the rows of
                              // all cones stacked (rows x u), per (particle, stage) if per_stage
  int ncones, qs[4], rows, per_stage;  // general form: cones per stage, their sizes (0 = linear row), total rows; 1 cone, shared data, q >= 1: k_cone_step
  const double *R;            // cost blocks (penalty scale rho = rho_scale * (trace(R)/u + reg_u)); float array if r32
  int r32;
  double reg_u, rho_scale;
  double *z, *rec;            // multipliers (M,N,rows) and the per-cone record of the prepared round (M,N,ncones,PMPC_CONE_REC)
  double *H, *g;              // outputs: Newton terms (M,N,u,u) column-major blocks, (M,N,u)
  int *cnt, *settled, *open;  // per particle: as_cnt (3 ints: case changes are added to [1]), settled flag (cleared), open cones
  int *jhi;                   // per particle or null: raised to the highest stage with a changed / open cone (LQArgs::as_jhi)
  const int *done;
  const AsCtl *ctl;           // round control block (tolerance of the round); null: 1e-11 dual_scale
  int finish;                 // 0: prepare only (first round of an attempt)
  double tol_step, tol_phi, dual_scale;
};
#define PMPC_CONE_REC 12
bool cone_as_supported(int u, int q);
bool cone_as_dims_supported(int x, int u);  // kernels_as.hip: (xdim, udim) pairs with CONE instantiations of the sweeps
void launch_cone_step(const ConeArgs &a, hipStream_t s);
void launch_cone_drop_redundant_lo(double *lo, const double *A, const double *c, int q, long long rows, int u, hipStream_t s);

// ---- kernels_xbox.hip: state boxes inside the active-set rounds -----------------------------------------------------
// One pass per round behind the forward sweep (multipliers, statuses and counters of the round that just ran, then the next round's
// terms xb_D, xb_g).  A state cannot be put ON its bound the way a control can (the dynamics decide it), so a binding state box is
// a row of the semismooth Newton iteration on the natural map  s - max(0, s - z) = 0,  s = x - lo  (or hi - x):  inactive while
// s - z >= 0; else held by penalty + multiplier estimate, rho/2 (s_b + dx)^2 - z dx, z+ = z - rho s_new.
struct XboxArgs {
  int M, N, x;
  const double *X;           // new base states (finish) / first base point
  const double *lo, *hi;     // (M,N,x)
  const double *Q, *pw;      // cost blocks and particle weights (or null): penalty rho_i = rho_scale pw_i (max_j,r |Q_ijrr| + reg_x)
  double reg_x, rho_scale;
  double *qmax;              // per particle: max_j,r |Q_ijrr|, written by the prepare call (finish = 0) and read by the finish calls of the attempt —
                             // the strided pass over the cost diagonals touches every line of Q (236 MB at config D), once per attempt instead of per round
  double *z;                 // multipliers (M,N,x), >= 0, of the side named by st
  int *st;                   // 0 free, 1 lower side held, 2 upper side held
  double *D, *g;             // outputs (M,N,x)
  int *cnt, *settled, *open; // per particle: as_cnt (status changes are added to [1]), settled flag (cleared), rows not yet on their bound
  int *jhi;                  // per particle or null: raised to 1 + the highest stage with a changed / open state row (LQArgs::as_jhi: the row's penalty
                             // sits in the cost-to-go at the top of ITS stage, so the factor sweep has to start above it)
  const int *done;
  const AsCtl *ctl;
  int finish;                // 0: prepare only (first round of an attempt)
  double tol, dual_scale;
  double z_tol;              // a held row stays open while its multiplier still moves by more than this (relative)
  int keep_on_clamp;         // a sweep whose forward pass clamped a control of the particle keeps the old multipliers
  double act_frac;           // a round holds only the violated rows within this fraction of the particle's largest violation (0: all)
  int ctrl_from;             // state entries r >= ctrl_from are controls in disguise (slew increment form): always held when violated
};
bool xbox_as_dims_supported(int x, int u);  // kernels_as.hip: (xdim, udim) pairs with XBOX instantiations of the factor sweep
void launch_xbox_step(const XboxArgs &a, hipStream_t s);
void launch_xbox_from_ipm(const struct Slab &sx, int *st, double *z, hipStream_t s);  // statuses / multipliers from an interior-point iterate

// ---- kernels_ipm.hip ----------------------------------------------------------------------------
void launch_block_transpose(const double *in, double *out, int rows, int cols, long long n, hipStream_t s);
void launch_host_checks(const double *lx, const double *ux, long long nx, const double *lu, const double *uu, long long nu,
                        const double *Q, long long nq, int x, const double *R, long long nr, int u, int *flags, hipStream_t s);
void launch_axpy(double *y, const double *xv, double alpha, long long n, hipStream_t s);
void launch_fill(double *y, double v, long long n, hipStream_t s);
void launch_cons_bounds(double *lo, double *hi, int M, int N, int u, int Nc, hipStream_t s);
void launch_init_base(double *U, const double *U_prev, int M, int N, int u, int Nc, hipStream_t s);
void launch_violation(const Slab &sl, double *part_max, hipStream_t s);
void launch_violation_sum(const Slab &sl, const double *za, const double *zb, double *part_max, hipStream_t s);  // of za + zb
void launch_ipm_clip(const Slab &sl, hipStream_t s);
// primal-dual active-set finish (kernels_ipm.hip): act 0 free / 1 lower / 2 upper; counters = {released, activated, NaN seen}
void launch_as_setup(const Slab &sl, int from_ipm, int keep_base, int *act, double *ztry, double big, hipStream_t s);
void launch_as_check(const Slab &sl, int *act, const double *ztry, double big, double tol_p, double tol_l, int *counters,
                     unsigned long long *worst_bits, hipStream_t s);
void launch_as_accept_all(const Slab &sl, const int *act, const double *ztry, double *Uout, const double *xtry, const double *dx,
                          long long nx, double *Xws, double *Xout, hipStream_t s);  // accepted point -> workspace + caller's outputs
void launch_ipm_init_slack(const Slab &sl, double mu0, hipStream_t s, double thr_frac = 1e-2);
void launch_ipm_prepare(const Slab &sl, int corrector, const IpmScal *sc, double *part_sum, double *part_cnt,
                        double *part_max, hipStream_t s);
void launch_ipm_step(const Slab &sl, int corrector, IpmScal *sc, double *part_s1, double *part_s2, hipStream_t s);
void launch_ipm_advance(const SlabEx &X, const SlabEx &U, int do_update, const IpmScal *sc, double *part_sum, double *part_cnt,
                        double *part_max, hipStream_t s);
// phases: 0 reset | 1 violation | 2 IPM start | 3 predictor | 4 corrector (see kernels_ipm.hip)
void launch_ipm_exchange(int phase, bool pack, bool unpack, IpmScal *sc, const int *fail, double *xch, int rank, int world,
                         const double *part_sum, const double *part_cnt, const double *part_max, int nblocks, hipStream_t s,
                         double mu_target = 0.0, double *part_dev = nullptr,  // these two: phase 0 only
                         IpmScal *mirror = nullptr, unsigned long long *mirror_seq = nullptr, unsigned long long seq = 0);

// ---- kernels_soc.hip (stage-wise second-order cones on the controls, primal-dual path following) ------------------
struct SocArgs {
  int M, N, u, Nc, q, owner;
  const double *U, *dU, *dU2;        // iterate, predictor step, corrector difference step
  int corr;                          // 0 predictor pass, 1 corrector pass
  double *cl, *cu, *cc;              // second-order terms of the box sides (M,N,u) and of the cone (M,N,q+1)
  const double *lo, *hi;             // control boxes (M,N,u) or null
  const double *W, *w0, *v;          // cone || W u + w0 || <= v'u + v0: W (q x u) row-major, w0 (q), v (u); device
  double v0, mu, sigmu;
  double *sl, *su, *sc;              // slacks of the box sides (M,N,u) and of the cone (M,N,q+1): variables of their own
  double *zl, *zu, *zc;              // their duals
  double *dsl, *dsu, *dsc, *dzl, *dzu, *dzc;  // Newton steps
  double *Hadd, *wu;                 // outputs: A'W^-2 A as full (u x u) blocks (-> LQArgs.Du, du_full), gradient shift
  int *fail;
};
// mode 0: cold start (s from u, z = mu s^-1), 2: warm start (s from u, z kept), 1: Newton system blocks from (u, s, z)
int launch_soc_prepare(const SocArgs &a, int mode, double *part_sum, double *part_cnt, hipStream_t s);
int launch_soc_step(const SocArgs &a, unsigned long long *amin_bits, double *part_s1, double *part_s2, hipStream_t s);
void launch_soc_update(const SocArgs &a, double alpha, double *X, const double *dX, const double *dX2, double *U, long long nx,
                       long long nu, long long ncz, hipStream_t s);
void launch_soc_fill_u(double *U, const double *u0, long long tot, int u, hipStream_t s);

// ---- kernels_epi.hip (cone objective with log-barrier smoothing: elementwise passes of the full-space Newton iteration) -------------
void launch_bar_prep(const double *X, const double *U, const double *lx, const double *ux, const double *lu, const double *uu, double *Dx, double *wx,
                     double *Du, double *wu, double mu, long long nx, long long nu, int u, int N, int Nc, int owner, double *part_val, double *part_min,
                     double *out2, hipStream_t s, int mode = 0, double beta = 1.0);  // out2 = {barrier value, smallest slack}; mode 1: squareplus hinge (mu = 1/alpha)
void launch_cost_dots(const LQArgs &a, const double *X, const double *U, const double *dX1, const double *dU1, const double *dX2, const double *dU2,
                      double *out, hipStream_t s, const double *bx = nullptr, const double *bu = nullptr);  // per particle {grad J . d1, grad J . d2, d1' hess J d1}
int launch_epi_newton(const double *Hc_part, const double *gb, const double *ga, const double *dots, const double *sig, int M, int nc, double k_minus_summu,
                      double *coef, double *duc, int *fail, hipStream_t s, int mode = 0, double *xch = nullptr);  // (Nc u + 1) system of the smoothed cone objective's Newton step (0: Nc u > 8)
void launch_axpy_particle(const double *a_, const double *b_, const double *coef, double *y, long long per, long long tot, hipStream_t s);
void launch_step_to(const double *a_, const double *b_, double alpha, double *y, long long tot, hipStream_t s);
void launch_interior(double *U, const double *lo, const double *hi, long long tot, double frac, hipStream_t s);
void launch_share_cons(double *U, int M, int N, int u, int Nc, hipStream_t s);  // consensus stages: particle 0's controls in every particle

// ---- dynamics.hip -------------------------------------------------------------------------------
// slew penalties on the MFMA path: the problem restated in control increments (kernels_slew.hip)
struct SlewAug {
  int x, u, N, M, Nc, has_xb, has_ub, has_um1;
  double dx, du;
  const double *f, *fx, *fu, *Xp, *Up, *Q, *R, *Xr, *Ur, *lx, *ux, *lu, *uu, *cons_lo, *cons_hi, *slew, *slew0, *um1;
  double *af, *afx, *afu, *aXp, *aUp, *aQ, *aR, *aXr, *aUr, *alo, *ahi;
};
void launch_slew_augment(const SlewAug &g, hipStream_t s);
void launch_slew_split(const double *Z, const double *W, double *X, double *U, long long rows, int x, int u, int N, int Nc,
                       const double *cons_lo, const double *cons_hi, hipStream_t s);
// (jac32: fx / fu are FLOAT arrays — the fp32-storage mode)
void launch_linearize(int model, int N, int M, const double *x0, const double *X_prev, const double *U_prev,
                      const double *params, double *f, double *fx, double *fu, hipStream_t s, int jac32 = 0);
void launch_linearize_with_residual(int model, int N, int M, const double *x0, const double *X_prev, const double *U_prev,
                                    const double *params, double *f, double *fx, double *fu, const double *Xr, const double *Xrp,
                                    const double *Ur, const double *Urp, int x, int u, double *res_out, hipStream_t s, int jac32 = 0);
void launch_widen_f32(const float *src, double *dst, long long n, hipStream_t s);  // dst[k] = (double)src[k]
bool f32_as_dims_supported(int x, int u);  // (xdim, udim) pairs with fp32-storage instantiations of the active-set sweeps
void launch_scp_residual(const double *X, const double *Xp, const double *U, const double *Up, long long rows, int x, int u,
                         double *out, hipStream_t s, bool zero_out = true);
