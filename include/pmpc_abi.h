/*
 * pmpc_abi.h — C ABI of libpmpc_hip.so, the MI355X-native replacement for the
 * PMPC.jl back end of StanfordASL/pmpc.
 *
 * Part 1 is the DROP-IN boundary: the two symbols the reference's pybind11 module
 * binds (PMPC.jl/pmpcjl/module.cpp:9-23) and the Julia library exports
 * (PMPC.jl/src/c_interface.jl:77-141 `c_lqp_solve`, :146-214 `c_lcone_solve`),
 * with the same argument order, layouts, sentinels and error behaviour.
 *
 * Part 2 is the device-resident extension (no reference counterpart): the same
 * solve with every buffer already in HBM, a persistent workspace, an explicit HIP
 * stream, particle sharding over RCCL and the on-device dynamics linearisation
 * that replaces the user's f_fx_fu_fn callback (pmpc/scp_mpc.py:338-342) for the
 * built-in models.
 *
 * Layouts (column-major, PMPC.jl/src/c_interface.jl:28-46):
 *   x0 (xdim,M)   f, X_prev, X_ref, lx, ux (xdim,N,M)   U_prev, U_ref, lu, uu (udim,N,M)
 *   fx, Q (xdim,xdim,N,M)   fu (xdim,udim,N,M)   R (udim,udim,N,M)
 *   slew_reg (M)   slew_reg0 (M)   slew_um1 (udim,M)
 *   X_out (xdim,N,M)  — steps 1..N, x0 is NOT included (c_interface.jl:138)
 *   U_out (udim,N,M)
 * i.e. particle-major, time-minor arrays of column-major blocks; equivalently the
 * C-order numpy arrays (M,N,d) and (M,N,col,row).
 *
 * Sentinels (c_interface.jl:56-70): any NaN in lx|ux => no state bounds; any NaN in
 * lu|uu => no control bounds; any NaN in slew_reg => 0; any NaN in slew_reg0 or
 * slew_um1 => both ignored.  Nc < 0 => Nc = N (PMPC.jl/src/main.jl:127-128).
 * +-inf entries of a bound array mean "unbounded on that side" (as for OSQP).
 *
 * Errors: void return; on failure X_out/U_out are filled with NaN
 * (PMPC.jl/src/osqp_solver.jl:65-71) which the Python host turns into
 * (None, None, None) (pmpc/scp_mpc.py:391-394).
 *
 * Threading: one caller thread at a time per process (as the reference, which holds
 * the GIL for the whole call, module.cpp:28-72).  Blocking.
 */
#ifndef PMPC_ABI_H
#define PMPC_ABI_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------------------------------------
 * Part 1 — drop-in boundary (host pointers)
 * ------------------------------------------------------------------------------------------- */

/* replaces PMPC.jl/src/c_interface.jl:77-141 (prototype: PMPC.jl/pmpcjl/module.cpp:9-15) */
void c_lqp_solve(double *X_out, double *U_out, size_t xdim, size_t udim, size_t N, size_t M, long long Nc,
                 double *x0, double *f, double *fx, double *fu, double *X_prev, double *U_prev, double *Q,
                 double *R, double *X_ref, double *U_ref, double *lx, double *ux, double *lu, double *uu,
                 double reg_x, double reg_u, double *slew_reg, double *slew_reg0, double *slew_um1,
                 long long verbose);

/* replaces PMPC.jl/src/c_interface.jl:146-214 (prototype: module.cpp:17-23).
 * Minimises the reference's epsilon-anchored epigraph objective (main.jl:204-238, k = M) — see
 * pmpc_lcone_solve_device below.  smooth_alpha = NaN => hard boxes (main.jl:242-244); finite => the boxes enter as
 * -1/alpha sum log(alpha slack) (smooth_cstr = "logbarrier", main.jl:246-262).  `solver` ("ecos" | "cosmo" | "mosek" |
 * "gurobi") only selects the conic back end upstream; all share one optimum (DESIGN.md section 2 notes the one
 * exception, the reference's exponential-cone row order under "ecos", which is not reproduced). */
void c_lcone_solve(double *X_out, double *U_out, size_t xdim, size_t udim, size_t N, size_t M, long long Nc,
                   double *x0, double *f, double *fx, double *fu, double *X_prev, double *U_prev, double *Q,
                   double *R, double *X_ref, double *U_ref, double *lx, double *ux, double *lu, double *uu,
                   double reg_x, double reg_u, double *slew_reg, double *slew_reg0, double *slew_um1,
                   long long verbose, double smooth_alpha, char *solver);

/* Extensions (not in the reference): the two entry points above for callers that hold fx, fu, Q, R as ROW-major blocks —
 * numpy's C-ordered (M, N, row, col) stacks, which the reference's Python side transposes on the host on every call to
 * reach the column-major layout (pmpc/static_backend.py:83-101 via pybind11's f_style cast: ~630 MB at M = 4096, the
 * dominant cost of the call once the solve runs on the GPU).  Bit k of `rowmajor` marks fx (0), fu (1), Q (2), R (3) as
 * row-major; everything else is as above.  The transposition is done in HBM after the upload. */
void pmpc_lqp_solve_host(double *X_out, double *U_out, size_t xdim, size_t udim, size_t N, size_t M, long long Nc,
                         double *x0, double *f, double *fx, double *fu, double *X_prev, double *U_prev, double *Q,
                         double *R, double *X_ref, double *U_ref, double *lx, double *ux, double *lu, double *uu,
                         double reg_x, double reg_u, double *slew_reg, double *slew_reg0, double *slew_um1,
                         long long verbose, unsigned rowmajor);
void pmpc_lcone_solve_host(double *X_out, double *U_out, size_t xdim, size_t udim, size_t N, size_t M, long long Nc,
                           double *x0, double *f, double *fx, double *fu, double *X_prev, double *U_prev, double *Q,
                           double *R, double *X_ref, double *U_ref, double *lx, double *ux, double *lu, double *uu,
                           double reg_x, double reg_u, double *slew_reg, double *slew_reg0, double *slew_um1,
                           long long verbose, double smooth_alpha, unsigned rowmajor, long long cone_k);
/* (cone_k: the reference's `k` setting, see pmpc_problem.cone_k; <= 0 = M) */
/* ... and with `smooth_cstr` (0 logbarrier, 1 squareplus) / `smooth_beta` (see pmpc_problem.smooth_cstr) */
void pmpc_lcone_solve_host_ex(double *X_out, double *U_out, size_t xdim, size_t udim, size_t N, size_t M, long long Nc, double *x0, double *f,
                              double *fx, double *fu, double *X_prev, double *U_prev, double *Q, double *R, double *X_ref, double *U_ref, double *lx,
                              double *ux, double *lu, double *uu, double reg_x, double reg_u, double *slew_reg, double *slew_reg0, double *slew_um1,
                              long long verbose, double smooth_alpha, unsigned rowmajor, long long cone_k, int smooth_cstr, double smooth_beta);

/* ---------------------------------------------------------------------------------------------
 * Part 2 — device-resident extension
 * ------------------------------------------------------------------------------------------- */

/* flags for pmpc_problem.flags */
#define PMPC_HAS_XBOUNDS 1u  /* lx/ux valid (no NaN sentinel) */
#define PMPC_HAS_UBOUNDS 2u  /* lu/uu valid */
#define PMPC_HAS_SLEW 4u     /* slew_reg valid (else treated as 0) */
#define PMPC_HAS_SLEW0 8u    /* slew_reg0 and slew_um1 valid */
#define PMPC_FORCE_GENERIC 16u /* debugging: never take the MFMA fast path */
#define PMPC_SYMMETRIC_COST 32u /* caller guarantees Q_j == Q_j' and R_j == R_j' exactly (OSQP keeps
                                   triu(P); the register-resident path relies on symmetric blocks) */

#define PMPC_COLD_START 64u /* do not start the interior-point iteration from the iterate remembered from the previous
                               solve of the same shape (see "warm start" in DESIGN.md section 2) */
#define PMPC_STATIC_CONS_BOUNDS 128u /* the caller guarantees that the CONTENTS of lu / uu (same device arrays) are those of the
                                       previous solve of this shape, as inside an SCP loop (pmpc/scp_mpc.py passes the same
                                       u_l / u_u every iteration).  The library then reuses (a) its working copy of the control
                                       boxes (the caller's, with global particle 0's on the consensus stages) and (b), sharded,
                                       the consensus bounds it broadcast then instead of two collectives per solve.  This
                                       promise is NOT checked on the device: a caller that changes lu / uu in place between
                                       solves (a moving trust region) must leave the flag off. */
#define PMPC_PREV_IS_LAST_SOLUTION 256u /* the caller guarantees that X_prev / U_prev ARE the X_out / U_out of the previous solve
                                          of this shape, as in an SCP loop (pmpc/scp_mpc.py:313,430: X_prev, U_prev = X, U).
                                          A warm-started solve then takes the linearisation point itself as its first base
                                          point — its dynamics defect is f - X_prev, elementwise — instead of rolling the old
                                          controls out again (one sequential sweep less).  Checked on the device (U_prev must be
                                          the base point of the stored set under the boxes this solve uses): a broken promise
                                          costs time, not correctness. */

#define PMPC_CONE_OBJECTIVE 1024u /* pmpc_scp_loop_device only: the sub-problem of every iteration is the reference's DEFAULT solver path
                                    (solver = "ecos" -> c_lcone_solve, pmpc/static_backend.py:242-253: the eps-anchored epigraph objective of
                                    PMPC.jl/src/main.jl:204-238; barrier_mu > 0: log-barrier smoothing with alpha = 1 / barrier_mu) instead of
                                    the QP of c_lqp_solve */
#define PMPC_F32_MATRICES 512u /* fp32-STORAGE mode (BASELINE config E's "fp32"; the reference is fp64-only, c_interface.jl:6-25): fx, fu,
                                 Q, R point to FLOAT arrays of the same layouts (cast to const double * in the struct).  The
                                 active-set sweeps of an SCP loop's warm-started solves (control boxes and / or stage cones,
                                 Nc <= 1) then stream half the bytes of the dominant arrays and keep their factor records in
                                 fp32; every value is widened on load and ALL arithmetic stays fp64, so the solve is the exact
                                 solve of the fp32-rounded data.  Every other path (cold start, fallbacks, other dims) widens the
                                 four arrays into workspace copies first and runs the fp64 kernels. */

typedef struct pmpc_problem {
  size_t xdim, udim, N, M; /* M = particles held by THIS rank */
  long long Nc;
  unsigned flags;
  double reg_x, reg_u;
  /* device pointers, ABI layouts above; unused ones may be NULL */
  const double *x0, *f, *fx, *fu, *X_prev, *U_prev, *Q, *R, *X_ref, *U_ref;
  const double *lx, *ux, *lu, *uu;
  const double *slew_reg, *slew_reg0, *slew_um1;
  double *X_out, *U_out; /* device, (xdim,N,M) / (udim,N,M) */
  /* optional per-particle cost weights, device (M): minimise sum_i weights_i J_i (the reference's `weights`
   * setting, PMPC.jl/src/main.jl:96-112, and the building block of the cone objective); NULL = all 1 */
  const double *weights;
  /* > 0: log-barrier smoothing of the boxes (the cone path's smooth_cstr = "logbarrier", cone_utils.jl:173-232): the
   * hard boxes are replaced by -barrier_mu * sum log(slack) in the objective, barrier_mu = 1/smooth_alpha; 0 = hard */
  double barrier_mu;
  /* optional second-order cone on the controls of EVERY (particle, stage):  || W u + w0 ||_2 <= v'u + v0  — the
   * structured case of the reference's pyjulia-only `extra_cstrs` (README.md:219-239; config E's thrust cones, SURVEY.md
   * section 8d), which cannot cross its C ABI.  soc_q = rows of W (0 = no cone); all four are DEVICE pointers, W (q x udim)
   * row-major.  soc_u_interior (device, udim): a control strictly inside the boxes and the cone (e.g. hover); the
   * path-following method starts from it on every stage.  Used by pmpc_lsoc_solve_device only. */
  size_t soc_q;
  const double *soc_W, *soc_w0, *soc_v;
  double soc_v0;
  const double *soc_u_interior;
  /* General form of the stage cones (pmpc_lsoc_solve_device; the structured case of the reference's `extra_cstrs` tuples,
   * PMPC.jl/src/main.jl:293-316, that stays inside ONE stage's controls): `cone_count` cones on the controls of every
   * (particle, stage); cone k has cone_sizes[k] + 1 rows — cone_sizes[k] = 0: a linear row s >= 0 (several of them: a
   * stage-wise polytope), >= 1: the second-order cone |s[1..]| <= s[0] —, s = A u + c with the rows of all cones stacked:
   * cone_A (rows x udim, row-major), cone_c (rows), both DEVICE pointers.  cone_per_stage = 0: one (A, c) for every stage;
   * 1: per (particle, stage) data, (M, N, rows, udim) / (M, N, rows) — the consensus stages use particle 0's.  cone_sizes is a
   * HOST array.  At most 4 cones, 8 rows, cone size <= 3.  cone_count = 0: the single cone soc_* above.  Solved by the
   * active-set rounds (kernels_cone.hip) from soc_u_interior (or from U_prev if that is NULL); the general form has no
   * path-following fallback: rounds that do not settle are a failed solve (status 1). */
  size_t cone_count;
  const int *cone_sizes;
  const double *cone_A, *cone_c;
  int cone_per_stage;
  /* cone path only (pmpc_lcone_solve_device): the reference's `k` setting (PMPC.jl/src/main.jl:204-227), the weight
   * (1 - eps) k of the epigraph offset t in  (1 + eps) sum_i y_i + (1 - eps) k t,  J_i <= y_i + t,  y >= 0.  k = M (the
   * default, and the only value its C ABI reaches) is the sum of the particle costs up to the eps-anchoring; k < M is a
   * worst-k objective: only about k (1 - eps) / (1 + eps) costliest particles carry weight.  <= 0 or >= M (all ranks' particles): k = M. */
  long long cone_k;
  /* cone path with smoothing (barrier_mu > 0 = 1 / smooth_alpha): the reference's `smooth_cstr` setting (PMPC.jl/src/main.jl:247-279) —
   * 0 "logbarrier" (the default): -(1/alpha) log(alpha slack) per box side; 1 "squareplus": soft boxes, every side a'z <= b costs
   * beta/2 (v + sqrt(v^2 + 1/alpha^2)), v = a'z - b (smooth_beta = beta, cone_utils.jl:222-228).  Pyjulia-only upstream. */
  int smooth_cstr;
  double smooth_beta;
} pmpc_problem;

typedef struct pmpc_info {
  int status;          /* 0 ok, 1 not converged (outputs NaN), 2 numerical failure (outputs NaN) */
  int ipm_iters;       /* 0 = the equality-only optimum was feasible */
  int structured_solves; /* Riccati factorisations performed */
  int fast_path;       /* 1 = MFMA register-resident kernels were used; 2 = those kernels on fp32-stored matrices (PMPC_F32_MATRICES) */
  double mu;           /* final complementarity */
  double slack_res;    /* final slack residual (inf-norm) */
  double max_violation;/* bound violation of the equality-only optimum */
  int outer_solves;    /* cone path: weighted QPs solved (active-set + bisection on the threshold particle) */
  int active_set_rounds; /* structured solves spent in the active-set finish of the box interior-point iteration */
} pmpc_info;

/* Opaque solver context: owns the HIP stream, the workspace cache keyed on
 * (xdim,udim,N,M,Nc,flags) and the optional RCCL communicator. */
typedef struct pmpc_ctx pmpc_ctx;

int pmpc_create(pmpc_ctx **ctx, int device);          /* 0 on success */
void pmpc_destroy(pmpc_ctx *ctx);
/* Per-context algorithm switches (0 on success, -1 unknown key).  Every key has an environment variable of the same meaning that
 * sets its DEFAULT when a context is created (so a process-wide override still works): key / variable / default —
 *   as_warm PMPC_AS_WARM 1             warm start of the active-set rounds from the previous solve's set
 *   as_skip PMPC_AS_SKIP 1             settled particles skip the factor sweep of the later rounds
 *   as_defect PMPC_AS_DEFECT 1         no-rollout warm start under PMPC_PREV_IS_LAST_SOLUTION
 *   as_cold_rounds PMPC_AS_COLD 10     rounds of the cold start (0: interior-point iteration at once)
 *   polish_mu PMPC_POLISH_MU 1e-3      relative complementarity at which the interior-point iteration tries the rounds (0: rounds off altogether)
 *   warm_start PMPC_WARM_START 1       interior-point warm start from the remembered early iterate
 *   cone_as PMPC_CONE_AS 1             stage cones inside the rounds (0: path-following iteration)
 *   cone_cold_rounds PMPC_CONE_AS_COLD 16
 *   xbox_as PMPC_XBOX_AS 1             state boxes inside the rounds (0: a binding one sends the solve to the interior-point iteration)
 *   slew_increment_boxes PMPC_SLEW_INCREMENT_BOXES 1   boxed slew problems in increment form on the MFMA path (needs xbox_as)
 *   as_fuse_ctl PMPC_AS_FUSE_CTL 1, as_wave_cons PMPC_AS_WAVE_CONS 1   launch fusions of the rounds (measurement legs)
 *   host_reuse PMPC_HOST_REUSE 1       host ABI: unchanged 8 MB chunks are not uploaded again
 *   warn_slow_path PMPC_WARN_SLOW_PATH 1   one line on stderr when a context first leaves the register-resident path
 *   cone_rank_memory PMPC_CONE_RANK_MEMORY 1   cone objective: the weight assignment the previous solve of the shape settled on is tried first
 *   cone_epigraph    PMPC_CONE_EPIGRAPH    1   cone objective: hard boxes through the epigraph problem in the shared-control space, smoothing through the
 *                                             full-space Newton iteration (any tie pattern); 0: the rank-based weighted-QP iteration everywhere
 *   cond_grouped     PMPC_COND_GROUPED     1   Nc > 1: the condensed consensus Hessians are summed over groups of 8 particles inside the condensing
 *                                             kernel (no per-particle block travels through HBM) whenever nothing downstream needs one particle's block
 *   as_freeze_tol    PMPC_AS_FREEZE_TOL    1e-9  stage-cone rounds with one consensus stage: a step of the free shared controls below this (relative)
 *                                             is taken as zero by every particle, and the settled particles leave the forward sweep at once (0: off)
 *   as_ckpt          PMPC_AS_CKPT          1   the factor sweeps leave their cost-to-go at stages 4, 8, 16, 32, ..; an unsettled particle's factor sweep of a
 *                                             later round starts at the lowest of them at or above its highest changed stage (0: from the terminal cost)
 *   as_sens_min_m    PMPC_AS_SENS_MIN_M    3072  one consensus stage, at least this many particles on the rank: the forward sweep records the sensitivity
 *                                             of every stage to the shared-control step, and settled particles of the later rounds are updated elementwise
 *                                             from it instead of being swept again (0: never)
 *   as_perm_min_m    PMPC_AS_PERM_MIN_M    2048  from that many particles per rank a later round's launches take the UNSETTLED particles first: their long sweeps
 *                                             then spread one per SIMD instead of piling up where their indices fall (0: never)
 *   cone_path        PMPC_CONE_PATH        0   cone objective with hard boxes: which body answers.  0: automatic (the order depends on what the context learnt
 *                                             about the shape: free-particles body first where it worked before, else epigraph path, rank-based iteration,
 *                                             free-particles body last); 1: free-particles body first; 2: epigraph path, free-particles body never; 3: the
 *                                             rank-based weighted-QP iteration alone (two costs on the threshold at most)
 * Setting an option forgets the context's warm-start memory.  The reference has no counterpart (its solver settings travel in
 * `solver_settings`, pmpc/scp_mpc.py:45-66, and never reach the C ABI); kernel launch heuristics stay environment-only. */
int pmpc_set_option(pmpc_ctx *ctx, const char *key, double value);
int pmpc_get_option(pmpc_ctx *ctx, const char *key, double *value);
void *pmpc_stream(pmpc_ctx *ctx);                     /* hipStream_t the solver launches on */
void pmpc_sync(pmpc_ctx *ctx);                        /* hipStreamSynchronize(pmpc_stream) */

/* Solve with all buffers in HBM.  Asynchronous on pmpc_stream() except for one small
 * device->host read per IPM iteration.  Returns pmpc_info.status. */
int pmpc_lqp_solve_device(pmpc_ctx *ctx, const pmpc_problem *prob, pmpc_info *info, int verbose);

/* Cone-path objective of c_lcone_solve with every buffer in HBM (PMPC.jl/src/main.jl:194-354 through the C
 * ABI: k = M, no extra_cstrs): min (1+eps) sum_i y_i + (1-eps) M t  s.t.  J_i <= y_i + t, y >= 0, dynamics, boxes,
 * eps = 1e-3, J_i the particle cost of qp_utils.jl:60-162.  Eliminating (y, t) gives sum_i w_i J_i with w = 1+eps
 * above the threshold particle(s) — solved as a short sequence of weighted QPs.  smooth_alpha = NaN: hard boxes.
 * Sharded runs: equal particle counts per rank; the costs are gathered so that every rank ranks them identically. */
int pmpc_lcone_solve_device(pmpc_ctx *ctx, const pmpc_problem *prob, double smooth_alpha, pmpc_info *info, int verbose);

/* The QP of pmpc_lqp_solve_device plus the stage-wise control cones of pmpc_problem.soc_* (control boxes allowed, state
 * boxes / weights / slew not): feasible primal-dual path following with Nesterov-Todd scaling on the same Riccati
 * kernels, complementarity driven to 1e-12.  Returns status (0 ok, 1 not converged, 2 numerical failure, 3 infeasible
 * start: soc_u_interior is not strictly inside the boxes and the cone). */
int pmpc_lsoc_solve_device(pmpc_ctx *ctx, const pmpc_problem *prob, pmpc_info *info, int verbose);

/* J_out[i] (device, M) = J_i(X, U): per-particle cost of qp_utils.jl:60-162 (1/2 z'Pz + q'z + r), unweighted. */
int pmpc_particle_costs_device(pmpc_ctx *ctx, const pmpc_problem *prob, const double *X, const double *U, double *J_out);

/* Particle sharding over RCCL (one process per GPU).  unique_id is the 128-byte ncclUniqueId
 * produced by pmpc_comm_unique_id on rank 0 and distributed by the host (torch.distributed).
 * With a communicator set, pmpc_lqp_solve_device treats `M` as the local shard and
 * all-reduces the consensus Hessian/gradient and the IPM scalars. */
int pmpc_comm_unique_id(void *unique_id_128);
int pmpc_comm_init(pmpc_ctx *ctx, int rank, int world, const void *unique_id_128);
/* TEST HOOK (tests/test_multirank_gpu.py): an in-process stand-in for the communicator — `world` contexts on ONE device, one
 * host thread each, collectives staged through host memory — so that the world > 1 code paths run on a single-GPU box
 * (RCCL refuses two ranks on one device).  Not for production use. */
int pmpc_comm_init_mock(pmpc_ctx *ctx, int rank, int world, int group);
int pmpc_comm_rank(pmpc_ctx *ctx);
int pmpc_comm_world(pmpc_ctx *ctx);

/* On-device dynamics linearisation (row a2 of SURVEY.md §8: the f_fx_fu_fn callback for the
 * built-in models).  model: 0 = unicycle (reference tests/dubins_car.py:48-90, params (3,M)),
 * 1 = synthetic quadrotor (params (4,M)).  X_prev/U_prev/x0 and outputs in ABI layout. */
int pmpc_linearize_device(pmpc_ctx *ctx, int model, size_t N, size_t M, const double *x0, const double *X_prev,
                          const double *U_prev, const double *params, double *f, double *fx, double *fu);

/* the same with fx / fu written as FLOAT arrays (PMPC_F32_MATRICES) */
int pmpc_linearize_device_f32(pmpc_ctx *ctx, int model, size_t N, size_t M, const double *x0, const double *X_prev,
                              const double *U_prev, const double *params, double *f, float *fx, float *fu);

/* SCP residual of one iteration (pmpc/scp_mpc.py:397-403): *out (device, one double) = max over particles and stages of
 * ||X - X_prev||_2 and ||U - U_prev||_2 (inf if a trajectory holds a NaN); asynchronous on pmpc_stream(). */
int pmpc_scp_residual_device(pmpc_ctx *ctx, size_t xdim, size_t udim, size_t N, size_t M, const double *X, const double *X_prev,
                             const double *U, const double *U_prev, double *out);

/* `steps` iterations of the SCP loop (pmpc/scp_mpc.py:337-430) for a built-in dynamics model with the host out of the loop
 * body: per iteration  linearise about (X_prev, U_prev) -> convex sub-problem (pmpc_lqp_solve_device semantics, or
 * pmpc_lsoc_solve_device if p->soc_u_interior is set) -> SCP residual -> (X_prev, U_prev) <- (X, U).
 *   p          the sub-problem; p->X_prev / p->U_prev hold the start iterate (and are OVERWRITTEN: the two trajectory buffer
 *              pairs (X_prev, U_prev) and (X_out, U_out) swap roles every iteration), p->f / fx / fu are scratch the
 *              linearisation writes (device, caller-allocated), f2 / fx2 / fu2 a second set of the same sizes: the next
 *              linearisation is enqueued behind the sub-problem's rounds BEFORE the host knows they sufficed (if they did
 *              not, it is redone), and must not touch what a continued solve still reads
 *   first_cold non-zero: X_prev / U_prev of the first iteration are not the previous solve's outputs (no warm-start promise)
 *   res        device, `steps` doubles: the residual of each iteration (max over ranks when sharded)
 *   infos      host, `steps` entries (may be NULL)
 * The final iterate is in (X_out, U_out) if `steps` is odd, else in (X_prev, U_prev); *last_in_out says which.  Returns the
 * number of iterations completed (== steps unless a sub-problem failed: its status is in infos[returned]). */
int pmpc_scp_loop_device(pmpc_ctx *ctx, int model, const double *params, const pmpc_problem *p, double *f2, double *fx2, double *fu2,
                         int steps, int first_cold, double *res, pmpc_info *infos, int *last_in_out);

/* Live kernel timing for bench.py: HIP events on pmpc_stream() around the launches of a class
 * (0 backward+factor, 1 backward vector-only, 2 forward, 3 consensus reduce+solve).
 * level 0 = off, 1 = class 0 only (the dominant kernel; what bench.py's roofline needs), 2 = every class
 * (each event pair costs a few microseconds of launch gap). */
void pmpc_profile_enable(pmpc_ctx *ctx, int level);
void pmpc_profile_read(pmpc_ctx *ctx, double *ms4, long long *n4);
/* Factor sweeps of active-set rounds that skip the settled particles are timed in a class of their own (level 2 only) and
 * never enter class 0, whose launches all process every (particle, stage): totals as of the last pmpc_profile_read. */
void pmpc_profile_read_partial(pmpc_ctx *ctx, double *ms, long long *n);
/* every launch class as of the last pmpc_profile_read: 0 full factor sweep, 1 vector sweep, 2 forward sweep, 3 consensus
 * reduce + solve, 4 factor sweeps that skip settled particles, 5 active-set bookkeeping, 6 linearisation, 7 SCP residual */
void pmpc_profile_read_all(pmpc_ctx *ctx, double *ms, long long *launches, int count);

/* Checkpointed restart of the later rounds' factor sweeps (option as_ckpt), counted since the last reset: out4 = {sweeps that
 * started from a checkpoint, stages they ran, sweeps of unsettled particles that started from the terminal cost, stages they ran}.
 * Synchronises the solver's stream.  For tests and bench.py (how much of the horizon the later rounds re-factorise). */
void pmpc_restart_stats(pmpc_ctx *ctx, unsigned long long *out4, int reset);

/* version / build probe used by the loader and the tests */
const char *pmpc_version(void);
/* sizeof(pmpc_problem), sizeof(pmpc_info) as the library was built: a binding that mirrors the structs compares them with its
 * own at load time and refuses a mismatch (a stale .so under a newer binding silently misreads the trailing fields) */
void pmpc_abi_struct_sizes(size_t *problem, size_t *info);

#ifdef __cplusplus
}
#endif
#endif /* PMPC_ABI_H */
