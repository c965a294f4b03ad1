// solver_internal.h — what the translation units of the host side share: the context (pmpc_ctx), its workspace, the option table's
// indices, and the helpers every solve path uses (collectives, device-published scalars, the structured Newton solve).
//   solver.hip       contexts / options / communicators / profiling, the QP path (solve_impl_body, slew increment form), the SCP loop
//   solver_cone.hip  the cone objective (c_lcone_solve semantics): hard boxes, smoothed boxes, free particles, particle costs
//   solver_host.hip  the host-pointer drop-in entry points (c_lqp_solve, c_lcone_solve and their extensions)
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <map>
#include <atomic>
#include <mutex>
#include <thread>
#include <limits>
#include <vector>

#include "../../include/pmpc_abi.h"
#include "pmpc_dev.h"

namespace pmpc_impl {

struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
  bool ensure(size_t b) {  // true: the buffer was (re)allocated — whatever it held is gone
    if (b <= bytes) return false;
    if (p) HIP_WARN(hipFree(p));
    p = nullptr;
    bytes = 0;
    HIP_CHECK(hipMalloc(&p, b ? b : 8));  // (throws on failure, e.g. out of memory at a large M: the solve returns status 2)
    bytes = b;
    return true;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  double *d() const { return (double *)p; }
};

struct SlabBufs {
  DevBuf lo, hi, tl, tu, ll, lu, cl, cu, D, w;
};

struct Workspace {
  DevBuf X, U, dX, dU, dX2, dU2, xm, xd, um, ud, K, Hinv, kff, gc_part, Hc_part, scratch, red_tmp, Hg /* [Hc | gc] */, Lc, duc;
  DevBuf Hc_w, gc_w, cons_w, epi_lam, epi_out, epi_gath;
  DevBuf Hc_grp;
  DevBuf es_Dx, es_wx, es_Du, es_wu, es_xm, es_xd, es_um, es_ud, es_kff2, es_kff3, es_gc2, es_dots, es_coef, es_out2, es_Xt, es_Ut, es_U, es_zero;  // smoothed cone objective (lcone_smooth_body)
  long long es_key = -1;  // consensus weights of the cone objective and the scaled copies the reductions read
  DevBuf xch, zeros, zslew, zslew0, zum1, part_sum, part_cnt, part_max, sc, fail;
  DevBuf pw, Jc, Jg;  // cone path: particle weights / particle costs (local, gathered)
  DevBuf soc_zl, soc_zu, soc_zc, soc_dzl, soc_dzu, soc_dzc, soc_sl, soc_su, soc_sc, soc_dsl, soc_dsu, soc_dsc;
  DevBuf soc_cl, soc_cu, soc_cc;  // second-order (Mehrotra) terms
  DevBuf soc_wU, soc_wzl, soc_wzu, soc_wzc;  // remembered early iterate (warm start of the cone path)
  long long soc_key = -1;
  DevBuf Hadd, wu_soc;  // stage-cone extension: control Hessian blocks A'W^-2 A, gradient shift (path following) / Newton terms of the cones (active-set rounds)
  DevBuf cone_A, cone_c, cone_z, cone_rec, cone_uraw, as_open;  // stage cones inside the active-set rounds (kernels_cone.hip)
  DevBuf xb_qmax;  // per particle: largest diagonal cost entry (penalty scale of the state rows), found once per attempt
  DevBuf xb_z, xb_st, xb_D, xb_g;  // state boxes inside the active-set rounds (kernels_xbox.hip)
  DevBuf m64[4];  // fp32-storage mode: fx, fu, Q, R widened for the paths that run the fp64 kernels
  // warm start: the early interior-point iterate (mu <= 0.5) remembered from the previous solve of the same shape
  DevBuf warmU, warm_llu, warm_luu, warm_llx, warm_lux;
  DevBuf lateX, lateU;  // last interior-point iterate with mu <= 1e-10 mu_peak and small residuals (kept against a numerical breakdown at mu ~ 1e-12)
  long long warm_key = -1;
  double warm_mu = 0.0;  // barrier parameter the remembered iterate belongs to (0: the early iterate of a hard-constrained solve)
  DevBuf as_perm;  // later rounds: particle order of the launches (unsettled first); as_perm_m: the M it is a permutation for
  int as_perm_m = -1;
  DevBuf as_T;  // forward sweep's sensitivity records (one consensus stage: settled particles of the later rounds are updated elementwise)
  DevBuf as_ck, as_jhi, ck_stat;  // checkpoints of the factor sweeps' cost-to-go + highest changed stage per particle (restart of the later rounds' sweeps)
  DevBuf as_act, as_cnt, as_cntp, as_settled, as_ctl, as_delta, as_viol;  // active-set iteration: status per bounded control (int), counters,
                                                                 // per-particle counters, settled flags, control block, applied consensus step
  long long su_key = -1;  // shape / source arrays the working copy of the control boxes (w.su.lo, w.su.hi) was made for
  const double *su_src_lo = nullptr, *su_src_hi = nullptr;
  bool as_U_valid = false;  // w.U holds the solution that goes with the stored active set
  int as_pred_rounds = 3;  // rounds the last accepted solve took: how many the next one enqueues before it reads anything back
  DevBuf cons_lo, cons_hi;  // sharded runs: the consensus controls' bounds as last broadcast (PMPC_STATIC_CONS_BOUNDS)
  long long cons_key = -1;
  long long xb_block_key = -1;  // shape whose state boxes were found active: no active-set attempts for it
  int xb_warm_backoff = 0, xb_warm_fails = 0;  // state rows: solves left for which the warm start is not tried / its failures in a row
  long long as_key = -1;  // shape whose accepted active set (as_act) and solution (U) can start the next solve
  double as_scale = 1.0;
  DevBuf part_dev;  // barrier mode: block partials of the centrality deviation
  DevBuf sa_f, sa_fx, sa_fu, sa_Xp, sa_Up, sa_Q, sa_R, sa_Xr, sa_Ur, sa_lo, sa_hi, sa_Xo, sa_Uo, sa_cl, sa_ch;  // slew: increment form
  SlabBufs sx, su;
};


// Per-context algorithm switches (pmpc_set_option / pmpc_get_option, include/pmpc_abi.h).  Each has an environment variable that
// sets its DEFAULT when a context is created — so a process-wide override still works, and two contexts of one process (or a test
// that flips a switch) no longer depend on what the first solve of the process happened to read.
enum PmpcOpt {
  OPT_AS_WARM, OPT_AS_SKIP, OPT_AS_DEFECT, OPT_AS_COLD_ROUNDS, OPT_POLISH_MU, OPT_WARM_START, OPT_CONE_AS, OPT_CONE_COLD_ROUNDS, OPT_XBOX_AS,
  OPT_SLEW_INCREMENT_BOXES, OPT_AS_FUSE_CTL, OPT_AS_WAVE_CONS, OPT_HOST_REUSE, OPT_WARN_SLOW_PATH, OPT_CONE_RANK_MEMORY, OPT_CONE_EPIGRAPH, OPT_COND_GROUPED, OPT_AS_FREEZE_TOL, OPT_AS_CKPT, OPT_AS_SENS_MIN_M, OPT_AS_PERM_MIN_M, OPT_CONE_PATH, OPT_COUNT
};
}  // namespace pmpc_impl
using namespace pmpc_impl;

struct ProfCat {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending, pool;
  double ms = 0.0;
  long long n = 0;
};

struct pmpc_ctx {
  double opt[OPT_COUNT];
  bool warned_slow_path = false;
  std::vector<double> cone_rw;  // cone objective: the weight assignment (by cost rank) the last solve settled on, and what it belongs to
  long long cone_rw_key = -1;
  long long fp_key = -1;   // cone objective, free-particles path: the shape it was last tried on ...
  int fp_ok = -1;          // ... and whether its assumption held there (0: a particle's own box was violated — not tried again on that shape)
  std::vector<double> cone_lam;  // cone objective in the shared-control space: multipliers of the epigraph rows the last solve settled on
  long long cone_lam_key = -1;
  double cone_rho = 0.0;  // proximal parameter the smoothed cone objective ended with (next solve of the shape starts there)
  std::vector<double> cons_w_host, epi_lam_host;  // what the device copies of the consensus weights / multipliers hold
  const double *cons_w_active = nullptr;  // consensus weights of the sub-problem solves lcone_body issues (LQArgs::cons_w)
  int xb_ctrl_from = -1;  // set around the inner solve of the slew increment form: state entries from this index on are the controls (their boxes the control boxes)
  AsCtlCall as_pend{};  // round control of the previous active-set round, to ride in the next consensus-partials launch (structured_solve)
  int prof = 0;  // 0 off, 1 dominant kernel (factor sweep) only, 2 every launch class
  double partial_ms = 0.0;  // class 4 of the last pmpc_profile_read
  long long partial_n = 0;
  double last_ms[8] = {0};  // every class of the last pmpc_profile_read (pmpc_profile_read_all)
  long long last_n[8] = {0};
  ProfCat cat[8];  // 0 backward+factor (all particles), 1 backward vector-only, 2 forward, 3 consensus reduce+solve,
                   // 4 backward+factor of an active-set round that skips the settled particles (never part of the roofline figure),
                   // 5 active-set bookkeeping (first base point, round control), 6 on-device linearisation, 7 SCP residual
  int device = 0;
  hipStream_t stream = nullptr;
  Workspace ws;
  IpmScal *sc_host = nullptr;  // host snapshot of the device scalars
  int *fail_host = nullptr;
  // host-coherent mapped mirror the exchange kernel publishes into (zero-copy; the host polls `seq`)
  struct ScMirror { IpmScal sc; unsigned long long seq; int as_cnt[4]; unsigned long long as_seq; AsCtl ctl; double epi[4]; unsigned long long epi_seq; };
  ScMirror *mirror = nullptr, *mirror_dev = nullptr;
  unsigned long long seq = 0, as_seq = 0, epi_seq = 0;
  // RCCL
  ncclComm_t comm = nullptr;
  bool mock_comm = false;  // comm is a MockRank (test hook), not an RCCL communicator
  int rank = 0, world = 1;
  bool single_rank_comm = false;  // a real 1-rank RCCL communicator drives the multi-rank code paths (PMPC_RCCL_SINGLE, test hook)
  bool multi() const { return world > 1 || single_rank_comm; }
  // staging for the host-pointer ABI
  DevBuf stage[19], stage_t[4];
  void *pinned = nullptr;  // host-coherent bounce buffer of the host-pointer ABI (threaded memcpy -> DMA)
  void *sm_pinned = nullptr;  // pinned host arrays of the smoothed cone objective's Newton iteration (its per-step gathers: pageable targets cost ~0.1 ms a step)
  size_t sm_pinned_bytes = 0;
  size_t pinned_bytes = 0;
  struct StagedChunk { void *dst; size_t bytes, off; };
  std::vector<StagedChunk> staged;  // what the bounce buffer (and the device staging buffers) hold from the previous call
  DevBuf host_flags;
  // pmpc_scp_loop_device: work to enqueue right behind the first batch of active-set rounds, BEFORE the host waits for their
  // outcome (the residual of this iteration and the linearisation of the next); spec_ok: that batch was the whole solve
  std::function<void()> post_batch;
  bool spec_fired = false, spec_ok = false;
};

namespace pmpc_impl {

void allreduce(pmpc_ctx *c, void *buf, size_t n, ncclDataType_t dt, ncclRedOp_t op);
void broadcast(pmpc_ctx *c, void *buf, size_t n, ncclDataType_t dt, int root);

struct ProfScope {  // HIP events on the solver's own stream around one launch (bench.py's live kernel timing)
  pmpc_ctx *c;
  int k;
  std::pair<hipEvent_t, hipEvent_t> ev;
  bool on;
  ProfScope(pmpc_ctx *c_, int k_) : c(c_), k(k_), on(c_->prof >= 2 || (c_->prof == 1 && k_ == 0)) {  // (level 1: dominant kernel only)
    if (!on) return;
    ProfCat &pc = c->cat[k];
    if (pc.pool.empty()) {
      HIP_CHECK(hipEventCreate(&ev.first));
      HIP_CHECK(hipEventCreate(&ev.second));
    } else {
      ev = pc.pool.back();
      pc.pool.pop_back();
    }
    HIP_CHECK(hipEventRecord(ev.first, c->stream));
  }
  ~ProfScope() {
    if (!on) return;
    HIP_WARN(hipEventRecord(ev.second, c->stream));
    c->cat[k].pending.push_back(ev);
  }
};

// Poll a host-coherent word the device publishes into (a blit kernel + stream sync costs ~25 us of idle GPU per read).
// Bounded by WALL-CLOCK time: after `limit_s` seconds of polling (a hung or very slow device) the caller falls back to a
// stream synchronisation, which reports device errors.  The clock is read every 1024 polls only.
template <class Pred>
bool spin_until(Pred ready, double limit_s = 2.0) {
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    for (int k = 0; k < 1024; k++) {
      if (ready()) return true;
      __builtin_ia32_pause();
    }
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit_s) return ready();
  }
}
void wait_published(pmpc_ctx *c, volatile unsigned long long *seq, unsigned long long want);
void read_scalars(pmpc_ctx *c);
void exchange(pmpc_ctx *c, int phase);
void structured_solve(pmpc_ctx *c, LQArgs &a, bool factor, bool fast, bool prep_done = false);
int fail_after_error(pmpc_ctx *c, const pmpc_problem *p, pmpc_info *info);
void fill_nan_outputs(pmpc_ctx *c, const pmpc_problem *p);

}  // namespace pmpc_impl

// defined inside the extern "C" blocks of the translation units (C linkage names, C++ signatures): shared between them
extern "C" {
pmpc_problem widened_f32_problem(pmpc_ctx *c, const pmpc_problem *p, bool jacobians = true);  // solver.hip: fp32-stored matrix stacks widened for the fp64 paths
void build_slew_increment_problem(pmpc_ctx *c, const pmpc_problem *p, pmpc_problem &q, SlewAug &g);  // solver.hip: slew penalties restated in control increments
int lcone_body(pmpc_ctx *c, const pmpc_problem *p, double smooth_alpha, pmpc_info *info, int verbose);  // solver_cone.hip
}
