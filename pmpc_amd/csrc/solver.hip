// solver.hip — host orchestration of the device solver and the C ABI of include/pmpc_abi.h.
//
// One call = one convex sub-problem of the reference's SCP loop, i.e. what
// PMPC.jl/src/main.jl:115-171 `lqp_solve` does (assemble joint QP -> OSQP -> split), solved here as
//   1. equality-only optimum by ONE structured Newton step (Riccati + consensus condensing),
//   2. if a box constraint is violated: Mehrotra predictor-corrector on the boxes, each Newton
//      system being the same structured solve with modified diagonals.
// The only cross-particle (and therefore cross-GPU) data are the condensed consensus Hessian /
// gradient [Hc | gc] and a handful of IPM scalars -> RCCL all-reduce when a communicator is set.
#include <dlfcn.h>

#include "solver_internal.h"

namespace {

// ---- lazily bound RCCL (the library must load on machines where no communicator is ever made) ----
struct Rccl {
  void *h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  bool load() {
    if (h) return true;
    const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};  // torch bundles soname librccl.so: reuse the loaded copy
    for (const char *n : names)
      if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) return false;
    GetUniqueId = (decltype(GetUniqueId))dlsym(h, "ncclGetUniqueId");
    CommInitRank = (decltype(CommInitRank))dlsym(h, "ncclCommInitRank");
    CommDestroy = (decltype(CommDestroy))dlsym(h, "ncclCommDestroy");
    AllReduce = (decltype(AllReduce))dlsym(h, "ncclAllReduce");
    Broadcast = (decltype(Broadcast))dlsym(h, "ncclBroadcast");
    return GetUniqueId && CommInitRank && AllReduce && Broadcast;
  }
};
Rccl g_rccl;

// ---- in-process communicator (TEST HOOK, pmpc_comm_init_mock) --------------------------------------------------------
// RCCL refuses two ranks on one device, so on a single-GPU box the world > 1 code paths (packed scalar exchange, consensus
// all-reduce, owner / bounds broadcast) could never run.  This stand-in lets N contexts on ONE device, each driven by its
// own host thread, play the ranks: a collective synchronises the caller's stream, meets the other ranks at a barrier,
// reduces on the host in rank order and writes the result back.  Same call signature as the RCCL entry points it
// replaces; correctness only, no performance meaning.
struct MockGroup {
  int world = 0, arrived = 0, generation = 0;
  std::mutex m;
  std::condition_variable cv;
  std::vector<std::vector<unsigned char>> slot;  // one staging buffer per rank
  void barrier() {
    std::unique_lock<std::mutex> lk(m);
    const int gen = generation;
    if (++arrived == world) { arrived = 0; generation++; cv.notify_all(); }
    else cv.wait(lk, [&] { return gen != generation; });
  }
};
struct MockRank { MockGroup *g; int rank; };
std::map<int, MockGroup *> g_mock_groups;
std::mutex g_mock_mutex;

template <class T>
void mock_reduce(MockGroup *g, size_t n, ncclRedOp_t op, T *out) {
  for (size_t k = 0; k < n; k++) {
    T acc = ((const T *)g->slot[0].data())[k];
    for (int r = 1; r < g->world; r++) {
      const T v = ((const T *)g->slot[r].data())[k];
      acc = op == ncclSum ? (T)(acc + v) : (op == ncclMin ? std::min(acc, v) : std::max(acc, v));
    }
    out[k] = acc;
  }
}
ncclResult_t mock_allreduce(const void *send, void *recv, size_t n, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm, hipStream_t s) {
  MockRank *mr = (MockRank *)comm;
  MockGroup *g = mr->g;
  const size_t esz = dt == ncclFloat64 ? 8 : 4;
  HIP_CHECK(hipStreamSynchronize(s));
  g->slot[mr->rank].resize(n * esz);
  HIP_CHECK(hipMemcpy(g->slot[mr->rank].data(), send, n * esz, hipMemcpyDeviceToHost));
  g->barrier();
  std::vector<unsigned char> out(n * esz);
  if (dt == ncclFloat64) mock_reduce<double>(g, n, op, (double *)out.data());
  else mock_reduce<int>(g, n, op, (int *)out.data());
  g->barrier();  // everyone has read the slots before anyone overwrites them in the next collective
  HIP_CHECK(hipMemcpy(recv, out.data(), n * esz, hipMemcpyHostToDevice));
  return ncclSuccess;
}
ncclResult_t mock_broadcast(const void *send, void *recv, size_t n, ncclDataType_t dt, int root, ncclComm_t comm, hipStream_t s) {
  MockRank *mr = (MockRank *)comm;
  MockGroup *g = mr->g;
  const size_t esz = dt == ncclFloat64 ? 8 : 4;
  HIP_CHECK(hipStreamSynchronize(s));
  if (mr->rank == root) {
    g->slot[root].resize(n * esz);
    HIP_CHECK(hipMemcpy(g->slot[root].data(), send, n * esz, hipMemcpyDeviceToHost));
  }
  g->barrier();
  std::vector<unsigned char> out(g->slot[root].begin(), g->slot[root].begin() + n * esz);
  g->barrier();
  HIP_CHECK(hipMemcpy(recv, out.data(), n * esz, hipMemcpyHostToDevice));
  return ncclSuccess;
}

}  // namespace

static const struct { const char *key, *env; double dflt; } kPmpcOptions[OPT_COUNT] = {
    {"as_warm", "PMPC_AS_WARM", 1},                // warm start of the active-set rounds from the previous solve's set
    {"as_skip", "PMPC_AS_SKIP", 1},                // settled particles skip the factor sweep of the later rounds
    {"as_defect", "PMPC_AS_DEFECT", 1},            // no-rollout warm start under PMPC_PREV_IS_LAST_SOLUTION
    {"as_cold_rounds", "PMPC_AS_COLD", 10},        // rounds of the cold start (0: straight to the interior-point iteration)
    {"polish_mu", "PMPC_POLISH_MU", 1e-3},         // relative complementarity at which the interior-point iteration tries the rounds (0: never; also switches the warm / cold starts off)
    {"warm_start", "PMPC_WARM_START", 1},          // interior-point warm start from the remembered early iterate
    {"cone_as", "PMPC_CONE_AS", 1},                // stage cones inside the rounds (0: path-following iteration)
    {"cone_cold_rounds", "PMPC_CONE_AS_COLD", 16}, // rounds of the cone cold start
    {"xbox_as", "PMPC_XBOX_AS", 1},                // state boxes inside the rounds (0: interior-point iteration when one binds)
    {"slew_increment_boxes", "PMPC_SLEW_INCREMENT_BOXES", 1},  // boxed slew problems in increment form on the MFMA path (needs xbox_as)
    {"as_fuse_ctl", "PMPC_AS_FUSE_CTL", 1},        // round control rides in the next round's consensus-partials launch
    {"as_wave_cons", "PMPC_AS_WAVE_CONS", 1},      // consensus system solved by every wave of the forward sweep
    {"host_reuse", "PMPC_HOST_REUSE", 1},          // host ABI: unchanged 8 MB chunks are not uploaded again
    {"warn_slow_path", "PMPC_WARN_SLOW_PATH", 1},  // one line on stderr when a context first leaves the register-resident path
    {"cone_rank_memory", "PMPC_CONE_RANK_MEMORY", 1},  // cone objective: the weight assignment the previous solve of the shape settled on is tried first
    {"cone_epigraph", "PMPC_CONE_EPIGRAPH", 1},    // cone objective with hard boxes: epigraph problem in the shared-control space (any tie pattern); 0: weighted-QP fixed point
    {"cond_grouped", "PMPC_COND_GROUPED", 1},      // Nc > 1: condensed Hessians summed over groups of particles inside the condensing kernel
    {"as_freeze_tol", "PMPC_AS_FREEZE_TOL", 1e-9}, // stage-cone rounds: a shared-control step below this (relative) is zero for every particle; settled ones skip the forward sweep
    {"as_ckpt", "PMPC_AS_CKPT", 1},                // factor sweeps checkpoint their cost-to-go at stages 4, 8, 16, 32, ..; the later rounds' sweeps restart at the lowest checkpoint above the highest changed stage
    {"as_sens_min_m", "PMPC_AS_SENS_MIN_M", 3072}, // particles per rank from which the forward sweep records sensitivities to the shared step and settled particles of the later rounds are updated elementwise (one consensus stage; 0: never)
    {"as_perm_min_m", "PMPC_AS_PERM_MIN_M", 2048}, // particles per rank from which a later round's launches take the unsettled particles first (their long sweeps spread one per SIMD); 0: never
    {"cone_path", "PMPC_CONE_PATH", 0},            // cone objective with hard boxes, which body answers: 0 automatic (what the context learnt about the shape decides the order), 1 free-particles body first, 2 epigraph path (free-particles body never), 3 rank-based weighted-QP iteration only
};

namespace pmpc_impl {


void allreduce(pmpc_ctx *c, void *buf, size_t n, ncclDataType_t dt, ncclRedOp_t op) {
  if (!c->multi()) return;
  ncclResult_t r = g_rccl.AllReduce(buf, buf, n, dt, op, c->comm, c->stream);
  if (r != ncclSuccess) {
    fprintf(stderr, "pmpc_hip: ncclAllReduce failed (%d)\n", (int)r);
    throw PmpcHipError{(int)r, "ncclAllReduce", __FILE__, __LINE__};
  }
}
void broadcast(pmpc_ctx *c, void *buf, size_t n, ncclDataType_t dt, int root) {
  ncclResult_t r = g_rccl.Broadcast(buf, buf, n, dt, root, c->comm, c->stream);
  if (r != ncclSuccess) {
    fprintf(stderr, "pmpc_hip: ncclBroadcast failed (%d)\n", (int)r);
    throw PmpcHipError{(int)r, "ncclBroadcast", __FILE__, __LINE__};
  }
}

// wait until the device has published sequence number `want` into host-coherent memory; stream sync after 2 s of polling
void wait_published(pmpc_ctx *c, volatile unsigned long long *seq, unsigned long long want) {
  if (!spin_until([&] { return *seq == want; })) {
    HIP_CHECK(hipStreamSynchronize(c->stream));
    if (*seq != want) {
      fprintf(stderr, "pmpc_hip: device scalars were never published\n");
      throw PmpcHipError{-1, "wait_published", __FILE__, __LINE__};
    }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
}
void read_scalars(pmpc_ctx *c) {  // sc->status carries the (cross-rank) failure flag of the last exchange
  wait_published(c, &c->mirror->seq, c->seq);
  memcpy(c->sc_host, (const void *)&c->mirror->sc, sizeof(IpmScal));
  *c->fail_host = c->sc_host->status;
}

// one sync point of the IPM: local partials -> exchange table -> (RCCL all-reduce(sum) == all-gather) -> scalars
void exchange(pmpc_ctx *c, int phase) {
  Workspace &w = c->ws;
  IpmScal *sc = (IpmScal *)w.sc.p;
  const int B2 = 2 * PMPC_RED_BLOCKS;
  c->seq++;
  if (!c->multi()) {
    launch_ipm_exchange(phase, true, true, sc, (const int *)w.fail.p, w.xch.d(), 0, 1, w.part_sum.d(), w.part_cnt.d(),
                        w.part_max.d(), B2, c->stream, 0.0, nullptr, &c->mirror_dev->sc, &c->mirror_dev->seq, c->seq);
    return;
  }
  launch_ipm_exchange(phase, true, false, sc, (const int *)w.fail.p, w.xch.d(), c->rank, c->world, w.part_sum.d(),
                      w.part_cnt.d(), w.part_max.d(), B2, c->stream);
  allreduce(c, w.xch.p, (size_t)c->world * 8, ncclFloat64, ncclSum);
  launch_ipm_exchange(phase, false, true, sc, (const int *)w.fail.p, w.xch.d(), c->rank, c->world, w.part_sum.d(),
                      w.part_cnt.d(), w.part_max.d(), B2, c->stream, 0.0, nullptr, &c->mirror_dev->sc, &c->mirror_dev->seq, c->seq);
}

// one structured Newton solve: backward (factor or vector-only) -> reduce -> all-reduce -> dense solve -> forward
void structured_solve(pmpc_ctx *c, LQArgs &a, bool factor, bool fast, bool prep_done) {
  hipStream_t s = c->stream;
  Workspace &w = c->ws;
  const int nc = a.Nc * a.u;
  if (fast && factor && !prep_done) launch_grad_prep(a, s);
  {
    ProfScope ps(c, factor ? ((fast && a.as_settled_in) ? 4 : 0) : 1);
    if (fast && a.as_act) launch_bwd_as(a, s);  // a round of the active-set iteration (kernels_as.hip)
    else if (fast) launch_bwd_fast(a, factor, s);
    else launch_bwd_generic(a, factor, s);
  }
  if (nc > 0) {
    ProfScope ps(c, 3);
    double *Hc = w.Hg.d(), *gc = w.Hg.d() + (size_t)nc * nc;
    // condensed Hessians summed over groups of particles inside the condensing kernel: when nothing downstream looks at ONE particle's
    // H_i again (settled particles of the rounds refresh g_i += H_i delta; consensus weights scale H_i; the cone path's host reads them)
    const bool grouped = fast && factor && a.Nc > 1 && nc * nc + nc > 32 && !a.as_act && !a.as_settled_in && !a.cons_w && c->opt[OPT_COND_GROUPED] != 0.0;
    a.Hc_grp = nullptr;
    if (grouped) {
      w.Hc_grp.ensure((size_t)cond_fast_groups(a.M) * nc * nc * sizeof(double));
      a.Hc_grp = w.Hc_grp.d();
    }
    if (fast && factor) launch_cond_fast(a, s);
    // sharded active-set rounds: the previous round's change counters travel behind [Hc | gc] (one collective per round
    // instead of two); the decision about that round is taken right behind the all-reduce, before this round's forward sweep
    const size_t tail = (a.as_merge && factor) ? 5 : 0;  // {released, activated, bad, failure, open cones}
    double *tl = Hc + (size_t)nc * nc + nc;
    auto merged_exchange = [&]() {
      if (a.as_merge == 2) launch_as_ctl(const_cast<AsCtl *>(a.as_ctl), a.as_cnt, a.M, (const int *)w.fail.p, 1, 0, 0, nullptr, nullptr, 0, s, tl, nullptr, a.as_open);
      else if (a.as_merge == 1) HIP_CHECK(hipMemsetAsync(tl, 0, 5 * sizeof(double), s));
      if (factor) allreduce(c, Hc, (size_t)nc * nc + nc + tail, ncclFloat64, ncclSum);
      else allreduce(c, gc, nc, ncclFloat64, ncclSum);
      if (a.as_merge == 2)
        launch_as_ctl(const_cast<AsCtl *>(a.as_ctl), nullptr, a.M, (const int *)w.fail.p, 0, 1, 0, &c->mirror_dev->ctl, &c->mirror_dev->as_seq, c->as_seq, s, tl);
    };
    const bool wave_solve = c->opt[OPT_AS_WAVE_CONS] != 0.0;
    a.cons_G = 0;
    // consensus weights (cone objective): the reductions read lambda_i (H_i, g_i) from scaled copies; the per-particle arrays stay
    // unweighted (the settled particles' g_i += H_i delta and the host's epigraph solve want them so)
    const double *HcP = a.Hc_part, *gcP = a.gc_part;
    if (a.cons_w) {
      w.Hc_w.ensure((size_t)a.M * nc * nc * sizeof(double)); w.gc_w.ensure((size_t)a.M * nc * sizeof(double));
      launch_cons_scale(a.Hc_part, a.gc_part, a.cons_w, a.M, nc, factor, w.Hc_w.d(), w.gc_w.d(), a.as_act, a.as_big, a.as_act ? nullptr : a.Du,
                        a.as_act ? nullptr : a.wu, a.u, a.owner, s);
      HcP = factor ? w.Hc_w.d() : a.Hc_part; gcP = w.gc_w.d();
    }
    if (fast && a.as_act && factor && !c->multi() && a.Nc == 1 && wave_solve) {
      // active-set round on one rank with one consensus stage: block partials only — every wave of the forward sweep sums
      // them (same order everywhere) and solves the u x u system itself: the second launch of the reduction is gone
      a.cons_G = launch_cons_partials(HcP, gcP, a.M, nc, w.red_tmp.d(), c->as_pend, s);
      c->as_pend.ctl = nullptr;
      a.cons_tH = w.red_tmp.d();
      a.cons_tg = w.red_tmp.d() + (size_t)64 * nc * nc;
    } else if (nc * nc + nc <= 32) {
      const bool solve_now = !c->multi();
      launch_cons_small(HcP, gcP, a.M, nc, factor, Hc, w.red_tmp.d(), solve_now, w.Lc.d(), w.duc.d(), (int *)w.fail.p, s);
      if (!solve_now) {
        merged_exchange();
        if (fast && a.as_act && factor && a.Nc == 1 && wave_solve) {
          // sharded active-set round: the all-reduced [Hc | gc] is ONE partial for the forward sweep's waves to solve
          a.cons_G = 1; a.cons_tH = Hc; a.cons_tg = gc;
        } else {
          launch_cons_solve(Hc, w.Lc.d(), gc, w.duc.d(), nc, factor, (int *)w.fail.p, s);
        }
      }
    } else {
      if (factor && grouped) launch_reduce_particles_hg(a.Hc_grp, gcP, w.red_tmp.d(), Hc, a.M, nc, s, cond_fast_groups(a.M));
      else if (factor) launch_reduce_particles_hg(HcP, gcP, w.red_tmp.d(), Hc, a.M, nc, s);  // (gc sits right behind Hc)
      else launch_reduce_particles(gcP, w.red_tmp.d(), gc, a.M, nc, s);
      merged_exchange();
      launch_cons_solve(Hc, w.Lc.d(), gc, w.duc.d(), nc, factor, (int *)w.fail.p, s);
    }
  }
  ProfScope ps(c, 2);
  if (fast && a.as_act) launch_fwd_as(a, s);
  else if (fast) launch_fwd_fast(a, s);
  else launch_fwd_generic(a, s);
}

// epilogue of a solve that threw (failed HIP / RCCL call, out of memory): forget every warm-start memory, NaN outputs if the
// device still takes work, status 2.  Never throws.
int fail_after_error(pmpc_ctx *c, const pmpc_problem *p, pmpc_info *info) {
  Workspace &w = c->ws;
  w.as_key = w.warm_key = w.soc_key = w.cons_key = w.xb_block_key = w.su_key = -1;
  w.as_U_valid = false;
  w.xb_warm_backoff = w.xb_warm_fails = 0;
  c->cone_rw_key = -1;
  c->xb_ctrl_from = -1;
  c->as_pend.ctl = nullptr;
  c->staged.clear();
  (void)hipGetLastError();
  try {
    if (p && p->X_out && p->U_out) {
      const double nan = std::numeric_limits<double>::quiet_NaN();
      launch_fill(p->X_out, nan, (long long)p->M * p->N * p->xdim, c->stream);
      launch_fill(p->U_out, nan, (long long)p->M * p->N * p->udim, c->stream);
    }
    HIP_WARN(hipStreamSynchronize(c->stream));
  } catch (...) {
  }
  if (info) {
    memset(info, 0, sizeof(*info));
    info->status = 2;
  }
  return 2;
}

void fill_nan_outputs(pmpc_ctx *c, const pmpc_problem *p) {
  const double nan = std::numeric_limits<double>::quiet_NaN();
  launch_fill(p->X_out, nan, (long long)p->M * p->N * p->xdim, c->stream);
  launch_fill(p->U_out, nan, (long long)p->M * p->N * p->udim, c->stream);
  HIP_CHECK(hipStreamSynchronize(c->stream));
}

}  // namespace pmpc_impl

// =================================================================================================
extern "C" {

const char *pmpc_version(void) { return "pmpc_hip 0.4 (gfx950)"; }
// layout check of the two structs the bindings mirror (a stale library under a newer binding, or the reverse, must fail loudly)
void pmpc_abi_struct_sizes(size_t *problem, size_t *info) { *problem = sizeof(pmpc_problem); *info = sizeof(pmpc_info); }

int pmpc_create(pmpc_ctx **out, int device) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device) {
    fprintf(stderr, "pmpc_hip: no HIP device %d available (found %d) — this library has no CPU path\n", device, ndev);
    return 1;
  }
  pmpc_ctx *c = new pmpc_ctx();
  c->device = device;
  for (int k = 0; k < OPT_COUNT; k++) {
    const char *e = getenv(kPmpcOptions[k].env);
    c->opt[k] = (e && *e) ? atof(e) : kPmpcOptions[k].dflt;
  }
  try {
    HIP_CHECK(hipSetDevice(device));
    HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->sc_host = (IpmScal *)calloc(1, sizeof(IpmScal));
    c->fail_host = (int *)calloc(1, sizeof(int));
    HIP_CHECK(hipHostMalloc((void **)&c->mirror, sizeof(pmpc_ctx::ScMirror), hipHostMallocMapped | hipHostMallocCoherent));
    memset(c->mirror, 0, sizeof(pmpc_ctx::ScMirror));
    HIP_CHECK(hipHostGetDevicePointer((void **)&c->mirror_dev, c->mirror, 0));
  } catch (const PmpcHipError &) {
    free(c->sc_host);
    free(c->fail_host);
    delete c;
    return 1;
  }
  *out = c;
  return 0;
}

int pmpc_set_option(pmpc_ctx *c, const char *key, double value) {
  if (!c || !key) return -1;
  for (int k = 0; k < OPT_COUNT; k++)
    if (!strcmp(key, kPmpcOptions[k].key)) {
      c->opt[k] = value;
      c->fp_key = -1;
      c->ws.as_key = c->ws.warm_key = c->cone_rw_key = c->cone_lam_key = c->ws.es_key = -1;  // (a remembered set / iterate / weight assignment was found under the old switches)
      return 0;
    }
  return -1;
}
int pmpc_get_option(pmpc_ctx *c, const char *key, double *value) {
  if (!c || !key || !value) return -1;
  for (int k = 0; k < OPT_COUNT; k++)
    if (!strcmp(key, kPmpcOptions[k].key)) {
      *value = c->opt[k];
      return 0;
    }
  return -1;
}

void pmpc_destroy(pmpc_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->comm && c->mock_comm) delete (MockRank *)c->comm;
  else if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
  Workspace &w = c->ws;
  DevBuf *all[] = {&w.X, &w.U, &w.dX, &w.dU, &w.dX2, &w.dU2, &w.xm, &w.xd, &w.um, &w.ud, &w.K, &w.Hinv, &w.kff, &w.gc_part, &w.Hc_part, &w.Hc_w, &w.gc_w, &w.cons_w, &w.epi_lam, &w.epi_out, &w.epi_gath, &w.es_Dx, &w.es_wx, &w.es_Du, &w.es_wu, &w.es_xm, &w.es_xd, &w.es_um, &w.es_ud, &w.es_kff2, &w.es_kff3, &w.es_gc2, &w.es_dots, &w.es_coef, &w.es_out2, &w.es_Xt, &w.es_Ut, &w.es_U, &w.es_zero, &w.scratch,
                   &w.red_tmp, &w.Hg, &w.Lc, &w.duc, &w.xch, &w.zeros, &w.zslew, &w.zslew0, &w.zum1, &w.part_sum, &w.part_cnt,
                   &w.part_max, &w.sc, &w.fail, &w.pw, &w.Jc, &w.Jg, &w.part_dev, &w.warmU, &w.lateX, &w.lateU, &w.warm_llu, &w.warm_luu, &w.warm_llx,
                   &w.warm_lux, &w.Hadd, &w.wu_soc, &w.soc_zl, &w.soc_zu, &w.soc_zc, &w.soc_dzl, &w.soc_dzu, &w.soc_dzc, &w.soc_sl, &w.soc_su, &w.soc_sc, &w.soc_dsl,
                   &w.soc_dsu, &w.soc_dsc, &w.soc_cl, &w.soc_cu, &w.soc_cc, &w.soc_wU, &w.soc_wzl, &w.soc_wzu, &w.soc_wzc, &w.as_act, &w.as_cnt, &w.as_cntp, &w.as_settled, &w.cons_lo, &w.cons_hi, &w.as_ctl, &w.as_delta, &w.as_viol, &w.as_ck, &w.as_jhi, &w.ck_stat, &w.Hc_grp, &w.as_T, &w.xb_qmax, &w.as_perm,
                   &w.sa_f, &w.sa_fx, &w.sa_fu, &w.sa_Xp, &w.sa_Up, &w.sa_Q, &w.sa_R, &w.sa_Xr, &w.sa_Ur, &w.sa_lo, &w.sa_hi, &w.sa_Xo, &w.sa_Uo,
                   &w.sa_cl, &w.sa_ch, &w.cone_A, &w.cone_c, &w.cone_z, &w.cone_rec, &w.cone_uraw, &w.as_open, &w.xb_z, &w.xb_st, &w.xb_D, &w.xb_g, &w.m64[0], &w.m64[1], &w.m64[2], &w.m64[3]};
  for (DevBuf *b : all) b->release();
  for (SlabBufs *sb : {&w.sx, &w.su})
    for (DevBuf *b : {&sb->lo, &sb->hi, &sb->tl, &sb->tu, &sb->ll, &sb->lu, &sb->cl, &sb->cu, &sb->D, &sb->w}) b->release();
  for (DevBuf &b : c->stage) b.release();
  (void)hipHostFree(c->mirror);
  if (c->pinned) (void)hipHostFree(c->pinned);
  if (c->sm_pinned) (void)hipHostFree(c->sm_pinned);
  c->host_flags.release();
  free(c->sc_host);
  free(c->fail_host);
  (void)hipStreamDestroy(c->stream);
  delete c;
}

void *pmpc_stream(pmpc_ctx *c) { return (void *)c->stream; }
void pmpc_sync(pmpc_ctx *c) { HIP_WARN(hipStreamSynchronize(c->stream)); }

void pmpc_profile_enable(pmpc_ctx *c, int level) { c->prof = level < 0 ? 0 : level; }

// Sums of HIP-event durations (ms) and launch counts per kernel class since the last read:
// 0 backward+factor, 1 backward vector-only, 2 forward sweep, 3 consensus reduce + dense solve.
void pmpc_profile_read(pmpc_ctx *c, double *ms4, long long *n4) {
  HIP_WARN(hipStreamSynchronize(c->stream));
  for (int k = 0; k < 8; k++) {
    ProfCat &pc = c->cat[k];
    for (auto &ev : pc.pending) {
      float t = 0.f;
      HIP_WARN(hipEventElapsedTime(&t, ev.first, ev.second));
      pc.ms += t;
      pc.n++;
      pc.pool.push_back(ev);
    }
    pc.pending.clear();
    if (k < 4) { ms4[k] = pc.ms; n4[k] = pc.n; }
    else if (k == 4) { c->partial_ms = pc.ms; c->partial_n = pc.n; }
    c->last_ms[k] = pc.ms;
    c->last_n[k] = pc.n;
    pc.ms = 0.0;
    pc.n = 0;
  }
}

// every class as of the last pmpc_profile_read (see pmpc_ctx::cat): ms[count], n[count], count <= 8
void pmpc_profile_read_all(pmpc_ctx *c, double *ms, long long *n, int count) {
  for (int k = 0; k < count && k < 8; k++) { ms[k] = c->last_ms[k]; n[k] = c->last_n[k]; }
}

// class 4 (factor sweeps of active-set rounds that skipped the settled particles) as of the last pmpc_profile_read
void pmpc_profile_read_partial(pmpc_ctx *c, double *ms, long long *n) {
  *ms = c->partial_ms;
  *n = c->partial_n;
}

void pmpc_restart_stats(pmpc_ctx *c, unsigned long long *out4, int reset) {
  for (int k = 0; k < 4; k++) out4[k] = 0;
  if (!c || !c->ws.ck_stat.p) return;
  (void)hipSetDevice(c->device);
  HIP_WARN(hipStreamSynchronize(c->stream));
  HIP_WARN(hipMemcpy(out4, c->ws.ck_stat.p, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  if (reset) HIP_WARN(hipMemset(c->ws.ck_stat.p, 0, 4 * sizeof(unsigned long long)));
}

int pmpc_comm_unique_id(void *out128) {
  if (!g_rccl.load()) return 1;
  ncclUniqueId id;
  if (g_rccl.GetUniqueId(&id) != ncclSuccess) return 2;
  memcpy(out128, &id, sizeof(id));
  return 0;
}

int pmpc_comm_init(pmpc_ctx *c, int rank, int world, const void *id128) {
  // TEST HOOK: PMPC_RCCL_SINGLE=1 makes a world of one a real 1-rank RCCL communicator and sends every solve through the
  // multi-rank code paths (packed exchange, consensus all-reduce, bounds broadcast) — the only way to run the actual
  // ncclAllReduce / ncclBroadcast calls on a one-GPU box (RCCL refuses two ranks on one device).
  const char *single = getenv("PMPC_RCCL_SINGLE");
  if (world <= 1 && !(single && single[0] == '1')) {
    c->rank = 0;
    c->world = 1;
    return 0;
  }
  if (world <= 1) { world = 1; rank = 0; c->single_rank_comm = true; }
  if (!g_rccl.load()) return 1;
  if (hipSetDevice(c->device) != hipSuccess) return 3;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  if (g_rccl.CommInitRank(&c->comm, world, id, rank) != ncclSuccess) return 2;
  c->rank = rank;
  c->world = world;
  return 0;
}
// TEST HOOK: join the in-process communicator `group` (see MockGroup above) as rank `rank` of `world`.  Every context of
// the group lives on one device and must be driven by its own host thread.  Installs the stand-in collectives
// process-wide: do not mix with a real RCCL communicator in the same process.
int pmpc_comm_init_mock(pmpc_ctx *c, int rank, int world, int group) {
  std::lock_guard<std::mutex> lk(g_mock_mutex);
  MockGroup *&g = g_mock_groups[group];
  if (!g) {
    g = new MockGroup();
    g->world = world;
    g->slot.resize(world);
  }
  if (g->world != world || rank < 0 || rank >= world) return 1;
  g_rccl.AllReduce = mock_allreduce;
  g_rccl.Broadcast = mock_broadcast;
  c->comm = (ncclComm_t) new MockRank{g, rank};
  c->mock_comm = true;
  c->rank = rank;
  c->world = world;
  return 0;
}
int pmpc_comm_rank(pmpc_ctx *c) { return c->rank; }
int pmpc_comm_world(pmpc_ctx *c) { return c->world; }

int pmpc_scp_residual_device(pmpc_ctx *c, size_t xdim, size_t udim, size_t N, size_t M, const double *X, const double *X_prev,
                             const double *U, const double *U_prev, double *out) {
  try {
    HIP_CHECK(hipSetDevice(c->device));
    ProfScope ps(c, 7);
    launch_scp_residual(X, X_prev, U, U_prev, (long long)M * (long long)N, (int)xdim, (int)udim, out, c->stream);
    HIP_CHECK(hipGetLastError());
  } catch (const PmpcHipError &) {
    return 2;
  }
  return 0;
}

int pmpc_linearize_device_f32(pmpc_ctx *c, int model, size_t N, size_t M, const double *x0, const double *X_prev,
                              const double *U_prev, const double *params, double *f, float *fx, float *fu) {
  try {
    HIP_CHECK(hipSetDevice(c->device));
    ProfScope ps(c, 6);
    launch_linearize(model, (int)N, (int)M, x0, X_prev, U_prev, params, f, (double *)fx, (double *)fu, c->stream, 1);
    HIP_CHECK(hipGetLastError());
  } catch (const PmpcHipError &) {
    return 2;
  }
  return 0;
}

int pmpc_linearize_device(pmpc_ctx *c, int model, size_t N, size_t M, const double *x0, const double *X_prev,
                          const double *U_prev, const double *params, double *f, double *fx, double *fu) {
  try {
    HIP_CHECK(hipSetDevice(c->device));
    ProfScope ps(c, 6);
    launch_linearize(model, (int)N, (int)M, x0, X_prev, U_prev, params, f, fx, fu, c->stream);
    HIP_CHECK(hipGetLastError());
  } catch (const PmpcHipError &) {
    return 2;
  }
  return 0;
}

// -------------------------------------------------------------------------------------------------
static int solve_impl_body(pmpc_ctx *c, const pmpc_problem *p, pmpc_info *info, int verbose, bool soc);
constexpr int PMPC_NEEDS_F64 = -7;  // solve_impl_body on an fp32-storage problem: this solve needs a path that runs the fp64 kernels
// fp32-storage problem -> the same problem with fx, fu, Q, R widened (exactly) into workspace copies, flag cleared
pmpc_problem widened_f32_problem(pmpc_ctx *c, const pmpc_problem *p, bool jacobians) {
  Workspace &w = c->ws;
  const long long rows = (long long)p->M * (long long)p->N, x = (long long)p->xdim, u = (long long)p->udim;
  const long long cnt[4] = {rows * x * x, rows * x * u, rows * x * x, rows * u * u};
  const double *src[4] = {p->fx, p->fu, p->Q, p->R};
  for (int k = jacobians ? 0 : 2; k < 4; k++) {
    w.m64[k].ensure((size_t)cnt[k] * sizeof(double));
    launch_widen_f32((const float *)src[k], w.m64[k].d(), cnt[k], c->stream);
  }
  pmpc_problem q = *p;
  q.flags &= ~(unsigned)PMPC_F32_MATRICES;
  if (jacobians) { q.fx = w.m64[0].d(); q.fu = w.m64[1].d(); }
  q.Q = w.m64[2].d(); q.R = w.m64[3].d();
  return q;
}
static int solve_impl(pmpc_ctx *c, const pmpc_problem *p, pmpc_info *info, int verbose, bool soc) {
  try {
    int st = solve_impl_body(c, p, info, verbose, soc);
    if (st == PMPC_NEEDS_F64) {
      // fp32-storage mode outside the warm-started active-set rounds (first solve of a loop, fallbacks, other dims / consensus
      // horizons): widen fx, fu, Q, R into workspace copies — exact — and run the ordinary solve on them
      const pmpc_problem q = widened_f32_problem(c, p);
      if (verbose) printf("pmpc_hip: fp32-storage problem: widened for the fp64 kernels\n");
      st = solve_impl_body(c, &q, info, verbose, soc);
    }
    HIP_CHECK(hipGetLastError());  // a kernel launch that was refused (bad configuration, lost device) is a failed solve
    return st;
  } catch (const PmpcHipError &) {
    return fail_after_error(c, p, info);
  }
}

int pmpc_lqp_solve_device(pmpc_ctx *c, const pmpc_problem *p, pmpc_info *info, int verbose) {
  return solve_impl(c, p, info, verbose, false);
}
int pmpc_lsoc_solve_device(pmpc_ctx *c, const pmpc_problem *p, pmpc_info *info, int verbose) {
  return solve_impl(c, p, info, verbose, true);
}

// Slew penalties on the MFMA path: restate the problem in control increments (kernels_slew.hip: state [x; u], control
// u_j - u_{j-1}, control boxes -> boxes on the state), solve that plain problem, split the state back into (X, U).
static bool slew_increment_form_applies(const pmpc_ctx *c, const pmpc_problem *p, bool soc) {
  if (soc || !(p->flags & (PMPC_HAS_SLEW | PMPC_HAS_SLEW0)) || (p->flags & PMPC_FORCE_GENERIC) || !(p->flags & PMPC_SYMMETRIC_COST))
    return false;
  if (p->N < 2) return false;  // N = 1: the reference's diagonal rule is not the plain penalty (lqp_utils.jl:31-39)
  // With boxes the control boxes become STATE boxes of the restated problem.  Until r03 those met the interior-point iteration only
  // (1.8x - 2.6x slower cold and ~10x slower warm than the generic kernels' active-set rounds), so boxed slew problems stayed on the
  // generic kernels; with the state-box rounds of kernels_xbox.hip the restated form is 2.1x - 5.4x FASTER than the generic kernels cold, 1.45x - 2x warm,
  // and agrees with them to 1e-15 (tools/debug/slew_paths.py, profiles/r03_f_slew_paths.txt, CHANGELOG.md 3.3).  It needs the XBOX instantiation of the
  // factor sweep for (x + u, u); the options slew_increment_boxes = 0 / xbox_as = 0 put boxed slew problems back on the generic kernels.
  const bool with_boxes = c->opt[OPT_SLEW_INCREMENT_BOXES] != 0.0 && c->opt[OPT_XBOX_AS] != 0.0;
  if ((p->flags & (PMPC_HAS_XBOUNDS | PMPC_HAS_UBOUNDS)) && !(with_boxes && xbox_as_dims_supported((int)(p->xdim + p->udim), (int)p->udim))) return false;
  // (barrier mode: the shared controls' boxes carry ONE barrier term — particle 0's — which M state boxes on the u-part would count M times)
  if ((p->flags & (PMPC_HAS_XBOUNDS | PMPC_HAS_UBOUNDS)) && p->barrier_mu > 0.0) return false;
  LQArgs t;
  memset(&t, 0, sizeof(t));
  t.x = (int)(p->xdim + p->udim); t.u = (int)p->udim; t.N = (int)p->N; t.M = (int)p->M; t.sym_cost = 1;
  return lq_fast_supported(t);
}

// the restated problem `q` (state [x; u], control increments; outputs in the workspace) and the augmentation record `g` the split needs
void build_slew_increment_problem(pmpc_ctx *c, const pmpc_problem *p, pmpc_problem &q, SlewAug &g) {
  HIP_CHECK(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  Workspace &w = c->ws;
  const int x = (int)p->xdim, u = (int)p->udim, N = (int)p->N, M = (int)p->M, n = x + u;
  const int Nc = p->Nc < 0 ? N : (int)p->Nc;
  const bool has_xb = p->flags & PMPC_HAS_XBOUNDS, has_ub = p->flags & PMPC_HAS_UBOUNDS;
  const bool has_slew = p->flags & PMPC_HAS_SLEW, has_slew0 = p->flags & PMPC_HAS_SLEW0;
  const size_t rows = (size_t)M * N, D8 = sizeof(double);
  w.sa_f.ensure(rows * n * D8); w.sa_fx.ensure(rows * n * n * D8); w.sa_fu.ensure(rows * n * u * D8);
  w.sa_Xp.ensure(rows * n * D8); w.sa_Up.ensure(rows * u * D8); w.sa_Q.ensure(rows * n * n * D8); w.sa_R.ensure(rows * u * u * D8);
  w.sa_Xr.ensure(rows * n * D8); w.sa_Ur.ensure(rows * u * D8); w.sa_Xo.ensure(rows * n * D8); w.sa_Uo.ensure(rows * u * D8);
  const bool boxes = has_xb || has_ub;
  if (boxes) { w.sa_lo.ensure(rows * n * D8); w.sa_hi.ensure(rows * n * D8); }
  if (w.zslew.bytes < (size_t)M * D8 || w.zum1.bytes < (size_t)M * u * D8) {
    w.zslew.ensure((size_t)M * D8); w.zslew0.ensure((size_t)M * D8); w.zum1.ensure((size_t)M * u * D8);
    HIP_CHECK(hipMemsetAsync(w.zslew.p, 0, (size_t)M * D8, s));
    HIP_CHECK(hipMemsetAsync(w.zslew0.p, 0, (size_t)M * D8, s));
    HIP_CHECK(hipMemsetAsync(w.zum1.p, 0, (size_t)M * u * D8, s));
  }
  memset(&g, 0, sizeof(g));
  g.x = x; g.u = u; g.N = N; g.M = M; g.Nc = Nc; g.has_xb = has_xb; g.has_ub = has_ub;
  g.has_um1 = (has_slew0 && Nc >= 1) ? 1 : 0;  // the linear term -s0 u_0'u_{-1} exists only with consensus stages (lqp_utils.jl:165)
  const double reg = std::min(p->reg_x, p->reg_u);  // the one regulariser of the restated problem; the excess goes into the cost blocks
  g.dx = p->reg_x - reg; g.du = p->reg_u - reg;
  g.f = p->f; g.fx = p->fx; g.fu = p->fu; g.Xp = p->X_prev; g.Up = p->U_prev; g.Q = p->Q; g.R = p->R; g.Xr = p->X_ref; g.Ur = p->U_ref;
  g.lx = p->lx; g.ux = p->ux; g.lu = p->lu; g.uu = p->uu;
  g.slew = has_slew ? p->slew_reg : w.zslew.d();
  g.slew0 = has_slew0 ? p->slew_reg0 : w.zslew0.d();
  g.um1 = has_slew0 ? p->slew_um1 : w.zum1.d();
  if (has_ub && Nc > 0) {  // consensus controls: (global) particle 0's boxes, lqp_utils.jl:329-330
    const size_t nc = (size_t)Nc * u;
    w.sa_cl.ensure(nc * D8); w.sa_ch.ensure(nc * D8);
    HIP_CHECK(hipMemcpyAsync(w.sa_cl.p, p->lu, nc * D8, hipMemcpyDeviceToDevice, s));
    HIP_CHECK(hipMemcpyAsync(w.sa_ch.p, p->uu, nc * D8, hipMemcpyDeviceToDevice, s));
    if (c->multi()) {
      broadcast(c, w.sa_cl.p, nc, ncclFloat64, 0);
      broadcast(c, w.sa_ch.p, nc, ncclFloat64, 0);
    }
    g.cons_lo = w.sa_cl.d(); g.cons_hi = w.sa_ch.d();
  }
  g.af = w.sa_f.d(); g.afx = w.sa_fx.d(); g.afu = w.sa_fu.d(); g.aXp = w.sa_Xp.d(); g.aUp = w.sa_Up.d();
  g.aQ = w.sa_Q.d(); g.aR = w.sa_R.d(); g.aXr = w.sa_Xr.d(); g.aUr = w.sa_Ur.d();
  g.alo = boxes ? w.sa_lo.d() : nullptr; g.ahi = boxes ? w.sa_hi.d() : nullptr;
  launch_slew_augment(g, s);

  q = *p;
  q.xdim = (size_t)n;
  q.flags &= ~(PMPC_HAS_SLEW | PMPC_HAS_SLEW0 | PMPC_HAS_UBOUNDS | PMPC_HAS_XBOUNDS | PMPC_PREV_IS_LAST_SOLUTION | PMPC_STATIC_CONS_BOUNDS);
  if (boxes) q.flags |= PMPC_HAS_XBOUNDS;
  q.reg_x = reg;
  q.reg_u = 0.0;
  q.f = g.af; q.fx = g.afx; q.fu = g.afu; q.X_prev = g.aXp; q.U_prev = g.aUp; q.Q = g.aQ; q.R = g.aR; q.X_ref = g.aXr; q.U_ref = g.aUr;
  q.lx = g.alo; q.ux = g.ahi; q.lu = q.uu = nullptr;
  q.slew_reg = q.slew_reg0 = q.slew_um1 = nullptr;
  q.X_out = w.sa_Xo.d(); q.U_out = w.sa_Uo.d();
}

static int solve_slew_increment_form(pmpc_ctx *c, const pmpc_problem *p, pmpc_info *info, int verbose) {
  hipStream_t s = c->stream;
  Workspace &w = c->ws;
  const int x = (int)p->xdim, u = (int)p->udim, N = (int)p->N, M = (int)p->M;
  const int Nc = p->Nc < 0 ? N : (int)p->Nc;
  const size_t rows = (size_t)M * N;
  pmpc_problem q;
  SlewAug g;
  build_slew_increment_problem(c, p, q, g);
  pmpc_info inf;
  memset(&inf, 0, sizeof(inf));
  c->xb_ctrl_from = x;
  int st;
  try {
    st = solve_impl_body(c, &q, &inf, verbose, false);
  } catch (...) {
    c->xb_ctrl_from = -1;
    throw;
  }
  c->xb_ctrl_from = -1;
  if (st == 0) launch_slew_split(w.sa_Xo.d(), w.sa_Uo.d(), p->X_out, p->U_out, (long long)rows, x, u, N, (M > 1 || c->multi()) ? Nc : 0, g.cons_lo, g.cons_hi, s);
  else fill_nan_outputs(c, p);
  if (info) *info = inf;
  return st;
}

static int solve_impl_body(pmpc_ctx *c, const pmpc_problem *p, pmpc_info *info, int verbose, bool soc) {
  const bool f32 = (p->flags & PMPC_F32_MATRICES) != 0;
  if (f32 && (p->flags & (PMPC_HAS_SLEW | PMPC_HAS_SLEW0 | PMPC_HAS_XBOUNDS | PMPC_FORCE_GENERIC))) return PMPC_NEEDS_F64;
  if (p->xdim > 0 && p->udim > 0 && p->N > 0 && p->M > 0 && p->Nc <= (long long)p->N && slew_increment_form_applies(c, p, soc))
    return solve_slew_increment_form(c, p, info, verbose);
  HIP_CHECK(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  Workspace &w = c->ws;
  const int x = (int)p->xdim, u = (int)p->udim, N = (int)p->N, M = (int)p->M;
  const int Nc = p->Nc < 0 ? N : (int)std::min<long long>(p->Nc, (long long)N);  // main.jl:127-128 (Nc > N is refused below)
  const bool has_xb = p->flags & PMPC_HAS_XBOUNDS, has_ub = p->flags & PMPC_HAS_UBOUNDS;
  const bool has_slew = p->flags & PMPC_HAS_SLEW, has_slew0 = p->flags & PMPC_HAS_SLEW0;
  const int nc = Nc * u;
  pmpc_info inf;
  memset(&inf, 0, sizeof(inf));
  if (x <= 0 || u <= 0 || N <= 0 || M <= 0 || p->Nc > (long long)N) {
    // Nc > N: the reference indexes U[:, 1:Nc] out of bounds (lqp_utils.jl:17-61 -> BoundsError); here: a failed solve
    if (p->Nc > (long long)N) fprintf(stderr, "pmpc_hip: consensus horizon Nc = %lld exceeds N = %d\n", p->Nc, N);
    inf.status = 2;
    if (x > 0 && u > 0 && N > 0 && M > 0 && p->X_out && p->U_out) fill_nan_outputs(c, p);
    if (info) *info = inf;
    return inf.status;
  }
  const size_t nx = (size_t)M * N * x, nu = (size_t)M * N * u, D8 = sizeof(double);

  LQArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.u = u; a.N = N; a.M = M; a.Nc = Nc;
  a.w = has_slew ? u : 0;  // a zero slew vector still takes the augmented path: correct, only slower
  a.n = x + a.w;
  a.reg_x = p->reg_x; a.reg_u = p->reg_u;
  a.f = p->f; a.fx = p->fx; a.fu = p->fu; a.Q = p->Q; a.R = p->R;
  a.X_prev = p->X_prev; a.U_prev = p->U_prev; a.X_ref = p->X_ref; a.U_ref = p->U_ref;
  a.owner = (c->rank == 0);
  a.any_slew = (has_slew || has_slew0) ? 1 : 0;
  a.sym_cost = (p->flags & PMPC_SYMMETRIC_COST) ? 1 : 0;
  a.pw = p->weights;
  a.cons_w = c->cons_w_active;  // (set by lcone_body around its sub-problem solves; null otherwise)

  // ---- workspace ---------------------------------------------------------------------------------
  w.X.ensure(nx * D8); w.U.ensure(nu * D8); w.dX.ensure(nx * D8); w.dU.ensure(nu * D8);
  w.dX2.ensure(nx * D8); w.dU2.ensure(nu * D8);
  {  // generic path: K (u x n) per stage; fast path: one 64-double factor record per stage
    size_t kb = nu * a.n * D8, rb = (size_t)M * N * 64 * D8;
    w.K.ensure(kb > rb ? kb : rb);
  }
  w.Hinv.ensure(nu * u * D8);
  w.kff.ensure(nu * D8);
  w.gc_part.ensure((size_t)M * nc * D8); w.Hc_part.ensure((size_t)M * nc * nc * D8);
  w.scratch.ensure((size_t)M * 3 * a.n * nc * D8);
  w.red_tmp.ensure((size_t)64 * ((size_t)nc * nc + nc) * D8);
  w.Hg.ensure(((size_t)nc * nc + nc + 5) * D8);  // (+ 4: change counters of the active-set rounds, sharded runs)
  w.Lc.ensure(((size_t)nc * nc + (size_t)((nc + 15) / 16) * 272) * D8);  // (+ the inverse diagonal blocks of k_cons_solve_lds)
  w.duc.ensure((size_t)nc * D8);
  w.sc.ensure(sizeof(IpmScal)); w.fail.ensure(sizeof(int)); w.xch.ensure((size_t)c->world * 8 * D8);
  const bool fresh_parts = w.part_sum.bytes == 0;
  w.part_sum.ensure(2 * PMPC_RED_BLOCKS * D8); w.part_cnt.ensure(2 * PMPC_RED_BLOCKS * D8);
  w.part_max.ensure(2 * PMPC_RED_BLOCKS * D8);
  if (fresh_parts) {
    HIP_CHECK(hipMemsetAsync(w.part_sum.p, 0, 2 * PMPC_RED_BLOCKS * D8, s));
    HIP_CHECK(hipMemsetAsync(w.part_cnt.p, 0, 2 * PMPC_RED_BLOCKS * D8, s));
    HIP_CHECK(hipMemsetAsync(w.part_max.p, 0, 2 * PMPC_RED_BLOCKS * D8, s));
  }
  const long long as_prev = w.as_key;  // accepted active set + solution of the previous solve (valid only if nothing ran since)
  w.as_key = -1;
  if (!has_slew || !has_slew0) {
    if (w.zslew.bytes < (size_t)M * D8 || w.zum1.bytes < (size_t)M * u * D8) {
      w.zslew.ensure((size_t)M * D8); w.zslew0.ensure((size_t)M * D8); w.zum1.ensure((size_t)M * u * D8);
      HIP_CHECK(hipMemsetAsync(w.zslew.p, 0, (size_t)M * D8, s));
      HIP_CHECK(hipMemsetAsync(w.zslew0.p, 0, (size_t)M * D8, s));
      HIP_CHECK(hipMemsetAsync(w.zum1.p, 0, (size_t)M * u * D8, s));
    }
  }
  a.slew = has_slew ? p->slew_reg : w.zslew.d();
  a.slew0 = has_slew0 ? p->slew_reg0 : w.zslew0.d();
  a.um1 = has_slew0 ? p->slew_um1 : w.zum1.d();
  a.K = w.K.d(); a.Hinv = w.Hinv.d(); a.kff = w.kff.d();
  a.gc_part = w.gc_part.d(); a.Hc_part = w.Hc_part.d(); a.scratch = w.scratch.d(); a.duc = w.duc.d();
  a.dX = w.dX.d(); a.dU = w.dU.d(); a.fail = (int *)w.fail.p;
  a.X = w.X.d(); a.U = w.U.d();
  const bool fast = !(p->flags & PMPC_FORCE_GENERIC) && lq_fast_supported(a);
  if (!fast && !(p->flags & PMPC_FORCE_GENERIC) && c->opt[OPT_WARN_SLOW_PATH] != 0.0 && !c->warned_slow_path) {
    // a caller who forgets symmetric_cost = True (or picks dimensions nothing is compiled for) would get a several times slower solver
    // silently: say so once per context
    c->warned_slow_path = true;
    const char *why = !a.sym_cost ? "Q, R are not declared symmetric (flag PMPC_SYMMETRIC_COST / DeviceSolver(symmetric_cost=True))"
                      : a.any_slew ? "slew penalties whose increment form (xdim + udim, udim) is not a compiled pair (or N = 1)"
                      : ((size_t)M * N * (size_t)std::max(x, u) * D8 >= (1ull << 31)) ? "the problem exceeds the 2 GiB per-array addressing of the register-resident kernels"
                                                                                       : "(xdim, udim) is not a compiled pair (fast_common.h, PMPC_FAST_DIMS)";
    fprintf(stderr, "pmpc_hip: note: this problem (xdim %d, udim %d) runs on the generic kernels, several times slower than the register-resident MFMA path: %s. "
                    "Said once per context; pmpc_set_option(ctx, \"warn_slow_path\", 0) or PMPC_WARN_SLOW_PATH=0 silences it.\n", x, u, why);
  }
  if (f32) {
    // fp32-storage mode: only the warm-started active-set rounds of an SCP loop (no rollout, no equality phase) read the float
    // arrays; everything else asks the caller (solve_impl) for widened copies
    const bool f32_defect_on = c->opt[OPT_AS_DEFECT] != 0.0;
    if (!(fast && f32_as_dims_supported(x, u) && Nc <= 1 && (p->flags & PMPC_PREV_IS_LAST_SOLUTION) && !(p->flags & PMPC_COLD_START) &&
          f32_defect_on && !(p->barrier_mu > 0.0))) {
      w.as_key = as_prev;  // (nothing ran: the warm-start memory stands for the widened solve)
      return PMPC_NEEDS_F64;
    }
    a.mat32 = 1;
  }
  if (w.zeros.bytes == 0) {
    w.zeros.ensure(64 * D8);
    HIP_CHECK(hipMemsetAsync(w.zeros.p, 0, 64 * D8, s));
  }
  a.zeros = w.zeros.d();
  if (fast) {
    w.xm.ensure(nx * D8); w.xd.ensure(nx * D8); w.um.ensure(nu * D8); w.ud.ensure(nu * D8);
    a.xm = w.xm.d(); a.xd = w.xd.d(); a.um = w.um.d(); a.ud = w.ud.d();
  }
  inf.fast_path = fast ? (f32 ? 2 : 1) : 0;  // (2: the active-set sweeps on fp32-stored matrices — set back to 1 by the widened re-solve)
  IpmScal *sc = (IpmScal *)w.sc.p;

  // ---- 1. equality-only optimum: one Newton step from a dynamics-consistent base point -----------
  const double mu_target = (p->barrier_mu > 0.0 && (has_xb || has_ub)) ? p->barrier_mu : 0.0;
  if (mu_target > 0.0 && w.part_dev.bytes == 0) {
    w.part_dev.ensure(2 * PMPC_RED_BLOCKS * D8);
    HIP_CHECK(hipMemsetAsync(w.part_dev.p, 0, 2 * PMPC_RED_BLOCKS * D8, s));
  }
  // failure flag and interior-point scalars: reset lazily — the warm-started active-set rounds (the path an SCP loop takes)
  // clear the flag in their own first kernel and never touch the scalars
  bool scalars_reset = false;
  auto reset_scalars = [&]() {
    if (scalars_reset) return;
    scalars_reset = true;
    HIP_CHECK(hipMemsetAsync(w.fail.p, 0, sizeof(int), s));
    launch_ipm_exchange(0, false, false, sc, (const int *)w.fail.p, w.xch.d(), c->rank, c->world, nullptr, nullptr, nullptr, 0, s,
                        mu_target, w.part_dev.d());
  };
  auto equality_solve = [&]() {
    launch_init_base(w.U.d(), p->U_prev, M, N, u, Nc, s);
    if (fast) launch_rollout_fast(a, w.U.d(), w.X.d(), s);
    else launch_rollout(a, w.U.d(), w.X.d(), s);
    a.Dx = a.Du = a.wx = a.wu = nullptr;
    structured_solve(c, a, true, fast);
    inf.structured_solves++;
    launch_axpy(w.X.d(), w.dX.d(), 1.0, (long long)nx, s);
    launch_axpy(w.U.d(), w.dU.d(), 1.0, (long long)nu, s);
  };

  bool outputs_written = false;
  auto finish = [&](int status) {
    inf.status = status;
    if (status == 0) {
      if (!outputs_written) {  // (an accepted active-set point is written to the outputs by its own kernel)
        HIP_CHECK(hipMemcpyAsync(p->X_out, w.X.p, nx * D8, hipMemcpyDeviceToDevice, s));
        HIP_CHECK(hipMemcpyAsync(p->U_out, w.U.p, nu * D8, hipMemcpyDeviceToDevice, s));
      }
    } else {
      fill_nan_outputs(c, p);
    }
    if (info) *info = inf;
    return status;
  };

  // ---- slabs of bounded variables ----------------------------------------------------------------
  Slab sx, su;
  memset(&sx, 0, sizeof(sx));
  memset(&su, 0, sizeof(su));
  auto setup_slab = [&](Slab &sl, SlabBufs &b, size_t cnt, int d, bool is_u, const double *lo, const double *hi, double *z,
                        double *dz) {
    for (DevBuf *q : {&b.tl, &b.tu, &b.ll, &b.lu, &b.cl, &b.cu, &b.D, &b.w}) q->ensure(cnt * D8);
    sl.count = (long long)cnt; sl.d = d; sl.N = N; sl.Nc = Nc; sl.is_u = is_u ? 1 : 0; sl.owner = a.owner;
    sl.lo = lo; sl.hi = hi; sl.z = z; sl.dz = dz;
    sl.tl = b.tl.d(); sl.tu = b.tu.d(); sl.ll = b.ll.d(); sl.lu = b.lu.d(); sl.cl = b.cl.d(); sl.cu = b.cu.d();
    sl.D = b.D.d(); sl.w = b.w.d();
  };
  if (has_xb) setup_slab(sx, w.sx, nx, x, false, p->lx, p->ux, w.X.d(), w.dX.d());
  if (has_ub) {
    const double *lo = p->lu, *hi = p->uu;
    const long long sukey = (((((long long)u * 131 + N) * 1000003 + M) * 131 + Nc) * 2 + (soc ? 1 : 0));  // (the cone solver's copy drops a box side)
    if (Nc > 0 && (M > 1 || c->multi()) && (p->flags & PMPC_STATIC_CONS_BOUNDS) && w.su_key == sukey && w.su_src_lo == p->lu &&
        w.su_src_hi == p->uu && w.su.lo.bytes >= nu * D8) {
      // the caller vouches EXPLICITLY (PMPC_STATIC_CONS_BOUNDS, on any rank count) that the CONTENTS of lu / uu are those of the
      // previous solve of this shape, as inside an SCP loop: the working copy made then — the caller's boxes with particle 0's on
      // the consensus stages — still stands: two 6.5 MB copies and a kernel per solve saved.  Nothing on the device compares
      // contents, so without the flag the copy is remade (a caller that moves a trust region in place just leaves the flag off).
      lo = w.su.lo.d(); hi = w.su.hi.d();
    } else if (Nc > 0 && (M > 1 || c->multi())) {  // consensus bounds = global particle 0's (lqp_utils.jl:329-330)
      w.su_key = -1;  // (valid again only once every copy below is enqueued: a throw in between must not leave a half-built copy trusted)
      w.su.lo.ensure(nu * D8); w.su.hi.ensure(nu * D8);
      HIP_CHECK(hipMemcpyAsync(w.su.lo.p, p->lu, nu * D8, hipMemcpyDeviceToDevice, s));
      HIP_CHECK(hipMemcpyAsync(w.su.hi.p, p->uu, nu * D8, hipMemcpyDeviceToDevice, s));
      if (c->multi()) {
        // rank 0's bounds of the consensus controls reach every rank: two small broadcasts — or, when the caller vouches that
        // they are the previous solve's (PMPC_STATIC_CONS_BOUNDS: an SCP loop), the copy kept from then (each tiny
        // collective costs tens of microseconds over xGMI, a tenth of a sharded solve)
        const long long bkey = ((((long long)u * 131 + N) * 1000003 + M) * 131 + Nc);
        if ((p->flags & PMPC_STATIC_CONS_BOUNDS) && w.cons_key == bkey) {
          HIP_CHECK(hipMemcpyAsync(w.su.lo.p, w.cons_lo.p, (size_t)nc * D8, hipMemcpyDeviceToDevice, s));
          HIP_CHECK(hipMemcpyAsync(w.su.hi.p, w.cons_hi.p, (size_t)nc * D8, hipMemcpyDeviceToDevice, s));
        } else {
          broadcast(c, w.su.lo.p, (size_t)nc, ncclFloat64, 0);
          broadcast(c, w.su.hi.p, (size_t)nc, ncclFloat64, 0);
          w.cons_lo.ensure((size_t)nc * D8); w.cons_hi.ensure((size_t)nc * D8);
          HIP_CHECK(hipMemcpyAsync(w.cons_lo.p, w.su.lo.p, (size_t)nc * D8, hipMemcpyDeviceToDevice, s));
          HIP_CHECK(hipMemcpyAsync(w.cons_hi.p, w.su.hi.p, (size_t)nc * D8, hipMemcpyDeviceToDevice, s));
          w.cons_key = bkey;
        }
      }
      launch_cons_bounds(w.su.lo.d(), w.su.hi.d(), M, N, u, Nc, s);
      lo = w.su.lo.d(); hi = w.su.hi.d();
      w.su_key = sukey; w.su_src_lo = p->lu; w.su_src_hi = p->uu;
    }
    setup_slab(su, w.su, nu, u, true, lo, hi, w.U.d(), w.dU.d());
  }
  const int B = PMPC_RED_BLOCKS;

  // general form of the stage cones (pmpc_problem.cone_count > 0): several cones / linear rows per stage, optionally stage-dependent data
  const int ncones = soc ? (int)p->cone_count : 0;
  int cone_rows = 0;
  bool cones_ok = true;
  if (ncones > 0) {
    cones_ok = ncones <= 4 && p->cone_sizes && p->cone_A && p->cone_c;
    for (int k = 0; cones_ok && k < ncones; k++) {
      cones_ok = p->cone_sizes[k] >= 0 && p->cone_sizes[k] <= 3;
      cone_rows += p->cone_sizes[k] + 1;
    }
    cones_ok = cones_ok && cone_rows <= 8;
  }
  if (soc && (has_xb || a.any_slew || (ncones == 0 && p->soc_u_interior == nullptr) || (ncones == 0 && p->soc_q > 0 && (!p->soc_W || !p->soc_w0 || !p->soc_v)) ||
              u > 8 || p->soc_q > 4 || !cones_ok)) {
    fprintf(stderr, "pmpc_hip: pmpc_lsoc_solve_device supports control boxes + stage cones (udim <= 8; one cone soc_q <= 4 with soc_u_interior, or "
                    "the general form: <= 4 cones of size <= 3, <= 8 rows), no state boxes / slew\n");
    return finish(2);
  }
  // Stage cones inside the active-set rounds (kernels_cone.hip: semismooth Newton on the cones' natural map, boxes by the
  // primal-dual active-set rule) — warm-started from the previous solve's set and multipliers, cold-started from soc_u_interior;
  // the path-following iteration below is the fallback.  PMPC_CONE_AS=0 switches it off.
  const bool cone_as_env = c->opt[OPT_CONE_AS] != 0.0;
  const bool cone_as = soc && cone_as_env && fast && cone_as_dims_supported(x, u) &&
                       (ncones > 0 ? cone_as_supported(u, 0) : (p->soc_q > 0 && cone_as_supported(u, (int)p->soc_q)));
  if (soc && ncones > 0 && !cone_as) {
    fprintf(stderr, "pmpc_hip: the general form of the stage cones needs the register-resident path (symmetric cost, compiled dims with udim 2..4)\n");
    return finish(2);
  }
  if (ncones == 0 && soc && p->soc_q > 0) cone_rows = (int)p->soc_q + 1;
  // State boxes inside the active-set rounds (kernels_xbox.hip); PMPC_XBOX_AS=0 switches them off (then a binding state box sends
  // the solve to the interior-point iteration, as before r03).
  const bool xbox_as_env = c->opt[OPT_XBOX_AS] != 0.0;
  const bool xbox_as = !soc && has_xb && xbox_as_env && fast && !f32 && xbox_as_dims_supported(x, u);
  if ((soc && !has_ub && cone_as) || (xbox_as && !has_ub)) {  // no control boxes: the active-set sweeps still read them — unbounded working copies
    w.su.lo.ensure(nu * D8); w.su.hi.ensure(nu * D8);
    w.su_key = -1;
    launch_fill(w.su.lo.d(), -std::numeric_limits<double>::infinity(), (long long)nu, s);
    launch_fill(w.su.hi.d(), std::numeric_limits<double>::infinity(), (long long)nu, s);
    setup_slab(su, w.su, nu, u, true, w.su.lo.d(), w.su.hi.d(), w.U.d(), w.dU.d());
  }
  if (soc && has_ub) {
    // working copy of the control boxes on every path of the cone solver: particle 0's on the consensus stages, and a lower
    // side that the cone implies (thrust >= 0 next to the thrust cone) dropped — see k_cone_drop_lo
    if (su.lo == p->lu) {
      w.su.lo.ensure(nu * D8); w.su.hi.ensure(nu * D8);
      w.su_key = -1;
      HIP_CHECK(hipMemcpyAsync(w.su.lo.p, p->lu, nu * D8, hipMemcpyDeviceToDevice, s));
      HIP_CHECK(hipMemcpyAsync(w.su.hi.p, p->uu, nu * D8, hipMemcpyDeviceToDevice, s));
      su.lo = w.su.lo.d(); su.hi = w.su.hi.d();
    }
  }
  if (soc && ncones == 0 && p->soc_q > 0) {  // cone data as one block A = [v'; W], c = (v0, w0) for kernels_cone.hip
    w.cone_A.ensure((size_t)(p->soc_q + 1) * u * D8); w.cone_c.ensure((size_t)(p->soc_q + 1) * D8);
    HIP_CHECK(hipMemcpyAsync(w.cone_A.p, p->soc_v, (size_t)u * D8, hipMemcpyDeviceToDevice, s));
    HIP_CHECK(hipMemcpyAsync(w.cone_A.d() + u, p->soc_W, (size_t)p->soc_q * u * D8, hipMemcpyDeviceToDevice, s));
    launch_fill(w.cone_c.d(), p->soc_v0, 1, s);  // (by value: no asynchronous read of the caller's struct)
    HIP_CHECK(hipMemcpyAsync(w.cone_c.d() + 1, p->soc_w0, (size_t)p->soc_q * D8, hipMemcpyDeviceToDevice, s));
    if (has_ub) launch_cone_drop_redundant_lo(w.su.lo.d(), w.cone_A.d(), w.cone_c.d(), (int)p->soc_q, (long long)M * N, u, s);
  }
  // (finish_now = false: the caller finishes — it may replace the last digits by cone rounds started from this iterate, see the dispatch)
  auto soc_interior_point = [&](bool finish_now) -> int {
    reset_scalars();
    // ---- stage-wise control cones: primal-dual path following on the same Riccati kernels (kernels_soc.hip) ----------
    const int q = (int)p->soc_q;
    w.Hadd.ensure(nu * u * D8); w.wu_soc.ensure(nu * D8);
    const size_t ncz = (size_t)M * N * (q + 1);
    for (DevBuf *b : {&w.soc_zl, &w.soc_zu, &w.soc_dzl, &w.soc_dzu, &w.soc_sl, &w.soc_su, &w.soc_dsl, &w.soc_dsu, &w.soc_cl, &w.soc_cu}) {
      b->ensure(nu * D8);
      HIP_CHECK(hipMemsetAsync(b->p, 0, nu * D8, s));
    }
    for (DevBuf *b : {&w.soc_zc, &w.soc_dzc, &w.soc_sc, &w.soc_dsc, &w.soc_cc}) {
      b->ensure(ncz * D8);
      HIP_CHECK(hipMemsetAsync(b->p, 0, ncz * D8, s));
    }
    SocArgs sa;
    memset(&sa, 0, sizeof(sa));
    sa.M = M; sa.N = N; sa.u = u; sa.Nc = Nc; sa.q = q; sa.owner = a.owner;
    sa.U = w.U.d(); sa.dU = w.dU.d(); sa.dU2 = w.dU2.d();
    sa.cl = w.soc_cl.d(); sa.cu = w.soc_cu.d(); sa.cc = w.soc_cc.d();
    sa.lo = has_ub ? su.lo : nullptr; sa.hi = has_ub ? su.hi : nullptr;
    sa.W = p->soc_W; sa.w0 = p->soc_w0; sa.v = p->soc_v; sa.v0 = p->soc_v0;
    sa.zl = w.soc_zl.d(); sa.zu = w.soc_zu.d(); sa.zc = w.soc_zc.d();
    sa.dzl = w.soc_dzl.d(); sa.dzu = w.soc_dzu.d(); sa.dzc = w.soc_dzc.d();
    sa.sl = w.soc_sl.d(); sa.su = w.soc_su.d(); sa.sc = w.soc_sc.d();
    sa.dsl = w.soc_dsl.d(); sa.dsu = w.soc_dsu.d(); sa.dsc = w.soc_dsc.d();
    sa.Hadd = w.Hadd.d(); sa.wu = w.wu_soc.d(); sa.fail = (int *)w.fail.p;
    a.Dx = a.wx = nullptr;
    a.Du = w.Hadd.d();  // full u x u blocks
    a.du_full = 1;
    a.wu = w.wu_soc.d();
    const unsigned long long one_bits = 0x4000000000000000ull;  // 2.0: upper end of the step kernel's search
    std::vector<double> hs(PMPC_RED_BLOCKS), hc(PMPC_RED_BLOCKS);
    double cone_cnt = 1.0;  // number of cones (degree of the complementarity measure), all ranks
    struct { unsigned long long amin; int fail; } host_rd;
    // complementarity mu = sum s'z / (number of cones: one per finite box side, one per stage cone), measured on the
    // device by the prepare kernel; cross-rank: summed
    auto measure = [&](int nblk, double &mu_out) -> int {
      HIP_CHECK(hipMemcpyAsync(hs.data(), w.part_sum.p, nblk * D8, hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipMemcpyAsync(hc.data(), w.part_cnt.p, nblk * D8, hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipMemcpyAsync(&host_rd.fail, w.fail.p, sizeof(int), hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipStreamSynchronize(s));
      double sum = 0.0, cnt = 0.0;
      for (int k = 0; k < nblk; k++) { sum += hs[k]; cnt += hc[k]; }
      if (c->multi()) {  // tiny host-staged all-reduce through the device (two doubles)
        double pair[2] = {sum, cnt};
        HIP_CHECK(hipMemcpyAsync(w.xch.p, pair, 2 * D8, hipMemcpyHostToDevice, s));
        allreduce(c, w.xch.p, 2, ncclFloat64, ncclSum);
        allreduce(c, w.fail.p, 1, ncclInt32, ncclMax);
        HIP_CHECK(hipMemcpyAsync(pair, w.xch.p, 2 * D8, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipMemcpyAsync(&host_rd.fail, w.fail.p, sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        sum = pair[0]; cnt = pair[1];
      }
      cone_cnt = std::max(cnt, 1.0);
      mu_out = sum / cone_cnt;
      return host_rd.fail;
    };
    const double mu_tol = 1e-12;  // (C mu with C ~ 3e3 on these problems: trajectories within ~3e-9)
    double mu = 1.0;
    int status = 1, newton = 0;
    // warm start as on the box path: the early iterate (mu <= 0.5) of the previous solve of this
    // shape — controls and duals; the slacks are recomputed from the new data — if it is strictly feasible for them
    const bool soc_warm_off = c->opt[OPT_WARM_START] == 0.0;
    const long long skey = ((((((long long)x * 131 + u) * 131 + N) * 1000003 + M) * 131 + Nc) * 8 + q) * 2 + (has_ub ? 1 : 0);
    bool warm = !soc_warm_off && !(p->flags & PMPC_COLD_START) && w.soc_key == skey, remembered = false;
    int nblk, fl;
  soc_restart:
    sa.mu = 1.0; sa.sigmu = 0.0;
    if (warm) {
      HIP_CHECK(hipMemcpyAsync(w.U.p, w.soc_wU.p, nu * D8, hipMemcpyDeviceToDevice, s));
      HIP_CHECK(hipMemcpyAsync(sa.zl, w.soc_wzl.p, nu * D8, hipMemcpyDeviceToDevice, s));
      HIP_CHECK(hipMemcpyAsync(sa.zu, w.soc_wzu.p, nu * D8, hipMemcpyDeviceToDevice, s));
      HIP_CHECK(hipMemcpyAsync(sa.zc, w.soc_wzc.p, ncz * D8, hipMemcpyDeviceToDevice, s));
    } else {
      launch_soc_fill_u(w.U.d(), p->soc_u_interior, (long long)nu, u, s);
    }
    if (fast) launch_rollout_fast(a, w.U.d(), w.X.d(), s);
    else launch_rollout(a, w.U.d(), w.X.d(), s);
    nblk = launch_soc_prepare(sa, warm ? 2 : 0, w.part_sum.d(), w.part_cnt.d(), s);  // cold: z = mu0 s^-1, a centred start
    fl = measure(nblk, mu);
    if (fl && warm) {  // the remembered controls are not strictly inside the new boxes / cones
      if (verbose) printf("pmpc_hip: stage cones: remembered iterate rejected, cold start\n");
      warm = false;
      w.soc_key = -1;
      HIP_CHECK(hipMemsetAsync(w.fail.p, 0, sizeof(int), s));
      goto soc_restart;
    }
    if (fl) return finish(fl == 3 ? 3 : 2);
    // the prepare pass above (cold / warm start) measured mu; from here on the pass that follows every update both
    // measures mu and builds the predictor system of the next iteration (a.corr = 0, sigma = 0)
    sa.corr = 0; sa.sigmu = 0.0;
    nblk = launch_soc_prepare(sa, 1, w.part_sum.d(), w.part_cnt.d(), s);
    fl = measure(nblk, mu);
    if (fl) { status = fl == 3 ? 3 : 2; }
    auto read_step = [&](double &amax) -> int {  // step length of the last step kernel (+ failure flag), across ranks
      if (c->multi()) allreduce(c, &sc->amin_bits, 1, ncclFloat64, ncclMin);  // bit pattern of a non-negative double
      HIP_CHECK(hipMemcpyAsync(&host_rd.amin, &sc->amin_bits, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipMemcpyAsync(&host_rd.fail, w.fail.p, sizeof(int), hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipStreamSynchronize(s));
      memcpy(&amax, &host_rd.amin, sizeof(double));
      return host_rd.fail;
    };
    for (int it = 0; it < 100 && status == 1; it++) {
      // ---- predictor: factorisation, affine step, step polynomial, second-order terms -------------------------------
      a.dX = w.dX.d(); a.dU = w.dU.d();
      structured_solve(c, a, true, fast);
      inf.structured_solves++;
      HIP_CHECK(hipMemcpyAsync(&sc->amin_bits, &one_bits, sizeof(one_bits), hipMemcpyHostToDevice, s));
      sa.corr = 0; sa.sigmu = 0.0;
      nblk = launch_soc_step(sa, &sc->amin_bits, w.part_sum.d(), w.part_cnt.d(), s);
      HIP_CHECK(hipMemcpyAsync(hs.data(), w.part_sum.p, nblk * D8, hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipMemcpyAsync(hc.data(), w.part_cnt.p, nblk * D8, hipMemcpyDeviceToHost, s));
      double a_aff;
      if (read_step(a_aff)) { status = 2; break; }
      double s1 = 0.0, s2 = 0.0;
      for (int k = 0; k < nblk; k++) { s1 += hs[k]; s2 += hc[k]; }
      if (c->multi()) {
        double pair[2] = {s1, s2};
        HIP_CHECK(hipMemcpyAsync(w.xch.p, pair, 2 * D8, hipMemcpyHostToDevice, s));
        allreduce(c, w.xch.p, 2, ncclFloat64, ncclSum);
        HIP_CHECK(hipMemcpyAsync(pair, w.xch.p, 2 * D8, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        s1 = pair[0]; s2 = pair[1];
      }
      a_aff = std::min(1.0, a_aff);
      const double mu_aff = mu + a_aff * (s1 + a_aff * s2) / cone_cnt;  // (S0 + a S1 + a^2 S2) / deg
      double sigma = mu_aff / mu;
      sigma = std::min(1.0, std::max(0.0, sigma * sigma * sigma));
      // ---- corrector: difference step on the same factorisation -----------------------------------------------------
      sa.corr = 1; sa.sigmu = sigma * mu;
      launch_soc_prepare(sa, 1, w.part_sum.d(), w.part_cnt.d(), s);
      a.dX = w.dX2.d(); a.dU = w.dU2.d();
      structured_solve(c, a, false, fast);
      a.dX = w.dX.d(); a.dU = w.dU.d();
      HIP_CHECK(hipMemcpyAsync(&sc->amin_bits, &one_bits, sizeof(one_bits), hipMemcpyHostToDevice, s));
      launch_soc_step(sa, &sc->amin_bits, w.part_sum.d(), w.part_cnt.d(), s);
      double amax;
      if (read_step(amax)) { status = 2; break; }
      const double alpha = std::min(1.0, 0.99 * amax);  // strictly inside the cones, also when the boundary is just beyond 1
      if (!(alpha > 0.0)) { status = 2; break; }
      launch_soc_update(sa, alpha, w.X.d(), w.dX.d(), w.dX2.d(), w.U.d(), (long long)nx, (long long)nu, (long long)ncz, s);
      newton++;
      // ---- complementarity of the new iterate + the next predictor system ------------------------------------------
      sa.corr = 0; sa.sigmu = 0.0;
      nblk = launch_soc_prepare(sa, 1, w.part_sum.d(), w.part_cnt.d(), s);
      double mu_new;
      fl = measure(nblk, mu_new);
      if (fl) {  // round-off pushed a pair onto its cone boundary: accept what has been reached if that is the end game
        status = (mu <= 1e2 * mu_tol) ? 0 : (fl == 3 ? 3 : 2);
        break;
      }
      if (verbose) printf("pmpc_hip: soc it %3d  mu %9.3e -> %9.3e  alpha_aff %6.4f  sigma %8.2e  alpha %6.4f\n", newton, mu, mu_new, a_aff, sigma, alpha);
      const bool stalled = alpha < 1e-3 && mu <= 1e2 * mu_tol;  // at the precision floor
      mu = mu_new;
      if (!remembered && !soc_warm_off && mu <= 0.5) {
        for (DevBuf *b : {&w.soc_wU, &w.soc_wzl, &w.soc_wzu}) b->ensure(nu * D8);
        w.soc_wzc.ensure(ncz * D8);
        HIP_CHECK(hipMemcpyAsync(w.soc_wU.p, w.U.p, nu * D8, hipMemcpyDeviceToDevice, s));
        HIP_CHECK(hipMemcpyAsync(w.soc_wzl.p, sa.zl, nu * D8, hipMemcpyDeviceToDevice, s));
        HIP_CHECK(hipMemcpyAsync(w.soc_wzu.p, sa.zu, nu * D8, hipMemcpyDeviceToDevice, s));
        HIP_CHECK(hipMemcpyAsync(w.soc_wzc.p, sa.zc, ncz * D8, hipMemcpyDeviceToDevice, s));
        w.soc_key = skey;
        remembered = true;
      }
      if (mu <= mu_tol || stalled) { status = 0; break; }
    }
    if (status != 0 && warm) {  // a warm-started run that fails is repeated cold
      if (verbose) printf("pmpc_hip: stage cones: warm-started run failed (status %d), cold start\n", status);
      warm = false; remembered = false; status = 1; newton = 0;
      w.soc_key = -1;
      HIP_CHECK(hipMemsetAsync(w.fail.p, 0, sizeof(int), s));
      goto soc_restart;
    }
    inf.ipm_iters = newton;
    inf.mu = mu;
    if (verbose) printf("pmpc_hip: stage cones: status %d after %d Newton steps\n", status, newton);
    return (finish_now || status != 0) ? finish(status) : 0;
  };

  // returns 0: the equality-only optimum satisfies every box (done), 1: boxes violated (interior-point phase), 2: failure
  auto equality_phase = [&]() -> int {
    reset_scalars();
    equality_solve();
    if (has_xb) launch_violation(sx, w.part_max.d(), s);
    if (has_ub) launch_violation(su, w.part_max.d() + B, s);
    exchange(c, 1);  // (without boxes: only for the failure flag)
    read_scalars(c);
    inf.max_violation = c->sc_host->viol_max;
    if (*c->fail_host || !(c->sc_host->viol_max == c->sc_host->viol_max)) return 2;
    if (verbose) printf("pmpc_hip: equality-only optimum, max bound violation %.3e\n", c->sc_host->viol_max);
    if (!has_xb && !has_ub) return 0;
    if (c->sc_host->viol_max <= 0.0 && mu_target == 0.0) return 0;  // (a barrier acts on feasible points too)
    return 1;
  };
  // ---- primal-dual active-set iteration on the control boxes (kernels_ipm.hip, k_as_*) -----------------------------------
  // Given a guess of the active set, ONE structured solve from a base point with those controls ON their bounds gives the
  // exact optimum on that set; a check pass verifies the KKT signs and, where they fail, applies the primal-dual active-set
  // update (release negative multipliers, hold violated boxes).  An unchanged set is the optimum of the QP, complementarity
  // exactly zero.  Three uses: (a) WARM START — the accepted set and solution of the previous solve of this shape (consecutive
  // SCP sub-problems differ in a few hundred to a few thousand of ~1e6 entries) start the next solve directly: no equality-only
  // phase, no interior-point iteration; (b) COLD START — without one, the boxes the equality-only optimum violates are the
  // first guess; (c) FINISH of the interior-point iteration — once mu <= polish_mu * mu_peak its iterate names the set
  // (l > slack), which replaces the last predictor-corrector iterations (4 sweeps each) by a few factor + forward sweeps.
  // If the set does not settle the interior-point iteration runs (on), its state untouched.  The rounds act on the control
  // boxes (a state cannot be moved onto its bound without leaving the dynamics; state boxes that do not bind are verified at
  // acceptance, see below); not in barrier mode.  DESIGN.md section 2.4.
  const long long as_key_pre = (((((((long long)x * 131 + u) * 131 + N) * 1000003 + M) * 131 + Nc) * 2 + (fast ? 1 : 0)) * 2 + (has_xb ? 1 : 0)) * 64 + (soc ? 1 + (long long)p->soc_q + 8 * (long long)cone_rows : 0);
  const double polish_mu = c->opt[OPT_POLISH_MU];  // 0 switches both uses off
  const bool as_warm_on = c->opt[OPT_AS_WARM] != 0.0, as_skip_on = c->opt[OPT_AS_SKIP] != 0.0, as_defect_on = c->opt[OPT_AS_DEFECT] != 0.0;
  // State boxes: a state cannot be held on its bound this way, but boxes that are there and INACTIVE at the optimum (loose
  // limits, e.g. x in +-20 of the reference's tests/pmpcjl_test.py:164-219) change nothing: the accepted point only has to
  // be checked against them.  A violated state box sends the solve (and later solves of this shape) to the interior-point path.
  const bool polish_on = polish_mu > 0.0 && (has_ub || xbox_as) && mu_target == 0.0 && !(has_xb && !xbox_as && w.xb_block_key == as_key_pre);
  const long long as_key = as_key_pre;
  // mode 1: guess from the interior-point iterate in (w.U, slacks, multipliers); mode 0: the stored set, base point = w.U
  // (the previous solution).  Returns 0 accepted (w.X, w.U hold the optimum), 1 not settled, 2 numerical failure.
  // Fast path: the rounds run on the device's own decisions (k_as_ctl); the host enqueues as many rounds as the previous
  // solve of this shape took before it reads anything back, every kernel of a round that is no longer needed returns at once.
  // The base point lives in the caller's output buffers (the forward sweep writes base + step there), so an accepted round
  // leaves nothing to copy.  Returns 0 accepted, 1 not settled, 2 numerical failure.
  // xb (problems with state boxes on the XBOX sweeps): 1 state rows on, from the stored statuses / multipliers (mode 0), the
  // interior-point iterate (mode 1) or nothing (mode 2); 0 state boxes IGNORED (first phase of a cold start, see below); 2 on, nothing
  // stored.  mode 4: continue from the point an accepted attempt left in the output buffers (second phase of that cold start).
  auto active_set_fast = [&](double dual_scale, int mode, int max_rounds, int xb = 1) -> int {
    const double big = 1e30, tol_p = 1e-13;
    w.as_act.ensure(nu * sizeof(int) + 8); w.as_cntp.ensure((size_t)M * 3 * sizeof(int)); w.as_settled.ensure((size_t)M * sizeof(int));
    w.as_ctl.ensure(sizeof(AsCtl)); w.as_delta.ensure((size_t)std::max(nc, 1) * D8);
    int *act = (int *)w.as_act.p;
    AsCtl *ctl = (AsCtl *)w.as_ctl.p;
    LQArgs b = a;
    b.Dx = b.wx = b.Du = b.wu = nullptr; b.du_full = 0;
    b.as_act = act; b.as_lo = su.lo; b.as_hi = su.hi; b.as_cnt = (int *)w.as_cntp.p; b.as_big = big; b.as_tol_p = tol_p;
    b.as_settled_out = (int *)w.as_settled.p; b.as_delta = w.as_delta.d(); b.as_ctl = ctl; b.done = &ctl->done;
    w.as_viol.ensure((size_t)M * D8);
    b.as_viol = w.as_viol.d();
    b.Xb = p->X_out; b.Ub = p->U_out; b.Xo = p->X_out; b.Uo = p->U_out;
    {  // checkpointed restart of the later rounds' factor sweeps (kernels_as.hip)
      int slots = 0;  // stages FIRST << k <= N - 1
      if (c->opt[OPT_AS_CKPT] != 0.0 && as_skip_on && nc <= 32)
        while ((PMPC_AS_CK_FIRST << slots) <= N - 1) slots++;
      if (slots > 0) {
        const int ks = (x + 3) / 4;
        w.as_ck.ensure((size_t)M * slots * (64 * ks + 32) * D8);
        w.as_jhi.ensure((size_t)M * sizeof(int));
        if (w.ck_stat.ensure(4 * sizeof(unsigned long long))) HIP_CHECK(hipMemsetAsync(w.ck_stat.p, 0, 4 * sizeof(unsigned long long), s));
        b.as_ck = w.as_ck.d(); b.as_jhi = (int *)w.as_jhi.p; b.ck_slots = slots; b.ck_stat = (unsigned long long *)w.ck_stat.p;
      }
    }
    w.as_key = -1;
    // stage cones (mode 0 warm / 3 cold): Newton terms per round from kernels_cone.hip, see the header there
    const bool cone = cone_as && (mode == 0 || mode == 3 || mode == 5);
    b.as_freeze_tol = (cone && Nc == 1) ? c->opt[OPT_AS_FREEZE_TOL] : 0.0;
    ConeArgs ca;
    memset(&ca, 0, sizeof(ca));
    if (cone) {
      const int q = (int)p->soc_q;
      const size_t rows = (size_t)M * N;
      w.Hadd.ensure(nu * u * D8); w.wu_soc.ensure(nu * D8); w.cone_uraw.ensure(nu * D8); w.as_open.ensure((size_t)M * sizeof(int));
      const bool z_new = w.cone_z.ensure(rows * cone_rows * D8), rec_new = w.cone_rec.ensure(rows * std::max(ncones, 1) * PMPC_CONE_REC * D8);
      if (z_new || rec_new) w.as_key = -1;
      b.cone_H = w.Hadd.d(); b.cone_g = w.wu_soc.d(); b.as_uraw = w.cone_uraw.d(); b.as_open = (int *)w.as_open.p;
      ca.M = M; ca.N = N; ca.u = u; ca.q = q; ca.Nc = Nc; ca.owner = a.owner;
      ca.rows = cone_rows;
      if (ncones > 0) {
        ca.ncones = ncones; ca.per_stage = p->cone_per_stage ? 1 : 0;
        for (int k = 0; k < ncones; k++) ca.qs[k] = p->cone_sizes[k];
        ca.A = p->cone_A; ca.c = p->cone_c;
      } else {
        ca.ncones = 1; ca.qs[0] = q; ca.per_stage = 0;
        ca.A = w.cone_A.d(); ca.c = w.cone_c.d();
      }
      ca.R = p->R; ca.r32 = a.mat32; ca.reg_u = p->reg_u; ca.rho_scale = 1e7;
      ca.z = w.cone_z.d(); ca.rec = w.cone_rec.d(); ca.H = w.Hadd.d(); ca.g = w.wu_soc.d();
      ca.cnt = (int *)w.as_cntp.p; ca.settled = (int *)w.as_settled.p; ca.open = (int *)w.as_open.p; ca.done = &ctl->done; ca.ctl = ctl;
      ca.jhi = b.as_jhi;
      // (measured at config E: 1e-6 .. 1e-3 changes the round count by 7.25 -> 6.75 only — the rounds behind the last status change are the Newton iteration itself)
      ca.tol_step = 1e-6; ca.tol_phi = 1e-9;
      ca.dual_scale = dual_scale;
    }
    // state boxes: penalty + multiplier terms per round from kernels_xbox.hip, see the header there
    const bool xbox = xbox_as && xb != 0;
    {  // sensitivity records of the forward sweep (k_fwd_as<.., SENS>): worth their stores when later rounds are expected and the sweeps are
       // issue-bound (many waves per SIMD); a small shard's rounds sit at one wave's latency whatever the settled particles do
      const int min_m = (int)c->opt[OPT_AS_SENS_MIN_M];
      if (min_m > 0 && M >= min_m && as_skip_on && Nc == 1 && (mode != 0 || w.as_pred_rounds >= 2)) {
        w.as_T.ensure((size_t)M * N * 64 * D8);
        b.as_T = w.as_T.d();
      }
    }
    XboxArgs xa;
    memset(&xa, 0, sizeof(xa));
    if (xbox) {
      w.as_open.ensure((size_t)M * sizeof(int)); w.xb_D.ensure(nx * D8); w.xb_g.ensure(nx * D8);
      const bool z_new = w.xb_z.ensure(nx * D8), st_new = w.xb_st.ensure(nx * sizeof(int));
      if ((z_new || st_new) && mode == 0) return 1;
      if (mode == 2 || xb == 2) {  // cold: nothing held, no multipliers — the first pass holds what the base point violates
        HIP_CHECK(hipMemsetAsync(w.xb_z.p, 0, nx * D8, s));
        HIP_CHECK(hipMemsetAsync(w.xb_st.p, 0, nx * sizeof(int), s));
      } else if (mode == 1) {
        launch_xbox_from_ipm(sx, (int *)w.xb_st.p, w.xb_z.d(), s);
      }
      b.xb_D = w.xb_D.d(); b.xb_g = w.xb_g.d(); b.as_open = (int *)w.as_open.p;  // (the merged exchange of a sharded run carries the open rows: tail[4])
      xa.M = M; xa.N = N; xa.x = x; xa.lo = p->lx; xa.hi = p->ux; xa.Q = p->Q; xa.pw = p->weights; xa.reg_x = p->reg_x; xa.rho_scale = 1e7;  // (measured, bench.py --vmax: 1e5 .. 1e2 only add rounds)
      w.xb_qmax.ensure((size_t)M * D8);
      xa.qmax = w.xb_qmax.d();
      xa.z = w.xb_z.d(); xa.st = (int *)w.xb_st.p; xa.D = w.xb_D.d(); xa.g = w.xb_g.d();
      xa.cnt = (int *)w.as_cntp.p; xa.settled = (int *)w.as_settled.p; xa.open = (int *)w.as_open.p; xa.done = &ctl->done; xa.ctl = ctl;
      xa.jhi = b.as_jhi;
      xa.tol = 1e-9; xa.dual_scale = dual_scale;
      // (a held row stays open while |s| > tol; its multiplier moves by rho s, so the second test only matters for rows with a small multiplier.
      //  Measured, bench.py --vmax 3 / 2: 1e-6 costs one more round per solve than 1e-3 (709 -> 777 it/s, 267 -> 295), same answers to 1e-9)
      xa.z_tol = 1e-3;
      // (measured on bench.py --vmax 2: 302 it/s with both, 175 without the first, 302 -> 396 and no interior-point iteration at all with the second)
      // — for genuine state rows (a velocity limit violated over a window of stages).  In the increment form of a slew problem the boxes on
      // the u-part of the state are the control boxes, each moved by its own increment: there the plain rule (hold everything violated) settles in 7-11 rounds
      // and partial activation only delays it (tools/debug/slew_paths.py: cold start back on the interior-point iteration)
      xa.keep_on_clamp = 1;
      xa.act_frac = 0.5;
      xa.ctrl_from = c->xb_ctrl_from >= 0 ? c->xb_ctrl_from : x;
    }
    // (a cold start on genuine state rows that does not contract is not worth its rounds: the interior-point iteration takes over and
    //  names a better first set; the control boxes of a slew problem in increment form do settle, in 7-11 rounds that need not contract one by one)
    launch_as_begin(ctl, (int *)w.fail.p, max_rounds, dual_scale, s, cone ? 8 : (xbox ? (c->xb_ctrl_from >= 0 ? 8 : (mode == 0 ? 6 : 3)) : 2));  // control block of this attempt (+ cleared failure flag)
    // warm start inside an SCP loop (PMPC_PREV_IS_LAST_SOLUTION): the base point is the linearisation point itself, whose
    // dynamics defect f - X_prev is elementwise and rides through the first round's sweeps — no sequential rollout, nothing
    // written before the sweep.  The forward sweep verifies that U_prev IS the base point of the stored set.
    // (with several consensus stages the condensed gradient of stage j also needs Y_j d_{j-1}, d = the defect propagated
    // FORWARD through the earlier consensus stages — a term no backward sweep can form: the condensing kernel, which walks
    // those stages forward anyway, carries d as one more column (k_cond_fast).  Found by the config-B full-consensus test,
    // which a single accepted round without the term got wrong by 8 %)
    const bool use_defect = mode == 0 && as_defect_on && (p->flags & PMPC_PREV_IS_LAST_SOLUTION);
    if (a.mat32 && !use_defect) return 1;  // (fp32 storage: no rollout kernel reads the float arrays)
    if (mode == 3) {  // cold start of the cone rounds: every control at the caller's interior point, nothing held, no multipliers
      ProfScope ps(c, 5);
      HIP_CHECK(hipMemsetAsync(act, 0, nu * sizeof(int) + 8, s));
      HIP_CHECK(hipMemsetAsync(w.cone_z.p, 0, (size_t)M * N * cone_rows * D8, s));
      if (p->soc_u_interior) launch_soc_fill_u(p->U_out, p->soc_u_interior, (long long)nu, u, s);
      else launch_init_base(p->U_out, p->U_prev, M, N, u, Nc, s);  // (any start will do for the rounds; the shared controls need ONE base value: 0)
      launch_rollout_fast(b, p->U_out, p->X_out, s);
    } else if (mode == 4) {
      // the base point is what the attempt that just ended left in the outputs: its controls on their bounds, its states rolled out
    } else if (!use_defect) {  // first base point: controls snapped into their boxes / onto their bounds, states by rollout
      ProfScope ps(c, 5);
      Slab st = su;
      // the previous solution: this context's copy, or — a caller inside an SCP loop that hands it back as U_prev (promise flag;
      // no copy was kept then) with a consensus horizon the no-rollout start above does not cover — the caller's U_prev
      st.z = (mode == 0 && !w.as_U_valid) ? const_cast<double *>(p->U_prev) : w.U.d();
      st.D = nullptr; st.w = nullptr;
      if (mode == 5) {  // finish of the cone path-following iteration: box statuses from ITS duals, cone multipliers = its cone duals
        st.ll = w.soc_zl.d(); st.lu = w.soc_zu.d();
        HIP_CHECK(hipMemcpyAsync(w.cone_z.p, w.soc_zc.p, (size_t)M * N * cone_rows * D8, hipMemcpyDeviceToDevice, s));
      }
      launch_as_setup(st, mode == 5 ? 1 : mode, 0, act, p->U_out, big, s);
      launch_rollout_fast(b, p->U_out, p->X_out, s);
    }
    // (same conditions as the in-wave consensus solve of structured_solve: one rank, one consensus stage)
    const bool fuse_env = c->opt[OPT_AS_FUSE_CTL] != 0.0 && c->opt[OPT_AS_WAVE_CONS] != 0.0;
    const bool fuse_ctl = fuse_env && !c->multi() && Nc == 1;
    c->as_pend.ctl = nullptr;
    const int *open_part = (cone || xbox) ? (const int *)w.as_open.p : nullptr;
    if (xbox) {  // terms of the first round, from the first base point and the stored statuses / multipliers
      ProfScope ps(c, 5);
      xa.finish = 0;
      xa.X = use_defect ? p->X_prev : p->X_out;
      launch_xbox_step(xa, s);
    }
    if (cone) {  // Newton terms of the first round, from the first base point and the stored multipliers
      ProfScope ps(c, 5);
      ca.finish = 0;
      ca.U = use_defect ? p->U_prev : p->U_out;
      launch_cone_step(ca, s);
    }
    const int perm_min_m = (int)c->opt[OPT_AS_PERM_MIN_M];
    int round = 0, depth = mode == 0 ? std::max(1, std::min(w.as_pred_rounds, max_rounds)) : std::min(3, max_rounds);
    if (verbose > 1 && xbox) depth = 1;  // (the debugging dump below wants every round)
    static const bool duc_trace = getenv("PMPC_DUC_TRACE") != nullptr;
    if (duc_trace) depth = 1;
    int n_batches_at_hook = -1000;  // batches waited for since the speculation hook fired (in THIS attempt)
    AsCtl h;
    memset(&h, 0, sizeof(h));
    while (true) {
      const int batch = std::min(depth, max_rounds - round);
      for (int k = 0; k < batch; k++) {
        const int r = round + k;
        b.defect = (use_defect && r == 0) ? p->f : nullptr;
        // particles without a status change in the previous round keep their factors, their condensed Hessian H_i and their
        // conditional optimum: no factor sweep for them — g_i follows the applied consensus step, g_i += H_i delta
        const bool skip = as_skip_on && r > 0 && nc <= 32;
        b.as_settled_in = skip ? (const int *)w.as_settled.p : nullptr;
        // the unsettled particles first: their sweeps are the launch's long waves (kernels_as.hip, k_as_perm)
        b.as_perm = nullptr;
        if (skip && perm_min_m > 0 && M >= perm_min_m && b.as_T) {  // (with every particle sweeping the index order is the better one: memory locality)
          ProfScope pp(c, 5);
          if (w.as_perm.ensure((size_t)M * sizeof(int)) || w.as_perm_m != M) {  // (what the buffer holds must be a permutation of 0 .. M-1 at any time)
            std::vector<int> id(M);
            for (int q_ = 0; q_ < M; q_++) id[q_] = q_;
            HIP_CHECK(hipMemcpyAsync(w.as_perm.p, id.data(), (size_t)M * sizeof(int), hipMemcpyHostToDevice, s));
            HIP_CHECK(hipStreamSynchronize(s));
            w.as_perm_m = M;
          }
          static const bool perm_fused = !(getenv("PMPC_AS_PERM_FUSED") && atoi(getenv("PMPC_AS_PERM_FUSED")) == 0);  // (A/B switch)
          if (c->as_pend.ctl && perm_fused) {
            // the round control of the round before rides in this round's consensus-partials launch (between the factor and the forward
            // sweep): the order is computed there, one launch less per round — the forward sweep gets it fresh, the factor sweep (short
            // restarted sweeps) runs in the order of the round before
            c->as_pend.settled = (const int *)w.as_settled.p;
            c->as_pend.perm = (int *)w.as_perm.p;
          } else {
            launch_as_perm((const int *)w.as_settled.p, M, (int *)w.as_perm.p, &ctl->done, s);
          }
          b.as_perm = (const int *)w.as_perm.p;
        }
        const bool last = k == batch - 1;
        // sharded with a consensus horizon: {released, activated, bad, failure} of round r ride in round r + 1's consensus
        // all-reduce (structured_solve), the decision about round r follows it there; only the last round of a batch needs a
        // collective of its own.  Every rank takes the same decisions from the same sums.
        const bool merge = c->multi() && nc > 0;
        b.as_merge = merge ? (k == 0 ? 1 : 2) : 0;  // (the first round of a batch carries nothing: the previous batch closed its last round)
        c->as_seq++;
        structured_solve(c, b, true, true, /*prep_done=*/true);
        ProfScope ps(c, 5);
        if (cone) {  // finish this round's cones (multipliers, cases, counters — BEFORE the round control reads them), prepare the next
          ca.finish = 1;
          ca.U = p->U_out; ca.Uraw = w.cone_uraw.d();
          launch_cone_step(ca, s);
        }
        if (xbox) {
          xa.finish = 1;
          xa.X = p->X_out;
          launch_xbox_step(xa, s);
        }
        if (merge) {
          if (last) {
            double *tl = w.Hg.d() + (size_t)nc * nc + nc;
            launch_as_ctl(ctl, (const int *)w.as_cntp.p, M, (const int *)w.fail.p, 1, 0, 0, nullptr, nullptr, 0, s, tl, nullptr, open_part);
            allreduce(c, tl, 5, ncclFloat64, ncclSum);
            launch_as_ctl(ctl, nullptr, M, (const int *)w.fail.p, 0, 1, 1, &c->mirror_dev->ctl, &c->mirror_dev->as_seq, c->as_seq, s, tl);
          }
        } else if (c->multi()) {  // no consensus exchange to ride on: one sum for all five (open cones + the four counters: contiguous)
          launch_as_ctl(ctl, (const int *)w.as_cntp.p, M, (const int *)w.fail.p, 1, 0, 0, nullptr, nullptr, 0, s, nullptr, nullptr, open_part);
          allreduce(c, &ctl->open, 5, ncclInt32, ncclSum);
          launch_as_ctl(ctl, nullptr, M, (const int *)w.fail.p, 0, 1, last ? 1 : 0, &c->mirror_dev->ctl, &c->mirror_dev->as_seq, c->as_seq, s);
        } else if (!last && fuse_ctl) {
          // the decision about this round rides in the next round's consensus-partials launch (structured_solve): its factor
          // sweep does not need it (settled particles leave it at once), its forward sweep sees it
          c->as_pend = AsCtlCall{ctl, (const int *)w.as_cntp.p, M, (const int *)w.fail.p, &c->mirror_dev->ctl, &c->mirror_dev->as_seq, c->as_seq,
                                 b.as_viol, open_part, nullptr, nullptr};
        } else {
          launch_as_ctl(ctl, (const int *)w.as_cntp.p, M, (const int *)w.fail.p, 1, 1, last ? 1 : 0, &c->mirror_dev->ctl, &c->mirror_dev->as_seq, c->as_seq, s,
                        nullptr, b.as_viol, open_part);
        }
      }
      if (round == 0 && c->post_batch) {  // the caller's follow-up work goes in behind the rounds before anything is read back
        std::function<void()> hook;
        hook.swap(c->post_batch);
        c->spec_fired = true;
        n_batches_at_hook = 0;
        hook();
      }
      n_batches_at_hook++;
      // the control block is published when the rounds are over (done) or at the end of the batch, whichever comes first,
      // with the sequence number of the round that published it: wait for any of this batch's numbers
      {
        const unsigned long long lo_seq = c->as_seq - (unsigned long long)batch + 1, hi_seq = c->as_seq;
        const bool seen = spin_until([&] {
          const unsigned long long v = *(volatile unsigned long long *)&c->mirror->as_seq;
          return v >= lo_seq && v <= hi_seq;
        });
        if (!seen) {
          HIP_CHECK(hipStreamSynchronize(s));
          const unsigned long long v = *(volatile unsigned long long *)&c->mirror->as_seq;
          if (!(v >= lo_seq && v <= hi_seq)) throw PmpcHipError{-1, "active-set control block never published", __FILE__, __LINE__};
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        memcpy(&h, (const void *)&c->mirror->ctl, sizeof(h));
      }
      if (verbose)
        for (int r = round; r < h.round && r < 16; r++)
          printf("pmpc_hip: active set (%s) round %d: %d released, %d activated (largest violation behind a change %.2e)\n",
                 mode == 5 ? "finish" : (mode == 4 ? "state rows" : (mode >= 2 ? "cold" : (mode ? "finish" : "warm"))), r + 1, h.hist[r][0], h.hist[r][1], h.worst[r]);
      if (verbose && cone) printf("pmpc_hip: active set: %d stage cones still open after round %d\n", h.open, h.round);
      if (duc_trace && nc > 0 && !a.cons_G) {
        std::vector<double> hd(nc);
        HIP_CHECK(hipMemcpy(hd.data(), w.as_delta.p, nc * D8, hipMemcpyDeviceToHost));  // (the step of the shared controls as APPLIED by the round's forward sweep)
        double m = 0.0;
        for (double v : hd) m = std::max(m, std::fabs(v));
        int nset = 0;
        { std::vector<int> hs(M); HIP_CHECK(hipMemcpy(hs.data(), w.as_settled.p, M * sizeof(int), hipMemcpyDeviceToHost)); for (int v : hs) nset += v; }
        printf("pmpc_hip: trace: round %d max |du_c| %.3e, %d of %d particles settled\n", h.round, m, nset, M);
        if (b.as_jhi) {  // histogram of the highest changed stage among the unsettled particles
          std::vector<int> hj(M), hs(M), hist(N + 1, 0);
          HIP_CHECK(hipMemcpy(hj.data(), w.as_jhi.p, M * sizeof(int), hipMemcpyDeviceToHost));
          HIP_CHECK(hipMemcpy(hs.data(), w.as_settled.p, M * sizeof(int), hipMemcpyDeviceToHost));
          for (int q = 0; q < M; q++) if (!hs[q]) hist[hj[q] < 0 ? N : hj[q]]++;
          printf("pmpc_hip: trace:   highest changed stage:");
          for (int q = 0; q <= N; q++) if (hist[q]) printf(" %d:%d", q == N ? -1 : q, hist[q]);
          printf("\n");
        }
      }
      if (verbose > 1 && xbox) {  // debugging aid: the state rows after this batch — held rows, largest multiplier, largest |x|, per worst particle
        HIP_CHECK(hipStreamSynchronize(s));
        std::vector<double> hz(nx), hX(nx), hU(nu);
        std::vector<int> hs(nx), hact(nu);
        HIP_CHECK(hipMemcpy(hz.data(), w.xb_z.p, nx * D8, hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(hs.data(), w.xb_st.p, nx * sizeof(int), hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(hX.data(), p->X_out, nx * D8, hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(hU.data(), p->U_out, nu * D8, hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(hact.data(), w.as_act.p, nu * sizeof(int), hipMemcpyDeviceToHost));
        int held = 0, uheld = 0, wi = 0;
        double zmax = 0.0, xmax = 0.0;
        for (size_t k = 0; k < nx; k++) {
          held += hs[k] != 0;
          if (hz[k] > zmax) { zmax = hz[k]; wi = (int)(k / ((size_t)N * x)); }
          xmax = std::max(xmax, std::fabs(hX[k]));
        }
        for (size_t k = 0; k < nu; k++) uheld += hact[k] != 0;
        printf("   state rows after round %d: %d held, %d controls held, largest multiplier %.3e (particle %d), largest |x| %.3e\n", h.round, held, uheld, zmax, wi, xmax);
        if (getenv("PMPC_XB_DUMP")) {
          const int pi = atoi(getenv("PMPC_XB_DUMP"));
          for (int j = 0; j < N; j++) {
            printf("     p%d j%2d st", pi, j);
            for (int r = 0; r < x; r++) printf(" %d", hs[((size_t)pi * N + j) * x + r]);
            printf(" | act");
            for (int r = 0; r < u; r++) printf(" %d", hact[((size_t)pi * N + j) * u + r]);
            printf(" | v");
            for (int r = 3; r < 6 && r < x; r++) printf(" %+.4f", hX[((size_t)pi * N + j) * x + r]);
            printf(" | z");
            for (int r = 3; r < 6 && r < x; r++) printf(" %.3e", hz[((size_t)pi * N + j) * x + r]);
            printf(" | u");
            for (int r = 0; r < u; r++) printf(" %+.4f", hU[((size_t)pi * N + j) * u + r]);
            printf("\n");
          }
        }
      }
      if (verbose > 1 && cone) {  // debugging aid: the active cones' records (small problems only)
        HIP_CHECK(hipStreamSynchronize(s));
        const int q1 = cone_rows;
        std::vector<double> hz((size_t)M * N * q1), hr((size_t)M * N * std::max(ncones, 1) * PMPC_CONE_REC), hu(nu), hraw(nu), hg(nu);
        HIP_CHECK(hipMemcpy(hz.data(), w.cone_z.p, hz.size() * D8, hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(hr.data(), w.cone_rec.p, hr.size() * D8, hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(hu.data(), p->U_out, nu * D8, hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(hraw.data(), w.cone_uraw.p, nu * D8, hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(hg.data(), w.wu_soc.p, nu * D8, hipMemcpyDeviceToHost));
        {
          std::vector<int> hcnt((size_t)M * 3);
          HIP_CHECK(hipMemcpy(hcnt.data(), w.as_cntp.p, hcnt.size() * sizeof(int), hipMemcpyDeviceToHost));
          std::vector<int> hact(nu);
          HIP_CHECK(hipMemcpy(hact.data(), w.as_act.p, nu * sizeof(int), hipMemcpyDeviceToHost));
          printf("   act:");
          for (size_t k = 0; k < nu && k < 24; k++) printf(" %d", hact[k]);
          printf("\n   as_cnt:");
          for (int v : hcnt) printf(" %d", v);
          printf(" | U of particle 0:");
          for (int k = 0; k < N * u && k < 12; k++) printf(" %.6e", hu[k]);
          printf(" | uraw:");
          for (int k = 0; k < N * u && k < 12; k++) printf(" %.6e", hraw[k]);
          printf(" | cone_g:");
          for (int k = 0; k < N * u && k < 12; k++) printf(" %.3e", hg[k]);
          printf("\n");
        }
        int shown = 0;
        for (size_t k = 0; k < (size_t)M * N && shown < 6; k++) {
          if (M > 64 ? hr[k * PMPC_CONE_REC + 11] < 3.0 : hr[k * PMPC_CONE_REC] == 0.0) continue;  // (large problems: the cones that keep changing case)
          printf("   [flips %g]", hr[k * PMPC_CONE_REC + 11]);
          shown++;
          printf("   cone (%zu,%zu) case %g rho %.3e curv %.3e nu %.6e | s_b", k / N, k % N, hr[k * PMPC_CONE_REC], hr[k * PMPC_CONE_REC + 1], hr[k * PMPC_CONE_REC + 2], hr[k * PMPC_CONE_REC + 3]);
          for (int r = 0; r < q1; r++) printf(" %.9e", hr[k * PMPC_CONE_REC + 4 + (q1 - 1) + r]);
          printf(" | z");
          for (int r = 0; r < q1; r++) printf(" %.9e", hz[k * q1 + r]);
          printf(" | u");
          for (int r = 0; r < u; r++) printf(" %.9e", hu[k * u + r]);
          printf(" | uraw");
          for (int r = 0; r < u; r++) printf(" %.9e", hraw[k * u + r]);
          printf(" | g");
          for (int r = 0; r < u; r++) printf(" %.3e", hg[k * u + r]);
          {
            std::vector<double> hk(u), hH((size_t)u * u);
            std::vector<int> ha(u);
            HIP_CHECK(hipMemcpy(hk.data(), w.kff.d() + k * u, u * D8, hipMemcpyDeviceToHost));
            HIP_CHECK(hipMemcpy(ha.data(), (int *)w.as_act.p + k * u, u * sizeof(int), hipMemcpyDeviceToHost));
            HIP_CHECK(hipMemcpy(hH.data(), w.Hadd.d() + k * u * u, (size_t)u * u * D8, hipMemcpyDeviceToHost));
            printf(" | kff");
            for (int r = 0; r < u; r++) printf(" %.3e", hk[r]);
            printf(" | act");
            for (int r = 0; r < u; r++) printf(" %d", ha[r]);
            printf(" | Hdiag");
            for (int r = 0; r < u; r++) printf(" %.3e", hH[r * (u + 1)]);
          }
          printf("\n");
        }
      }
      inf.structured_solves += h.round - round;
      inf.active_set_rounds += h.round - round;
      round = h.round;
      if (h.done || round >= max_rounds) break;
      depth = 1;
    }
    if (verbose && (h.cnt[2] || h.cnt[3])) printf("pmpc_hip: active set: numerical failure / broken promise (bad %d, fail %d)\n", h.cnt[2], h.cnt[3]);
    if (!h.done) return 1;
    if (h.status != 0) return h.status;
    if (has_xb && !xbox_as) {  // the candidate's states against their boxes
      reset_scalars();
      HIP_CHECK(hipMemsetAsync(w.part_max.p, 0, 2 * PMPC_RED_BLOCKS * D8, s));
      Slab sc2 = sx;
      sc2.z = p->X_out;
      launch_violation(sc2, w.part_max.d(), s);
      exchange(c, 1);
      read_scalars(c);
      if (*c->fail_host || !(c->sc_host->viol_max <= 1e-13)) {
        if (verbose) printf("pmpc_hip: active set settled but a state box is violated by %.3e: interior-point path\n", c->sc_host->viol_max);
        w.xb_block_key = as_key;
        return 1;
      }
    }
    // the next solve's warm start: a caller inside an SCP loop (promise flag) hands the solution back as U_prev, which is then
    // the base point itself — no copy; any other caller's next warm start snaps THIS copy into its boxes
    // (sharded with several consensus stages: neither the no-rollout start nor the caller's U_prev serves — see the warm attempt
    //  below — so the copy is kept there too)
    w.as_U_valid = !(p->flags & PMPC_PREV_IS_LAST_SOLUTION) || (c->multi() && Nc > 1);
    if (w.as_U_valid) HIP_CHECK(hipMemcpyAsync(w.U.p, p->U_out, nu * D8, hipMemcpyDeviceToDevice, s));
    outputs_written = true;
    w.as_key = as_key;  // the stored set (+ w.U) start the next solve of this shape
    w.as_scale = dual_scale;
    if (mode == 0) w.as_pred_rounds = round;
    c->spec_ok = n_batches_at_hook == 1 && !(has_xb && !xbox);  // (with state boxes ignored, a second phase follows)  // what was enqueued behind the first batch saw the final outputs
    return 0;
  };
  auto active_set_solve = [&](double dual_scale, int mode, int max_rounds, int xb = 1) -> int {
    if (fast) return active_set_fast(dual_scale, mode, max_rounds, xb);
    reset_scalars();
    // generic kernels: a check pass + rollout per round, decisions on the host.  `big` never meets a normal-sized term in a sum (the penalty's target is a ZERO step), so it only has to dwarf every
    // H_uu entry: gains, H_uu^-1 and the step of a held control come out ~1e-30 relative and -big du_b is its multiplier
    const double big = 1e30, tol_p = 1e-13;
    w.as_act.ensure(nu * sizeof(int) + 8); w.as_cnt.ensure(4 * sizeof(int) + 8);
    int *act = (int *)w.as_act.p, *cnt = (int *)w.as_cnt.p;
    unsigned long long *worst_dev = (unsigned long long *)(cnt + 4);
    double *Xtry = w.dX2.d(), *Utry = w.dU2.d();  // (free here: the corrector's difference step is already applied)
    LQArgs b = a;
    b.X = Xtry; b.U = Utry; b.Dx = b.wx = nullptr; b.Du = su.D; b.wu = su.w; b.dX = w.dX.d(); b.dU = w.dU.d();
    Slab st = su;
    st.z = w.U.d(); st.dz = w.dU.d(); st.dz2 = nullptr;
    int last_add = 1, last_changes = 0x7fffffff, stalls = 0;
    w.as_key = -1;
    for (int round = 0; round < max_rounds; round++) {
      // anti-cycling on (nearly) degenerate boxes — a control at its bound with a multiplier of a few ulps flips for ever —:
      // the sign tolerance of the multipliers widens tenfold per round after the fourth, up to 1e-8 of the dual scale
      const double tol_l = dual_scale * std::min(1e-8, 1e-11 * std::pow(10.0, std::max(0, round - 3)));
      {
        // a round that only RELEASED controls keeps its base point (a released control may start from its bound): no new
        // rollout; only D changes
        const bool same_base = round > 0 && last_add == 0;
        launch_as_setup(st, round == 0 ? mode : 0, same_base, act, Utry, big, s);
        if (!same_base) launch_rollout(b, Utry, Xtry, s);
        structured_solve(c, b, true, false);
        HIP_CHECK(hipMemsetAsync(cnt, 0, 4 * sizeof(int) + 8, s));
        launch_as_check(st, act, Utry, big, tol_p, tol_l, cnt, worst_dev, s);
        launch_as_publish(cnt, (const int *)w.fail.p, c->multi() ? nullptr : c->mirror_dev->as_cnt, &c->mirror_dev->as_seq, ++c->as_seq, s);
      }
      inf.structured_solves++;
      inf.active_set_rounds++;
      if (c->multi()) {  // {released, activated, NaN, failure}: one sum for all four
        allreduce(c, cnt, 4, ncclInt32, ncclSum);
        launch_as_publish(cnt, nullptr, c->mirror_dev->as_cnt, &c->mirror_dev->as_seq, c->as_seq, s);
      }
      wait_published(c, &c->mirror->as_seq, c->as_seq);
      struct { int rel, add, bad, fail; unsigned long long worst; } hc;
      memcpy(&hc, (const void *)c->mirror->as_cnt, 4 * sizeof(int));
      hc.worst = 0;
      if (verbose) {  // (diagnostic of the check pass only)
        HIP_CHECK(hipMemcpyAsync(&hc.worst, worst_dev, 8, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
      }
      double worst;
      memcpy(&worst, &hc.worst, sizeof(double));
      if (verbose)
        printf("pmpc_hip: active set (%s) round %d: %d released, %d activated (largest %.2e)%s\n", mode == 2 ? "cold" : (mode ? "finish" : "warm"), round + 1,
               hc.rel, hc.add, worst, (hc.bad || hc.fail) ? " (numerical failure)" : "");
      if (hc.bad || hc.fail) return 2;
      last_add = hc.add;
      const int changes = hc.rel + hc.add;
      if (changes == 0) {
        if (has_xb) {  // the candidate's states against their boxes, before anything of the interior-point state is overwritten
          HIP_CHECK(hipMemsetAsync(w.part_max.p, 0, 2 * PMPC_RED_BLOCKS * D8, s));
          launch_violation_sum(sx, Xtry, w.dX.d(), w.part_max.d(), s);
          exchange(c, 1);
          read_scalars(c);
          if (*c->fail_host || !(c->sc_host->viol_max <= 1e-13)) {
            if (verbose) printf("pmpc_hip: active set settled but a state box is violated by %.3e: interior-point path\n", c->sc_host->viol_max);
            w.xb_block_key = as_key;
            return 1;
          }
        }
        launch_as_accept_all(st, act, Utry, p->U_out, Xtry, w.dX.d(), (long long)nx, w.X.d(), p->X_out, s);
        w.as_U_valid = true;  // (the acceptance pass wrote the controls into the warm-start memory)
        outputs_written = true;
        w.as_key = as_key;  // act + w.U start the next solve of this shape
        w.as_scale = dual_scale;
        return 0;
      }
      if (changes * 2 > last_changes && ++stalls >= 2) return 1;  // not contracting: leave it to the interior-point iteration
      last_changes = changes;
    }
    return 1;
  };
  if (soc) {
    if (cone_as) {
      const int cone_cold_rounds = (int)c->opt[OPT_CONE_COLD_ROUNDS];
      const bool can_defect = as_defect_on && (p->flags & PMPC_PREV_IS_LAST_SOLUTION);
      const bool prev_is_base = !c->multi() && (p->flags & PMPC_PREV_IS_LAST_SOLUTION);
      int r = 1;
      if (as_warm_on && !(p->flags & PMPC_COLD_START) && as_prev == as_key && (w.as_U_valid || can_defect || prev_is_base)) {
        r = active_set_fast(w.as_scale, 0, 14);
        if (r == 0) return finish(0);
        if (verbose) printf("pmpc_hip: warm cone rounds not settled (%d): cold start\n", r);
      }
      if (f32) return PMPC_NEEDS_F64;
      if (cone_cold_rounds > 0) {
        if (r == 2) HIP_CHECK(hipMemsetAsync(w.fail.p, 0, sizeof(int), s));
        r = active_set_fast(1.0, 3, cone_cold_rounds);
        if (r == 0) return finish(0);
        if (verbose) printf("pmpc_hip: cold cone rounds not settled (%d): path-following iteration\n", r);
      }
      HIP_CHECK(hipMemsetAsync(w.fail.p, 0, sizeof(int), s));
    }
    if (f32) return PMPC_NEEDS_F64;
    if (ncones > 0) {  // the general form has no path-following fallback
      if (verbose) printf("pmpc_hip: stage cones (general form): the rounds did not settle\n");
      return finish(1);
    }
    // the path-following iteration, then — one shared cone on the register-resident path — cone rounds started from its iterate (box
    // statuses from its duals, cone multipliers = its cone duals): they replace its last digits (5e-7 -> round-off against the cone
    // oracle, tools/debug/fuzz_soc.py) and leave set and multipliers for the next solve's warm start; if they do not settle the
    // iterate itself is the answer, as before
    if (!cone_as) return soc_interior_point(true);
    const int st_pf = soc_interior_point(false);
    if (st_pf != 0) return st_pf;  // (failed: already finished, NaN outputs)
    const pmpc_info pf = inf;
    a.Du = nullptr; a.wu = nullptr; a.du_full = 0;
    const int r5 = active_set_fast(1.0, 5, 10);
    if (r5 != 0) {
      if (r5 == 2) HIP_CHECK(hipMemsetAsync(w.fail.p, 0, sizeof(int), s));
      outputs_written = false;
    }
    inf.ipm_iters = pf.ipm_iters;
    inf.mu = r5 == 0 ? 0.0 : pf.mu;
    return finish(0);
  }
  const bool as_can_defect = as_defect_on && (p->flags & PMPC_PREV_IS_LAST_SOLUTION) && fast;
  // (the caller's U_prev is the stored set's solution; one rank only: the shared controls' base must be the same on every rank,
  //  which only this context's own copy guarantees when a caller breaks its promise)
  const bool as_prev_is_base = fast && !c->multi() && (p->flags & PMPC_PREV_IS_LAST_SOLUTION);
  // (state rows: a warm start that did not settle — hundreds of rows changing at once, see kernels_xbox.hip — would not settle for the
  //  next, similar problem either: the next 1, 2, 4 solves of the shape skip it, by the number of failures in a row)
  const bool xb_backoff = xbox_as && w.xb_warm_backoff > 0;
  if (xb_backoff) w.xb_warm_backoff--;
  if (polish_on && as_warm_on && !xb_backoff && !(p->flags & PMPC_COLD_START) && as_prev == as_key && (w.as_U_valid || as_can_defect || as_prev_is_base)) {
    a.Dx = a.wx = nullptr;
    const int r = active_set_solve(w.as_scale, 0, xbox_as ? 14 : 8);
    if (r == 0) {
      w.xb_warm_fails = 0;
      return finish(0);
    }
    if (xbox_as) {
      w.xb_warm_fails = std::min(w.xb_warm_fails + 1, 3);
      w.xb_warm_backoff = 1 << (w.xb_warm_fails - 1);
    }
    if (verbose) printf("pmpc_hip: warm active-set iteration not settled (%d): interior-point path\n", r);
    if (r == 2) HIP_CHECK(hipMemsetAsync(w.fail.p, 0, sizeof(int), s));
  }
  if (f32) return PMPC_NEEDS_F64;
  // Warm start (see below): when the previous solve of this shape ended in the interior-point phase, go there directly —
  // the equality-only solve (one factorisation + forward sweep) would only tell us that the boxes are active again; it
  // is done later if the warm attempt is rejected or fails
  const bool warm_disabled = c->opt[OPT_WARM_START] == 0.0;
  const long long key = (((((long long)x * 131 + u) * 131 + N) * 1000003 + M) * 131 + Nc) * 4 + (has_xb ? 2 : 0) + (has_ub ? 1 : 0);
  // (barrier mode, r03: the previous solve's FINAL iterate — centred at the same mu for a nearby problem — is the start: a few Newton
  //  iterations instead of ~10 from the clipped equality-only optimum)
  const bool try_warm = !warm_disabled && !(p->flags & PMPC_COLD_START) && (has_xb || has_ub) && w.warm_key == key && w.warm_mu == mu_target;
  if (!try_warm) {
    const int r = equality_phase();
    if (r != 1) return finish(r);
    // cold start of the active-set iteration: the boxes the equality-only optimum violates are the first guess (the classical
    // start of the primal-dual active-set method); the interior-point iteration below only runs if that does not settle
    const int cold_as_rounds = (int)c->opt[OPT_AS_COLD_ROUNDS];
    if (polish_on && cold_as_rounds > 0) {
      // with state boxes, in two phases: the control boxes alone first (the primal-dual active-set rule is at home there, whatever
      // the start), then the state rows from that optimum — which violates about the rows that bind, where the equality-only optimum
      // clipped into its control boxes violates many more (a start the state rows' Newton iteration does not recover from)
      // (no control boxes — e.g. a boxed slew problem in increment form —: the first phase would be the equality-only optimum again)
      const bool two_phase = xbox_as && has_ub;
      int q = active_set_solve(1.0, 2, (xbox_as && !two_phase) ? std::max(cold_as_rounds, 14) : cold_as_rounds, two_phase ? 0 : 1);
      if (q == 0 && two_phase) {
        q = active_set_fast(1.0, 4, 14, 2);
        if (q != 0) outputs_written = false;  // (the first phase's point is not the answer)
      }
      if (q == 0) return finish(0);
      if (q == 2) HIP_CHECK(hipMemsetAsync(w.fail.p, 0, sizeof(int), s));
      // (w.U still holds the equality-only optimum: the rounds work in their own buffers)
    }
  }

  // ---- 2. Mehrotra predictor-corrector on the boxes ----------------------------------------------
  // Warm start: consecutive sub-problems of an SCP / MPC loop are close, so the EARLY iterate of the previous solve of
  // this shape (first iterate with mu <= 0.5: interior, centred, far from its boxes — a late iterate jams) is a better
  // start than the clipped equality-only optimum: 11 -> 9.3 iterations at config D, 11 -> 7.1 on the unicycle.  It is
  // used only if it is strictly inside the new boxes, and a warm-started iteration that fails is repeated cold.
  // complementarity (1e-10 leaves ~3e-7 relative trajectory error on the quadrotor: too close to the 1e-6 bar).  State boxes WITHOUT the
  // state-row rounds (generic kernels, or xbox_as = 0) have nothing that finishes the iteration exactly: 1e-12 left up to 1.9e-6 on
  // slew problems with ~15 % of the state entries binding (tools/debug/fuzz_xbox.py), 1e-14 leaves 8e-8 — a breakdown on the way
  // there returns the last good iterate (see below)
  const double tol = (has_xb && !xbox_as) ? 1e-14 : 1e-12;
  const int max_iter = 80;
  // slabs as the fused per-iteration pass sees them (an unbounded slab still takes the step and feeds the
  // gradient pre-pass)
  SlabEx ex, eu;
  memset(&ex, 0, sizeof(ex));
  memset(&eu, 0, sizeof(eu));
  ex.s = sx; eu.s = su;
  if (!has_xb) { ex.s.count = (long long)nx; ex.s.d = x; ex.s.N = N; ex.s.Nc = Nc; ex.s.owner = a.owner; ex.s.z = w.X.d(); ex.s.dz = w.dX.d(); }
  if (!has_ub) { eu.s.count = (long long)nu; eu.s.d = u; eu.s.N = N; eu.s.Nc = Nc; eu.s.owner = a.owner; eu.s.is_u = 1; eu.s.z = w.U.d(); eu.s.dz = w.dU.d(); }
  ex.bounded = has_xb; eu.bounded = has_ub;
  ex.s.dz2 = w.dX2.d(); eu.s.dz2 = w.dU2.d();
  ex.pw = eu.pw = p->weights; ex.per = (long long)N * x; eu.per = (long long)N * u;
  ex.ref = p->X_ref; ex.prev = p->X_prev; ex.reg = p->reg_x; ex.gm = fast ? w.xm.d() : nullptr; ex.gd = fast ? w.xd.d() : nullptr;
  eu.ref = p->U_ref; eu.prev = p->U_prev; eu.reg = p->reg_u; eu.gm = fast ? w.um.d() : nullptr; eu.gd = fast ? w.ud.d() : nullptr;
  auto rollout = [&]() {
    if (fast) launch_rollout_fast(a, w.U.d(), w.X.d(), s);
    else launch_rollout(a, w.U.d(), w.X.d(), s);
  };
  bool remembered = false;  // this solve has stored its early iterate
  // one interior-point run from a warm (remembered iterate) or cold (clipped equality-only optimum) start; returns the
  // status (0 converged, 1 not converged, 2 numerical failure) or -1: the remembered iterate does not fit the new boxes
  auto interior_point = [&](const bool warm) -> int {
    if (warm) {
      HIP_CHECK(hipMemcpyAsync(w.U.p, w.warmU.p, nu * D8, hipMemcpyDeviceToDevice, s));
      rollout();
      if (has_xb) launch_violation(sx, w.part_max.d(), s);  // the remembered controls must be inside the NEW boxes,
      if (has_ub) launch_violation(su, w.part_max.d() + B, s);  // and so must the states they roll out to
      exchange(c, 1);
      read_scalars(c);
      if (*c->fail_host || !(c->sc_host->viol_max <= 0.0)) {
        if (verbose) printf("pmpc_hip: remembered iterate is outside the new boxes: cold start\n");
        return -1;
      }
    } else if (has_ub) {
      launch_ipm_clip(su, s);
      rollout();
    }
    a.Dx = has_xb ? sx.D : nullptr; a.wx = has_xb ? sx.w : nullptr;
    a.Du = has_ub ? su.D : nullptr; a.wu = has_ub ? su.w : nullptr;
    if (has_xb) launch_ipm_init_slack(sx, 1.0, s, warm ? 1e-9 : 1e-2);
    if (has_ub) launch_ipm_init_slack(su, 1.0, s, warm ? 1e-9 : 1e-2);
    if (warm) {  // multipliers of the remembered iterate (slacks follow from the controls and the new boxes)
      if (has_xb) {
        HIP_CHECK(hipMemcpyAsync(sx.ll, w.warm_llx.p, nx * D8, hipMemcpyDeviceToDevice, s));
        HIP_CHECK(hipMemcpyAsync(sx.lu, w.warm_lux.p, nx * D8, hipMemcpyDeviceToDevice, s));
      }
      if (has_ub) {
        HIP_CHECK(hipMemcpyAsync(su.ll, w.warm_llu.p, nu * D8, hipMemcpyDeviceToDevice, s));
        HIP_CHECK(hipMemcpyAsync(su.lu, w.warm_luu.p, nu * D8, hipMemcpyDeviceToDevice, s));
      }
    }
    int status = 1;
    double mu_peak = 1.0;  // dual scale: on badly scaled problems mu first GROWS by orders of magnitude; the
                           // complementarity tolerance is relative to that peak (1e-12 absolute is then below round-off)
    // try the active-set finish once mu <= polish_next * mu_peak (relative, like `tol`).  With state rows the iterate has to name the
    // set more sharply (measured, tools/debug/xbox_check.py: attempts at 1e-3 fail two times in three, at 1e-6 .. 1e-8 they settle)
    double polish_next = xbox_as ? 1e-3 * polish_mu : polish_mu;
    bool advanced = false;  // this iteration's elementwise pass is already in flight (launched behind the last exchange)
    double late_mu = -1.0;  // complementarity of the iterate kept in w.lateX / w.lateU (< 0: none)
    for (int it = 1; it <= max_iter; it++) {
      // previous corrector step (it > 1), predictor preparation and gradient pre-pass in ONE pass
      if (!advanced) launch_ipm_advance(ex, eu, it > 1, sc, w.part_sum.d(), w.part_cnt.d(), w.part_max.d(), s);
      advanced = false;
      if (it == 1 || mu_target > 0.0) {  // later iterates get mu / residual from the corrector's step polynomial (phase 4);
        exchange(c, 2);                    // barrier mode re-measures them together with the centrality deviation
        read_scalars(c);
      }
      const IpmScal &h = *c->sc_host;
      if (verbose)
        printf("pmpc_hip: ipm it %2d  mu %9.3e  slack_res %9.3e  nu %9.3e  alpha %6.4f  sigma %8.2e  dev %8.2e\n", it, h.mu, h.res_max,
               h.nu, h.alpha, h.sigma, h.dev_max);
      inf.mu = h.mu; inf.slack_res = h.res_max; inf.ipm_iters = it - 1;
      if (*c->fail_host || !(h.mu == h.mu)) {
        // With thousands of binding state rows the iteration can break down numerically between mu ~ 1e-12 mu_peak and the
        // convergence test at 1e-12 (slack / multiplier ratios of 1e14 in the cost-to-go; seen at config D with |v| <= 2 m/s and the
        // state-row rounds switched off: mu 7.9e-10 -> 1.3e-12 -> NaN).  The last iterate with mu <= 1e-10 mu_peak and small
        // residuals is a certified near-optimal point (duality gap <= n mu): returned instead of a failed solve.
        if (late_mu >= 0.0 && mu_target == 0.0) {
          // (said on stderr whatever `verbose` is: the status is 0, and only info.mu tells this iterate from a converged one)
          fprintf(stderr, "pmpc_hip: note: interior-point iteration broke down numerically at iteration %d; returning the kept iterate (complementarity %.3e, convergence test %.3e)\n", it, late_mu, tol * mu_peak);
          HIP_CHECK(hipMemcpyAsync(w.X.p, w.lateX.p, nx * D8, hipMemcpyDeviceToDevice, s));
          HIP_CHECK(hipMemcpyAsync(w.U.p, w.lateU.p, nu * D8, hipMemcpyDeviceToDevice, s));
          HIP_CHECK(hipMemsetAsync(w.fail.p, 0, sizeof(int), s));
          *c->fail_host = 0;
          inf.mu = late_mu;
          status = 0;
          break;
        }
        status = 2;
        break;
      }
      if (mu_target == 0.0 && h.mu <= 1e-10 * std::max(mu_peak, h.mu) && h.res_max <= 1e-10 && h.nu <= 1e-8) {
        w.lateX.ensure(nx * D8); w.lateU.ensure(nu * D8);
        HIP_CHECK(hipMemcpyAsync(w.lateX.p, w.X.p, nx * D8, hipMemcpyDeviceToDevice, s));
        HIP_CHECK(hipMemcpyAsync(w.lateU.p, w.U.p, nu * D8, hipMemcpyDeviceToDevice, s));
        late_mu = h.mu;
      }
      const bool barrier_done = mu_target > 0.0 && h.dev_max <= 1e-9 * mu_target && h.res_max <= 1e-10 && h.nu <= 1e-8;
      if (!warm_disabled && ((!remembered && mu_target == 0.0 && it > 1 && h.mu <= 0.5) || barrier_done)) {
        // (the step that produced this iterate is already applied: the pass behind the last exchange is in flight)
        w.warmU.ensure(nu * D8);
        HIP_CHECK(hipMemcpyAsync(w.warmU.p, w.U.p, nu * D8, hipMemcpyDeviceToDevice, s));
        if (has_ub) {
          w.warm_llu.ensure(nu * D8); w.warm_luu.ensure(nu * D8);
          HIP_CHECK(hipMemcpyAsync(w.warm_llu.p, su.ll, nu * D8, hipMemcpyDeviceToDevice, s));
          HIP_CHECK(hipMemcpyAsync(w.warm_luu.p, su.lu, nu * D8, hipMemcpyDeviceToDevice, s));
        }
        if (has_xb) {
          w.warm_llx.ensure(nx * D8); w.warm_lux.ensure(nx * D8);
          HIP_CHECK(hipMemcpyAsync(w.warm_llx.p, sx.ll, nx * D8, hipMemcpyDeviceToDevice, s));
          HIP_CHECK(hipMemcpyAsync(w.warm_lux.p, sx.lu, nx * D8, hipMemcpyDeviceToDevice, s));
        }
        w.warm_key = key;
        w.warm_mu = mu_target;
        remembered = true;
      }
      if (h.mu > mu_peak) mu_peak = h.mu;
      if (mu_target > 0.0) {  // centred AT mu_target: every complementarity product equals it
        if (barrier_done) { status = 0; break; }
      } else if (h.mu <= tol * mu_peak && h.res_max <= 1e-10 && h.nu <= 1e-8) {
        status = 0;
        // converged on its own (every finish attempt on the way failed, or none was due): one more attempt from the
        // converged iterate — it names the set as sharply as it ever will; a settled attempt takes the answer from ~1e-7 to round-off and
        // leaves the set and multipliers for the next solve's warm start; a failed one changes nothing (the rounds work in the outputs)
        if (polish_on && !(has_xb && !xbox_as && w.xb_block_key == as_key)) {
          const int r = active_set_solve(std::max(1.0, mu_peak), 1, xbox_as ? 10 : 6);
          if (r == 0) inf.mu = 0.0;
          else if (r == 2) HIP_CHECK(hipMemsetAsync(w.fail.p, 0, sizeof(int), s));
        }
        break;
      }
      if (it == max_iter) {
        // out of iterations between the kept iterate (mu <= 1e-10 mu_peak, small residuals) and the convergence test: that iterate
        // is a certified near-optimal point, returned as the breakdown case above returns it — not a failed solve
        if (late_mu >= 0.0 && mu_target == 0.0) {
          fprintf(stderr, "pmpc_hip: note: interior-point iteration out of iterations above its tolerance; returning the kept iterate (complementarity %.3e, convergence test %.3e)\n", late_mu, tol * mu_peak);
          HIP_CHECK(hipMemcpyAsync(w.X.p, w.lateX.p, nx * D8, hipMemcpyDeviceToDevice, s));
          HIP_CHECK(hipMemcpyAsync(w.U.p, w.lateU.p, nu * D8, hipMemcpyDeviceToDevice, s));
          inf.mu = late_mu;
          status = 0;
        }
        break;
      }
      if (polish_on && it > 1 && h.mu <= polish_next * mu_peak && !(has_xb && !xbox_as && w.xb_block_key == as_key)) {
        const double mu_now = h.mu;  // (h aliases the host snapshot)
        const int r = active_set_solve(std::max(1.0, mu_peak), 1, xbox_as ? 10 : 6);
        if (r == 0) { inf.mu = 0.0; status = 0; break; }
        // not settled: the interior-point state (U, X, slacks, multipliers) is untouched; rebuild what the attempt
        // overwrote (D, w, gradient pre-pass arrays) and go on; try again two orders of magnitude further down
        if (verbose) printf("pmpc_hip: active-set finish not settled (%d): continuing the interior-point iteration\n", r);
        polish_next = mu_now / mu_peak * (xbox_as ? 1e-4 : 1e-2);  // (a failed attempt with state rows costs up to ten rounds)
        if (r == 2) HIP_CHECK(hipMemsetAsync(w.fail.p, 0, sizeof(int), s));
        launch_ipm_advance(ex, eu, 0, sc, w.part_sum.d(), w.part_cnt.d(), w.part_max.d(), s);
      }
      // predictor (factorisation) ...
      structured_solve(c, a, true, fast, /*prep_done=*/true);
      inf.structured_solves++;
      if (has_xb) launch_ipm_step(sx, 0, sc, w.part_sum.d(), w.part_cnt.d(), s);
      if (has_ub) launch_ipm_step(su, 0, sc, w.part_sum.d() + B, w.part_cnt.d() + B, s);
      exchange(c, 3);
      // ... corrector (vector sweeps only, same factorisation; solves for the difference step)
      if (has_xb) launch_ipm_prepare(sx, 1, sc, nullptr, nullptr, nullptr, s);
      if (has_ub) launch_ipm_prepare(su, 1, sc, nullptr, nullptr, nullptr, s);
      a.dX = w.dX2.d(); a.dU = w.dU2.d();  // the sweeps never read-modify-write: step = dz + dz2
      structured_solve(c, a, false, fast);
      a.dX = w.dX.d(); a.dU = w.dU.d();
      sx.dz2 = w.dX2.d(); su.dz2 = w.dU2.d();
      if (has_xb) launch_ipm_step(sx, 1, sc, w.part_sum.d(), w.part_cnt.d(), s);
      if (has_ub) launch_ipm_step(su, 1, sc, w.part_sum.d() + B, w.part_cnt.d() + B, s);
      sx.dz2 = su.dz2 = nullptr;
      exchange(c, 4);
      if (mu_target == 0.0) {
        // phase 4 already predicts the next iterate's scalars (step polynomial) and publishes them: enqueue the next
        // elementwise pass BEHIND it before polling — it has to run whether or not that iterate turns out to be converged
        // (it applies the step), and it keeps the GPU busy while the host decides and enqueues the next factor sweep
        launch_ipm_advance(ex, eu, 1, sc, w.part_sum.d(), w.part_cnt.d(), w.part_max.d(), s);
        advanced = true;
      }
      read_scalars(c);
    }
    return status;
  };

  reset_scalars();
  int status = try_warm ? interior_point(true) : interior_point(false);
  if (try_warm && status != 0) {  // rejected or failed: fresh scalars, the equality-only optimum after all, cold start
    if (verbose && status > 0) printf("pmpc_hip: warm-started iteration failed (status %d): repeating from a cold start\n", status);
    w.warm_key = -1;
    remembered = false;
    HIP_CHECK(hipMemsetAsync(w.fail.p, 0, sizeof(int), s));
    launch_ipm_exchange(0, false, false, sc, (const int *)w.fail.p, w.xch.d(), c->rank, c->world, nullptr, nullptr, nullptr, 0, s,
                        mu_target, w.part_dev.d());
    const int r = equality_phase();
    if (r != 1) return finish(r);
    status = interior_point(false);
  }
  if (verbose && status != 0) printf("pmpc_hip: interior-point iteration did not converge (status %d)\n", status);
  return finish(status);
}

// -------------------------------------------------------------------------------------------------
// SCP loop with the host out of the loop body (built-in dynamics)
// -------------------------------------------------------------------------------------------------
int pmpc_scp_loop_device(pmpc_ctx *c, int model, const double *params, const pmpc_problem *p0, double *f2, double *fx2, double *fu2,
                         int steps, int first_cold, double *res, pmpc_info *infos, int *last_in_out) {
  pmpc_problem p = *p0;
  // trajectory buffers A = (X_prev, U_prev), B = (X_out, U_out); linearisation buffers 0 = (f, fx, fu), 1 = (f2, fx2, fu2)
  double *XA = const_cast<double *>(p0->X_prev), *UA = const_cast<double *>(p0->U_prev), *XB = p0->X_out, *UB = p0->U_out;
  double *F[2][3] = {{const_cast<double *>(p0->f), const_cast<double *>(p0->fx), const_cast<double *>(p0->fu)}, {f2, fx2, fu2}};
  const bool soc = p0->soc_u_interior != nullptr || p0->cone_count > 0;
  const bool cone_obj = (p0->flags & PMPC_CONE_OBJECTIVE) != 0;  // the sub-problem is the reference's default path (c_lcone_solve)
  const int jac32 = (p0->flags & PMPC_F32_MATRICES) ? 1 : 0;  // (f / fx / fu scratch sets: fx, fu FLOAT arrays then)
  int done = 0, cur = 0;
  bool lin_ready = false;  // the linearisation of iteration `done` is already enqueued (valid speculation of the previous one)
  try {
    HIP_CHECK(hipSetDevice(c->device));
    HIP_CHECK(hipMemsetAsync(res, 0, (size_t)steps * sizeof(double), c->stream));  // (the residual kernel takes a maximum into its slot)
    for (; done < steps; done++) {
      double *Xp = (done & 1) ? XB : XA, *Up = (done & 1) ? UB : UA, *Xo = (done & 1) ? XA : XB, *Uo = (done & 1) ? UA : UB;
      if (!lin_ready) {
        ProfScope ps(c, 6);
        launch_linearize(model, (int)p.N, (int)p.M, p.x0, Xp, Up, params, F[cur][0], F[cur][1], F[cur][2], c->stream, jac32);
      }
      p.f = F[cur][0]; p.fx = F[cur][1]; p.fu = F[cur][2];
      p.X_prev = Xp; p.U_prev = Up; p.X_out = Xo; p.U_out = Uo;
      p.flags = p0->flags | PMPC_STATIC_CONS_BOUNDS;
      if (done > 0 || !first_cold) p.flags |= PMPC_PREV_IS_LAST_SOLUTION;
      else p.flags &= ~(unsigned)PMPC_PREV_IS_LAST_SOLUTION;
      bool res_dirty = false;  // the slot was zeroed with all the others when the loop started; a repeat must zero it again
      auto follow_up = [&, Xp, Up, Xo, Uo](bool with_next_lin) {  // residual of this iteration (+ the next linearisation)
        const bool next = with_next_lin && done + 1 < steps;
        if (next && !res_dirty && !c->multi()) {  // both in ONE launch (independent work)
          ProfScope ps(c, 6);
          launch_linearize_with_residual(model, (int)p.N, (int)p.M, p.x0, Xo, Uo, params, F[cur ^ 1][0], F[cur ^ 1][1], F[cur ^ 1][2], Xo, Xp, Uo,
                                         Up, (int)p.xdim, (int)p.udim, res + done, c->stream, jac32);
          res_dirty = true;
          return;
        }
        {
          ProfScope ps(c, 7);
          launch_scp_residual(Xo, Xp, Uo, Up, (long long)p.M * (long long)p.N, (int)p.xdim, (int)p.udim, res + done, c->stream, res_dirty);
          res_dirty = true;
        }
        if (c->multi()) allreduce(c, res + done, 1, ncclFloat64, ncclMax);
        if (next) {
          ProfScope ps(c, 6);
          launch_linearize(model, (int)p.N, (int)p.M, p.x0, Xo, Uo, params, F[cur ^ 1][0], F[cur ^ 1][1], F[cur ^ 1][2], c->stream, jac32);
        }
      };
      c->spec_fired = c->spec_ok = false;
      c->post_batch = [&]() { follow_up(true); };
      pmpc_info inf;
      const int st = cone_obj ? lcone_body(c, &p, p.barrier_mu > 0.0 ? 1.0 / p.barrier_mu : std::numeric_limits<double>::quiet_NaN(), &inf, 0)
                              : solve_impl(c, &p, &inf, 0, soc);
      c->post_batch = nullptr;
      if (infos) infos[done] = inf;
      if (st != 0) break;
      if (c->spec_fired && c->spec_ok) {
        lin_ready = true;  // residual and next linearisation are in flight behind the accepted rounds
      } else {
        follow_up(false);  // (a speculative copy, if any, was computed from unfinished outputs: redone; the next linearisation
        lin_ready = false;  //  is enqueued at the top of the next iteration, into the other buffer set)
      }
      cur ^= 1;
    }
    HIP_CHECK(hipGetLastError());
  } catch (const PmpcHipError &) {
    c->post_batch = nullptr;
    if (infos && done < steps) { memset(&infos[done], 0, sizeof(pmpc_info)); infos[done].status = 2; }
    fail_after_error(c, nullptr, nullptr);
    // both trajectory pairs hold unfinished iterates now: NaN, as every single-solve entry does with its outputs (res[done..] is
    // undefined); never throws
    try {
      const double nan = std::numeric_limits<double>::quiet_NaN();
      const long long ex = (long long)p.M * p.N * p.xdim, eu = (long long)p.M * p.N * p.udim;
      for (double *b : {XA, XB}) if (b) launch_fill(b, nan, ex, c->stream);
      for (double *b : {UA, UB}) if (b) launch_fill(b, nan, eu, c->stream);
      HIP_WARN(hipStreamSynchronize(c->stream));
    } catch (...) {
    }
  }
  if (last_in_out) *last_in_out = done & 1;
  return done;
}

// -------------------------------------------------------------------------------------------------
// cone path (c_lcone_solve): the epsilon-anchored epigraph objective as a sequence of weighted QPs
// -------------------------------------------------------------------------------------------------
}  // extern "C"
