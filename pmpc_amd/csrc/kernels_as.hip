// kernels_as.hip — the sweeps of the primal-dual active-set rounds on the control boxes (gfx950 / CDNA4), register-resident
// MFMA layout of kernels_fast.hip (fast_common.h).  These are the kernels bench.py's SCP loop spends its time in.
//
// What a round is (solver.hip, active_set_solve): with a guess of the active set, ONE structured solve from a base point whose
// held controls sit exactly on their bounds is the exact optimum on that set; the forward sweep tests the KKT signs as it goes,
// clamps / releases, propagates the clamped step and leaves base + step as the next base point.  Differences from the sweeps of
// kernels_fast.hip, all of them about memory passes and launches that the rounds do not need:
//   * no gradient pre-pass: the backward sweep forms  x - x_ref,  reg_x (x - x_prev),  u - u_ref,  reg_u (u - u_prev)  and the
//     penalty diagonal of the held controls itself, from the base point (a.Xb, a.Ub), the references and the status array;
//   * the forward sweep reads the base point and writes the NEW base point (absolute states and controls, a.Xo / a.Uo — the
//     caller's output buffers), so an accepted round needs no copy and a continued round no "base += step" pass;
//   * DEFECT (first round of a warm start inside an SCP loop): the base point is the linearisation point (X_prev, U_prev)
//     itself, whose dynamics defect f - X_prev is elementwise and rides through the sweeps (backward s += S r, forward
//     dx += r): no rollout, and nothing of the base point has to be written before the sweep;
//   * SKIP (later rounds): particles whose set did not change keep their factors; the wave only refreshes its condensed
//     consensus gradient, g_i += H_i delta, and leaves;
//   * every kernel returns at once when *a.done is set: the host enqueues several rounds ahead without reading anything
//     back, the device decides (k_as_ctl) whether the later ones still have work.
//
// Reference semantics: the QP of PMPC.jl/src/lqp_utils.jl:2-393 restricted to an active set (same Newton system as kernels_fast.hip).
#include <type_traits>

#include "fast_common.h"
#include "as_ctl_dev.h"

// Diagnostic build only (-DPMPC_STAGE_TIMELINE, tools/micro/stage_timeline.py; never in the shipped library): s_memtime stamps at the
// phase boundaries of every stage of ONE wave (the block in the middle of the grid) of the two sweeps, written to a buffer of their
// own that no kernel reads.  Scheduling barriers pin each stamp between the phases, so the instrumented stage is not the shipped
// schedule: the stamps say where the time of a stage goes, not what the shipped stage costs to the cycle.
#ifdef PMPC_STAGE_TIMELINE
#define PMPC_TL_STAMPS 10
__device__ unsigned long long pmpc_tl[3][128][PMPC_TL_STAMPS];  // [0 full factor sweep, 1 (unused), 2 forward sweep][stage][stamp]
#define TL_DECL(kind) unsigned long long tl_[PMPC_TL_STAMPS] = {}; const int tl_kind = (kind); const bool tl_on = blockIdx.x == gridDim.x / 2
#define TL(k) do { __builtin_amdgcn_sched_barrier(0); tl_[k] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define TL_FLUSH(j) do { if (tl_on && threadIdx.x == 0 && (j) < 128) { for (int k_ = 0; k_ < PMPC_TL_STAMPS; k_++) pmpc_tl[tl_kind][j][k_] = tl_[k_]; } } while (0)
#else
#define TL_DECL(kind)
#define TL(k)
#define TL_FLUSH(j)
#endif

#ifndef PMPC_AS_LEAN_WAVES
#define PMPC_AS_LEAN_WAVES 4  // occupancy the lean backward sweep is compiled for (A/B builds: -DPMPC_AS_LEAN_WAVES=1 lifts the cap)
#endif
#ifndef PMPC_AS_DEEP_WAVES
#define PMPC_AS_DEEP_WAVES 3  // occupancy of the deep-pipeline variant
#endif
#ifndef PMPC_AS_DEEP2_MAXM
#define PMPC_AS_DEEP2_MAXM 1024  // particles per GPU up to which the two-stages-ahead variant runs full sweeps (the partial sweeps of
                                 // the later rounds always use it); 0: never.  Since the merged loads (13 memory instructions per
                                 // stage) its three register sets fit: 166 - 185 registers, no scratch
#endif
#ifndef PMPC_AS_PINGPONG
#define PMPC_AS_PINGPONG 0    // main loop: two stages per trip, the prefetch register sets swap roles (0: one stage + rotation moves)
#endif
#ifndef PMPC_AS_DEFECT_G
#define PMPC_AS_DEFECT_G 0    // DEFECT sweeps: the defect term of the gradient as G'r from the product G = S F the stage forms anyway
                              // (3 multiply-adds on a row-distributed copy of r) instead of S r by 3 row sums over DPP moves (39 instructions)
#endif

namespace {

// ------------------------------------------------------------------------------------------------
// backward (factor) sweep of an active-set round
// ------------------------------------------------------------------------------------------------
// Loop structure: the free stages j >= 1 (the bulk of the horizon) run in a branch-free MAIN body, two stages per loop trip
// with the two prefetch register sets swapping roles (no rotation moves); the consensus stages and stage 0 (no incoming
// state) go through the general body afterwards.
// MODE — how far ahead a stage's data is requested:
//   0 lean   early data (F, R, control word, f) one stage ahead, mid / late data (Q, base state below) behind the Cholesky phase of the
//            stage above: <= 128 registers, 4 waves per SIMD (more than 3072 particles per GPU: the waves hide each other's latency)
//   1 deep   everything one full stage ahead (3 waves per SIMD)
//   2 deep2  everything TWO stages ahead (2 waves per SIMD): with at most two waves per SIMD — small shards, the few unsettled
//            particles of the later rounds — a stage is shorter than the HBM latency, and data requested one stage ahead would
//            pin every stage to that latency
// CONE: stage cones ride along (kernels_cone.hip prepares their Newton terms per round): a full (u x u) block cone_H added to H_uu
// — loaded like R — and a vector cone_g added to the control gradient — it comes in with the control word (lane XP + 3 of the
// control quad), so the sweep pays ONE more load per stage.  Consensus stages: the owner's particle 0 alone adds them.
// MT: storage type of the matrix stacks fx, fu, Q, R and of the factor record (float = the fp32-storage mode; arithmetic stays fp64).
// EX = 2, XBOX: state boxes ride along (kernels_xbox.hip prepares their terms per round): a penalty xb_D on the diagonal of the state
// cost of the stage below and xb_g in its gradient — two more loads per stage, on the state columns, next to the base state's.
template <int XD, int UD, int MODE, bool SKIP, bool DEFECT, int EX = 0, class MT = double>
__global__ void __launch_bounds__(64, MODE == 2 ? 2 : (MODE == 1 ? PMPC_AS_DEEP_WAVES : PMPC_AS_LEAN_WAVES)) k_bwd_as(LQArgs a) {
  constexpr bool CONE = EX == 1, XBOX = EX == 2;
  typedef Lane<XD, UD> LT;
  constexpr int KS = LT::KS, XP = LT::XP;
  constexpr bool PADX = (XD != XP);
  constexpr long long D8 = sizeof(double);
  constexpr bool DEEP = MODE != 0, SCHOL = !DEEP;
  constexpr long long MB = sizeof(MT);  // bytes per matrix entry
  if (a.done && *a.done) return;
  const int lane = threadIdx.x;
  // (as_perm: the round's unsettled particles first — their long sweeps then land one per SIMD instead of where their indices put them)
  const int N = a.N, Nc = a.Nc, i = a.as_perm ? __builtin_amdgcn_readfirstlane(a.as_perm[blockIdx.x]) : (int)blockIdx.x;
  if (SKIP && a.as_settled_in[i]) {
    // nothing of this particle changed: factors, condensed Hessian H_i and conditional optimum stand; its reduced consensus
    // gradient follows the consensus step that was applied, g_i += H_i delta (exact: the QP is quadratic).  nc <= 32 here.
    const int nc = Nc * UD;
    if (lane < nc) {
      const double *H = a.Hc_part + (size_t)i * nc * nc;  // symmetric; off-diagonal blocks live in the upper triangle
      double acc = 0.0;
      for (int cc = 0; cc < nc; cc++) acc = fma(H[(lane < cc ? lane : cc) + nc * (lane < cc ? cc : lane)], a.as_delta[cc], acc);
      a.gc_part[(size_t)i * nc + lane] += acc;
    }
    return;
  }
  const LT L(lane);
  const int g = L.g, c = L.c;
  const size_t pbase = (size_t)i * N;
  const bool own0 = (i == 0 && a.owner);
  const bool gu = g < UD;
  const double *Z = a.zeros;
  const double pwt = a.pw ? a.pw[i] : 1.0;
  const double regx = pwt * a.reg_x, regu = pwt * a.reg_u;
  // base point of the round
  const double *Xb = DEFECT ? a.X_prev : a.Xb, *Ub = DEFECT ? a.U_prev : a.Ub;
  // Checkpointed restart (SKIP sweeps): nothing above the highest stage whose status (or Newton term) changed is different from
  // the last sweep of this particle — the sweep starts at the lowest checkpoint at or above it, jtop = PMPC_AS_CK_FIRST << k, from the
  // recorded cost-to-go; jtop = N - 1: the whole horizon from the terminal cost.  Checkpoints sit on free stages only.
  constexpr int CKS = 64 * KS + 32;  // doubles per checkpoint: the S tile as the lanes hold it, s and the base state by state index
  int jtop = N - 1, ck_k = 0;  // ck_k: slot + 1 of the checkpoint the sweep starts from (0: from the terminal cost)
  if (SKIP && a.as_ck) {
    const int jh = a.as_jhi[i];
    const int lo = jh > Nc ? jh : (Nc > 1 ? Nc : 1);
    const int k = lo <= PMPC_AS_CK_FIRST ? 0 : 32 - __builtin_clz((unsigned)(lo - 1) >> PMPC_AS_CK_LOG);  // lowest k with FIRST << k >= lo
    if (k < a.ck_slots) { ck_k = __builtin_amdgcn_readfirstlane(k + 1); jtop = PMPC_AS_CK_FIRST << (ck_k - 1); }
    if (lane == 0 && a.ck_stat) {
      atomicAdd(a.ck_stat + (ck_k ? 0 : 2), 1ull);
      atomicAdd(a.ck_stat + (ck_k ? 1 : 3), (unsigned long long)(jtop + 1));
    }
  }
  // (null when off: the stages test this pointer alone — the sweeps run at the limit of the scalar register file)
  const double *ck_ = a.as_ck ? ubase(a.as_ck, (long long)i * a.ck_slots * (long long)(CKS * sizeof(double))) : nullptr;

  // F = [fx | fu]: per-lane pointer at stage jtop and per-lane byte stride (0 for lanes that read the zero buffer)
  const bool fF = L.cxv || L.cu;
  const MT *pF = L.cxv ? (const MT *)a.fx + (pbase + jtop) * (XD * XD) + XD * L.oc + L.row0
                       : (L.cu ? (const MT *)a.fu + (pbase + jtop) * (XD * UD) + XD * L.cb + L.row0 : (const MT *)Z);
  const int sF = fF ? -(int)MB * (L.cxv ? XD * XD : XD * UD) : 0;
  // everything else: UNIFORM stage base (scalar registers, advanced by the scalar unit) + a per-lane constant byte offset.
  // Lanes without an entry read entry 0 of the stage block (finite data) and are masked by a zero factor or a select.
  const unsigned lQ = (unsigned)((L.cxv ? XD * L.oc + L.row0 : 0) * MB);
  const double pwt_x = L.cxv ? pwt : 0.0;  // cost weight on the state columns, zero elsewhere (masks Q)
  const bool fR = L.cu && gu;
  const unsigned lR = (unsigned)((fR ? g + UD * L.cb : 0) * MB), lRd = (unsigned)((fR ? g + UD * L.cb : 0) * D8);
  const unsigned lxr = (unsigned)(L.row0 * D8), lxc = (unsigned)((L.cxv ? L.oc : 0) * D8);
  const unsigned lug = (unsigned)((gu ? g : 0) * D8);
  const double regx_c = L.cxv ? regx : 0.0;
  const bool umask = L.cu && g == L.cb;
  const double umask_d = umask ? 1.0 : 0.0;
  const bool diag_x = L.cxv && ((c & 3) == g);
  bool dmask[KS];
#pragma unroll
  for (int r = 0; r < KS; r++) dmask[r] = diag_x && (c >> 2) == r;
  // particle-local bases once (scalar registers); per stage only `stage index * block size` is added on the scalar unit
  const long long px = (long long)(pbase * XD) * D8, pu = (long long)(pbase * UD) * D8;
  const double *Xb_ = ubase(Xb, px), *Xr_ = ubase(a.X_ref, px), *Xp_ = ubase(a.X_prev, px), *f_ = ubase(a.f, px);
  const double *kff_ = ubase(a.kff, pu);
  const void *Q_ = ubase_v(a.Q, (long long)(pbase * (XD * XD)) * MB), *R_ = ubase_v(a.R, (long long)(pbase * (UD * UD)) * MB);
  const void *K_ = ubase_v(a.K, (long long)(pbase * 64) * MB);
  auto xoff = [&](int jj) { return (long long)(jj * (int)(XD * D8)); };
  auto uoff = [&](int jj) { return (long long)(jj * (int)(UD * D8)); };
  auto ld_Q = [&](int jj, double *dst) {  // Q_jj, rows g + 4r of column c (garbage on the control columns: masked by pwt_x)
    const void *q = ubase_v(Q_, (long long)(jj * (int)(XD * XD * MB)));
#pragma unroll
    for (int r = 0; r < KS; r++) {
      const bool rv = !PADX || (L.row0 + r < XD);
      const double v = ldom<MT>(q, rv ? lQ + r * (unsigned)MB : 0u);
      dst[r] = rv ? v : 0.0;
    }
  };
  auto ld_xm = [&](int jj, double *dst) {  // pw (x - x_ref) of stage jj, row-distributed
    const double *xb = ubase(Xb_, xoff(jj)), *xr = ubase(Xr_, xoff(jj));
#pragma unroll
    for (int r = 0; r < KS; r++) {
      const bool rv = !PADX || (L.row0 + r < XD);
      const unsigned o = rv ? lxr + r * 8u : 0u;
      const double v = pwt * (ldo(xb, o) - ldo(xr, o));
      dst[r] = rv ? v : 0.0;
    }
  };
  auto ld_gx = [&](int jj) -> double {  // pw reg_x (x - x_prev) on the state columns (the defect base IS x_prev)
    if (DEFECT) return 0.0;
    return regx_c * (ldo(ubase(Xb_, xoff(jj)), lxc) - ldo(ubase(Xp_, xoff(jj)), lxc));
  };
  // Per-control inputs of a stage in ONE load (the sweep's memory instructions are a cost of their own, see k_fwd_as): through a
  // per-lane pointer the control quad of k-group g (lanes XP .. XP + 3) reads {u base, u_ref, status}[g] in its lanes 0, 1, 2,
  // and lane 15 - XP - b of k-groups 0 / 1 reads u base[b] / u_prev[b] (a row mirror takes them to the control column XP + b).
  const char *pC = (const char *)Z;
  int sC = 0;
  if (gu && c == XP) { pC = (const char *)(Ub + (pbase + jtop) * UD + g); sC = -UD * (int)D8; }
  else if (gu && c == XP + 1) { pC = (const char *)(a.U_ref + (pbase + jtop) * UD + g); sC = -UD * (int)D8; }
  else if (gu && c == XP + 2) { pC = (const char *)(a.as_act + (pbase + jtop) * UD + g); sC = -UD * (int)sizeof(int); }  // (read as 8 bytes: spare bytes behind the buffer)
  else if (CONE && gu && c == XP + 3) { pC = (const char *)(a.cone_g + (pbase + jtop) * UD + g); sC = -UD * (int)D8; }
  else if (!DEFECT && g < 2 && 15 - XP - c >= 0 && 15 - XP - c < UD) {
    pC = (const char *)((g == 0 ? Ub : a.U_prev) + (pbase + jtop) * UD + (15 - XP - c));
    sC = -UD * (int)D8;
  }
  const double regu_s = (L.cu && g < 2) ? (g == 0 ? regu : -regu) : 0.0;  // reg_u (u - u_prev): the two terms summed over the k-groups
  auto ld_R = [&](int jj) -> double {
    const double v = ldom<MT>(ubase_v(R_, (long long)(jj * (int)(UD * UD * MB))), lR);
    return fR ? v : 0.0;
  };
  const double *CH_ = CONE ? ubase(a.cone_H, (long long)(pbase * (UD * UD)) * D8) : Z;
  auto ld_CH = [&](int jj) -> double {  // cone block entry [g][cb], laid out like R
    const double v = ldo(ubase(CH_, (long long)(jj * (int)(UD * UD * D8))), lRd);
    return fR ? v : 0.0;
  };
  const bool frec = (L.cxv || L.cu) && gu;
  const unsigned lrec = (unsigned)(lane * MB);
  auto st_rec = [&](int jj, double v) {  // this lane's slot of the stage's factor record: one coalesced store (512 bytes in fp64)
    gstom<MT>(ubase_v(K_, (long long)(jj * (int)(64 * MB))), lrec, v);
  };

  // prefetch register set of one stage: what it needs the moment it starts (F, R, control word, f of the stage) and — DEEP —
  // its mid / late data (Q and the base point of the stage BELOW it) as well
  struct PipeCone { double ch; };
  struct PipeXbox { double xd, xg; };
  struct PipeNone {};
  struct Pipe : std::conditional_t<CONE, PipeCone, std::conditional_t<XBOX, PipeXbox, PipeNone>> { double F[KS], R, ctl, f, Q[KS], xb, xr, xp; };
  const double *XD_ = XBOX ? ubase(a.xb_D, px) : Z, *XG_ = XBOX ? ubase(a.xb_g, px) : Z;
  int jF = jtop;  // stage pF / pC point at
  auto fetch_early = [&](int jj, Pipe &q) {  // called in descending stage order (a clamped repeat of stage 0 leaves the pointers alone)
    if (jj < jF) { pF = (const MT *)((const char *)pF + sF); pC += sC; jF = jj; }
#pragma unroll
    for (int r = 0; r < KS; r++) q.F[r] = (!PADX || L.row0 + r < XD || (const void *)pF == (const void *)Z) ? gldm<MT>(pF + r) : 0.0;
    q.R = ld_R(jj);
    q.ctl = gld(pC);
    if (DEFECT) q.f = ldo(ubase(f_, xoff(jj)), lxc);
    if constexpr (CONE) q.ch = ld_CH(jj);
  };
  // Q and the base point of stage jbelow (= jj - 1, clamped at 0) on the state COLUMNS: one load per array; the row-distributed
  // copy the Q x product needs is made by lane shuffles when the stage consumes it
  auto fetch_late = [&](int jj, int jbelow, Pipe &q) {
    ld_Q(jbelow, q.Q);
    q.xb = ldo(ubase(Xb_, xoff(jbelow)), lxc);
    q.xr = ldo(ubase(Xr_, xoff(jbelow)), lxc);
    if (!DEFECT) q.xp = ldo(ubase(Xp_, xoff(jbelow)), lxc);
    if constexpr (XBOX) {
      q.xd = ldo(ubase(XD_, xoff(jbelow)), lxc);
      q.xg = ldo(ubase(XG_, xoff(jbelow)), lxc);
    }
  };

  double S[KS], s_row[KS], s_col;
  bool bad = false;  // a pivot of some Huu was not positive
  TL_DECL(SKIP ? 1 : 0);
  if (SKIP && ck_k > 0) {
    // ---- restart: (S, s) of the checkpoint at the top of stage jtop, s moved to today's base state: s + S (x_now - x_then) ----
    const double *ck = ubase(ck_, (long long)(ck_k - 1) * (long long)(CKS * D8));
#pragma unroll
    for (int r = 0; r < KS; r++) S[r] = ldo(ck, (unsigned)((lane + 64 * r) * D8));
    const double s_ck = ldo(ck, (unsigned)(64 * KS * D8) + lxc), x_ck = ldo(ck, (unsigned)((64 * KS + 16) * D8) + lxc);  // (by ORIGINAL state index)
    const double dx_col = L.cxv ? ldo(ubase(Xb_, xoff(jtop)), lxc) - x_ck : 0.0;
    double dx_row[KS], part = 0.0;
    col_to_row<KS>(dx_col, g, dx_row);
#pragma unroll
    for (int r = 0; r < KS; r++) part = fma(S[r], dx_row[r], part);
    part = grp_allsum(part);
    s_col = L.cxv ? s_ck + part : 0.0;
    col_to_row<KS>(s_col, g, s_row);
  } else
  // ---- terminal: S = Q~_{N-1}, s = g_x,N-1 ---------------------------------------------------------------------
  {
    double Q0[KS], xm0[KS], part = 0.0;
    ld_Q(N - 1, Q0);
    ld_xm(N - 1, xm0);
#pragma unroll
    for (int r = 0; r < KS; r++) {
      part = fma(Q0[r], xm0[r], part);
      S[r] = fma(pwt_x, Q0[r], dmask[r] ? regx : 0.0);
    }
    part = grp_allsum(part);
    if constexpr (XBOX) {
      const double d0 = ldo(ubase(XD_, xoff(N - 1)), lxc);
      part += ldo(ubase(XG_, xoff(N - 1)), lxc);
#pragma unroll
      for (int r = 0; r < KS; r++) S[r] += dmask[r] ? d0 : 0.0;
    }
    s_col = L.cxv ? part + ld_gx(N - 1) : 0.0;
    col_to_row<KS>(s_col, g, s_row);
  }
  // DEFECT: x_prev of the stage being processed = the base state the stage above loaded as ITS stage below
  double xb_carry = DEFECT ? ldo(ubase(Xb_, xoff(N - 1)), lxc) : 0.0;

  // one stage.  MAIN: a free stage with j >= 1 (no branches).  Returns nothing; the caller stops after stage 0.
  auto stage = [&](auto main_tag, const int j, const Pipe &cur, Pipe &nxt) {
    constexpr bool MAIN = decltype(main_tag)::value;
    TL(0);  // stage entry
    // checkpoint at the top of a free stage j = FIRST << k: the cost-to-go as it stands and the base state it is expanded around
    // (ONE uniform branch per stage, ahead of everything else; DEFECT sweeps carry the base state of the stage anyway, the others
    // — cold starts and restarts from the second rung up — fetch it again: it was read one stage ago)
    if (MAIN && ck_ && (j & (j - 1)) == 0 && j >= PMPC_AS_CK_FIRST) {
      const double *ck = ubase(ck_, (long long)(__builtin_ctz((unsigned)j) - PMPC_AS_CK_LOG) * (long long)(CKS * D8));
#pragma unroll
      for (int r = 0; r < KS; r++) gsto(ck, (unsigned)((lane + 64 * r) * D8), S[r]);
      const double xj = DEFECT ? xb_carry : ldo(ubase(Xb_, xoff(j)), lxc);
      if (g == 0 && L.cxv) {
        gsto(ck, (unsigned)(64 * KS * D8) + lxc, s_col);
        gsto(ck, (unsigned)((64 * KS + 16) * D8) + lxc, xj);
      }
    }
    const bool cons = MAIN ? false : j < Nc;
    const bool below = MAIN ? true : j > 0;  // a stage j - 1 exists
    double Fr[KS], Qc[KS], xm_row[KS], gx_c, Du_c;
#pragma unroll
    for (int r = 0; r < KS; r++) Fr[r] = (!MAIN && j == 0 && L.cxv) ? 0.0 : cur.F[r];  // stage 0 has no incoming state: A~_0 = 0
    const double Rc = cur.R;
    // control word: quad broadcasts inside the control quad, row mirror for the column terms
    const double um_g = pwt * (dpp_d<0x00>(cur.ctl) - dpp_d<0x55>(cur.ctl));  // pw (u - u_ref)[g] (R is zero outside the control quad)
    const int act_g = __builtin_amdgcn_mov_dpp(__double2loint(cur.ctl), 0xAA, 0xF, 0xF, true);
    Du_c = (umask && act_g) ? a.as_big : 0.0;  // penalty of a held control, on the diagonal lanes (XP + b, b)
    const double gu_c0 = DEFECT ? 0.0 : regu_s * dpp_d<0x140>(cur.ctl);  // +reg_u u[b] in k-group 0, -reg_u u_prev[b] in k-group 1
    const bool cone_here = CONE && (MAIN || !cons || own0);
    // cone_g[g] sits in lane XP + 3 of the control quad; lane (XP + g, g) adds it.  The broadcast must run with ALL lanes enabled (a
    // DPP move reads 0 from a disabled lane): masked by a multiplication, not by a select the compiler could turn into a branch
    // around the move (it did: the gradient term was silently dropped)
    double gu_c = gu_c0;
    if constexpr (CONE) gu_c = fma(dpp_d<0xFF>(cur.ctl), cone_here ? umask_d : 0.0, gu_c0);
    double Hcone = 0.0;
    if constexpr (CONE) Hcone = cone_here ? cur.ch : 0.0;
    const double df_c = (DEFECT && L.cxv) ? cur.f - xb_carry : 0.0;     // dynamics defect of the base point on the state columns
    if (DEFECT) xb_carry = cur.xb;
    gx_c = DEFECT ? 0.0 : regx_c * (cur.xb - cur.xp);
    if constexpr (XBOX) gx_c += cur.xg;
    col_to_row<KS>(pwt_x * (cur.xb - cur.xr), g, xm_row);
#pragma unroll
    for (int r = 0; r < KS; r++) Qc[r] = cur.Q[r];
    // mid / late data of the stage below: DEEP issues it now (a full stage ahead: lowest latency, most registers), the lean
    // variant behind the Cholesky phase of this stage (its registers are free again by then; the loads still have the rest of
    // this stage and the head of the next to land — with 4 waves per SIMD interleaved that covers the memory latency)
    auto late_pf = [&]() { if (below) fetch_late(j - 1, j >= 2 ? j - 2 : 0, nxt); };
    TL(1);  // control word decoded, this stage's operands in place (waited for the loads of this stage)
    if (MODE == 1) late_pf();
    if (MODE != 2 && below) fetch_early(j - 1, nxt);  // (deep2: the caller's ring has requested stage j - 2 already)
    TL(2);  // next stage's loads issued

    double d_row[KS];
    if (DEFECT) {  // x_j = F [x_{j-1}; u_j] + r_j: the cost-to-go gradient seen through the stage is s + S r
      if (PMPC_AS_DEFECT_G) {
        col_to_row<KS>(df_c, g, d_row);  // h = F'(s + S r) = F's + G'r with G = S F: the S r product is never formed
      } else {
#pragma unroll
        for (int r = 0; r < KS; r++) s_row[r] += row_allsum(S[r] * df_c);
      }
    }
    // ---- h = F' s (+ control gradient) -----------------------------------------------------------------
    double hp = fma(Rc, um_g, gu_c);
#pragma unroll
    for (int r = 0; r < KS; r++) hp = fma(Fr[r], s_row[r], hp);
    double h_col = 0.0;
    double hu[UD];
    if (!(DEFECT && PMPC_AS_DEFECT_G)) {
      h_col = grp_allsum(hp);
#pragma unroll
      for (int b = 0; b < UD; b++) hu[b] = readlane_d(h_col, XP + b);
    }

    TL(3);  // gradient h = F' (s + S r) formed, control rows read out
    // ---- H = F' S F + blkdiag(Q~_{j-1}, R~_j) ------------------------------------------------------
    v4d H = {0.0, 0.0, 0.0, 0.0};
    double p2q = 0.0;  // Q_{j-1} xm_{j-1} part of s_{j-1}, formed now: Qc / xm_row are dead before the Cholesky phase (registers)
    if (below) {
#pragma unroll
      for (int r = 0; r < KS; r++) {
        if constexpr (XBOX) H[r] = fma(pwt_x, Qc[r], dmask[r] ? regx + cur.xd : 0.0);
        else H[r] = fma(pwt_x, Qc[r], dmask[r] ? regx : 0.0);
        p2q = fma(Qc[r], xm_row[r], p2q);
      }
    }
    if constexpr (CONE) H[KS] = fma(pwt, Rc, (umask ? regu : 0.0) + (cons ? 0.0 : Du_c + Hcone));
    else H[KS] = fma(pwt, Rc, (umask ? regu : 0.0) + (cons ? 0.0 : Du_c));
    v4d G = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < KS; r++) G = mfma(S[r], Fr[r], G);
#pragma unroll
    for (int r = 0; r < KS; r++) H = mfma(Fr[r], G[r], H);
    if (DEFECT && PMPC_AS_DEFECT_G) {
#pragma unroll
      for (int r = 0; r < KS; r++) hp = fma(G[r], d_row[r], hp);
      h_col = grp_allsum(hp);
#pragma unroll
      for (int b = 0; b < UD; b++) hu[b] = readlane_d(h_col, XP + b);
    }

    if (!DEEP && !MAIN) late_pf();
    if (!MAIN && cons) {
      // consensus stage: no minimisation.  Export the condensed gradient block and the diagonal Hessian block Huu; keep
      // Y_j = H_ux in the factor record (k_cond_fast) and carry P_{j-1} = H_xx on as the cost-to-go
      double v = H[KS];
      if (own0) v += CONE ? Du_c + Hcone : Du_c;
      const int nc = Nc * UD;
      if (L.cu && gu) a.Hc_part[(size_t)i * nc * nc + (size_t)(j * UD + g) + (size_t)nc * (j * UD + L.cb)] = v;
      st_rec(j, (L.cxv && gu) ? H[KS] : 0.0);
#pragma unroll
      for (int r = 0; r < KS; r++) S[r] = H[r];
      if (lane < UD) a.gc_part[(size_t)i * (Nc * UD) + j * UD + lane] = pick<UD>(hu, lane);
      if (j == 0) return;
      const double p2 = grp_allsum(p2q);
      s_col = L.cxv ? h_col + p2 + gx_c : 0.0;
      col_to_row<KS>(s_col, g, s_row);
      return;
    }

    TL(4);  // six MFMAs issued
    // ---- Cholesky of Huu on lane-uniform values (readlane broadcast of the lower triangle) --------
    double Lc[UD][UD], Ld[UD], col[UD];
#pragma unroll
    for (int q = 0; q < UD; q++) {
#pragma unroll
      for (int pp = q; pp < UD; pp++) {
        double v = readlane_d(H[KS], (XP + q) + 16 * pp);  // Huu[pp][q]
#pragma unroll
        for (int k = 0; k < q; k++) v -= Lc[pp][k] * Lc[q][k];
        // (the factor is lane-uniform: parked in scalar registers when SCHOL is set, which frees 2 * (u + u (u-1) / 2) vector
        // registers during the substitution phase — the difference between 3 and 4 waves per SIMD for the lean variant)
        if (pp == q) {
          bad |= !(v > 0.0);  // (reported once, behind the sweep: no branch inside the stage)
          Ld[q] = SCHOL ? rfl_d(rsqrt_d(v)) : rsqrt_d(v);
        } else {
          Lc[pp][q] = SCHOL ? rfl_d(v * Ld[q]) : v * Ld[q];
        }
      }
    }
    // ---- gather the control rows column-wise and substitute in-lane: state columns give K[:, c], the control columns
    //      get unit right-hand sides and give Huu^-1[:, c - XP] -------------------------------------------------
    TL(5);  // Cholesky factor of Huu done (its first read waited for the MFMA chain)
    double rows4[4];
    grp_gather(H[KS], rows4);
#pragma unroll
    for (int k = 0; k < UD; k++) col[k] = L.cu ? (L.cb == k ? 1.0 : 0.0) : rows4[k];
    chol_solve<UD>(Lc, Ld, col);
    const double Kg = pick<UD>(col, g);
    const double rec = frec ? Kg : 0.0;
    TL(6);  // gains by substitution
    v4d Sn = mfma(H[KS], (L.cxv && gu) ? -Kg : 0.0, H);  // S' = Hxx - Hxu K
#pragma unroll
    for (int r = 0; r < KS; r++) S[r] = Sn[r];
    st_rec(j, rec);
    if (!DEEP && MAIN) {
      __builtin_amdgcn_sched_barrier(0);  // (keep the loads HERE: hoisted above the Cholesky phase they cost the 4th wave per SIMD)
      late_pf();
    }
    const double Kreg = L.cxv ? rec : 0.0;
    // ---- feed-forward k = Huu^-1 hu: the control quad of k-group g holds row g of Huu^-1 ---------------
    const double hug = pick<UD>(hu, g);
    double kq = L.cu ? rec * pick<UD>(hu, L.cb) : 0.0;
    kq += dpp_d<0xB1>(kq);
    kq += dpp_d<0x4E>(kq);
    if (c == XP && gu) gsto(ubase(kff_, uoff(j)), lug, kq);
    TL(7);  // record + feed-forward stored
    if (!MAIN && j == 0) { TL_FLUSH(j); return; }
    // ---- s_{j-1} = h_x - K' hu + g_x,j-1 -------------------------------------------------------------
    const double red2 = grp_allsum(fma(-Kreg, hug, p2q));
    s_col = L.cxv ? h_col + red2 + gx_c : 0.0;
    col_to_row<KS>(s_col, g, s_row);
    TL(8);  // next cost-to-go gradient in place
    TL_FLUSH(j);
  };

  const int jmin = Nc > 1 ? Nc : 1;  // the MAIN body covers the free stages N-1 .. jmin
  int j = jtop;
  if (MODE == 2) {
    // three register sets in a ring, three stages per trip (static roles: no moves)
    Pipe P0, P1, P2;
    auto fetch = [&](int jj, Pipe &q) {  // everything stage jj needs (jj clamped at 0: the last stages re-read stage 0, no branch)
      const int k = jj > 0 ? jj : 0;
      fetch_early(k, q);
      fetch_late(k, k >= 1 ? k - 1 : 0, q);
    };
    fetch(jtop, P0);
    fetch(jtop - 1, P1);
    for (; j - 2 >= jmin; j -= 3) {
      fetch(j - 2, P2);
      stage(std::true_type{}, j, P0, P0);
      __builtin_amdgcn_sched_barrier(0);  // (no instruction motion across stages: overlapped stages do not fit the register file)
      fetch(j - 3, P0);
      stage(std::true_type{}, j - 1, P1, P1);
      __builtin_amdgcn_sched_barrier(0);
      fetch(j - 4, P1);
      stage(std::true_type{}, j - 2, P2, P2);
      __builtin_amdgcn_sched_barrier(0);
    }
    for (; j >= 0; j--) {  // the last free stages, the consensus stages, stage 0: ring rotated by moves
      fetch(j - 2, P2);
      if (j >= jmin) stage(std::true_type{}, j, P0, P0);
      else stage(std::false_type{}, j, P0, P0);
      P0 = P1;
      P1 = P2;
    }
  } else {
    Pipe A, B;
    fetch_early(jtop, A);
    fetch_late(jtop, jtop >= 1 ? jtop - 1 : 0, A);
#if PMPC_AS_PINGPONG
    for (; j - 1 >= jmin; j -= 2) {
      stage(std::true_type{}, j, A, B);
      __builtin_amdgcn_sched_barrier(0);  // (no instruction motion across stages: the scheduler would overlap them and spill)
      stage(std::true_type{}, j - 1, B, A);
      __builtin_amdgcn_sched_barrier(0);
    }
#else
    for (; j >= jmin; j--) {
      stage(std::true_type{}, j, A, B);
      A = B;
    }
#endif
    if (j >= jmin) {
      stage(std::true_type{}, j, A, B);
      A = B;
      j--;
    }
    for (; j >= 0; j--) {
      stage(std::false_type{}, j, A, B);
      A = B;
    }
  }
  if (bad && lane == 0) *a.fail = 2;
}

// ------------------------------------------------------------------------------------------------
// forward sweep of an active-set round: feedback law on the (clamped) state, KKT sign tests, clamping / release, new
// base point = base + clamped step (absolute), statuses, feed-forward of a settled particle's next round, counters
// ------------------------------------------------------------------------------------------------
// Loop structure as in k_bwd_as: the consensus stages and stage 0 go through the general body, the free stages j >= 1 through
// a branch-free MAIN body, two per loop trip with the prefetch register sets swapping roles.
// PF2: data requested two stages ahead (ring of three register sets; the default); else one stage ahead (two sets).  Both fit
// 4 waves per SIMD.
// Memory instructions per stage (the sweep is bound by their count, not by bytes — DESIGN.md section 6): the per-control
// inputs {base control, feed-forward, lower, upper, status} come in ONE load — lanes c = 0..3 and 7 of k-group g read one array
// each through a per-lane pointer, quad / row DPP moves gather them in lane c = 0, the only lane whose decisions count; the
// base state comes in ONE load (lane c < KS of k-group g owns row row0 + c) and the new state leaves in ONE store; the new
// base control and the next feed-forward leave in ONE store (lanes c = 0 and 1 through a per-lane pointer).
//
// MFMA form (r02; the first version of this sweep did the two products as row sums over DPP moves): per stage the state
// update and the feedback are ONE 16 x 16 tile product,
//     [dx_j ; K_{j+1}..] = [[fx_j, fu_j], [K_j, 0]] [dx_{j-1} ; du_j]   in two passes (K dx first: du needs it, then + B du),
// and the C layout of the result (lane (c, g), register r <-> row g + 4r) is exactly the B-operand layout of the next
// stage's state, so the state never moves between lanes: no row sums over DPP moves (36 per stage before), no lane
// shuffles (2 LDS round trips per stage before).  The tile is the A operand: lanes (c, g) load T[c][g + 4r] — state rows from
// fx / fu, the udim gain rows from the factor record — with one per-lane pointer, KS + 1 loads per stage (as many as before).
// Columns 0..3 of the product carry the same state vector (so every lane of a k-group that stores or decides has it), the
// other twelve stay zero.
// CONE: also records each stage's own Newton step u_b + du BEFORE clamping (a.as_uraw: the cone multiplier updates of
// kernels_cone.hip are valid for that step only) — a third lane of the per-control store, no further instruction.
// SENS (one consensus stage; a.as_T): the sweep also carries the SENSITIVITY of the particle's closed-loop trajectory to the step of
// the shared controls — UD more columns of the same tile products (the tile has 16 columns, the state uses 4: the arithmetic is there
// anyway) — and leaves it as one 64-double record per stage, as_T[j] = d(x_j, du_j) / d delta.  A SETTLED particle of a later round
// (no status change, no factor sweep) then does not walk the horizon again: its wave takes the consensus stage as usual and updates
// the other stages elementwise, four stages per instruction — X += T_x delta, U += T_u delta on the free controls, the multipliers of
// the held ones by the same rule, with the sweep's own box / sign tests.  A violation un-settles the particle (it counts as a change:
// the round is not accepted, the next one sweeps the particle), so an accepted set is exactly as exact as before.  Why it pays: at
// 4096 particles the sweeps are bound by instruction issue (profiles/r05_pmc_sq_D.txt), a sweep stage costs ~260 instructions, an
// elementwise stage ~10, and from the second round on 50 - 99 % of the particles are settled: a later round's launch takes 43 - 65 us
// instead of 113 (bound by the 0.9 KB per stage the elementwise update moves).  Measured same-box: 4096 particles +9 %, 2048 and fewer
// -4 .. -6 % (two waves per SIMD: the launch waits for the few sweeping waves either way, and the records cost the first round
// 25 us) — solver.hip switches it on from 3072 particles per rank (option as_sens_min_m).
template <int XD, int UD, bool DEFECT, bool PF2, bool CONE = false, class MT = double, bool SENS = false>
__global__ void __launch_bounds__(64, 4) k_fwd_as(LQArgs a) {
  typedef Lane<XD, UD> LT;
  constexpr int KS = LT::KS;
  constexpr bool PADX = (XD != LT::XP);
  constexpr long long D8 = sizeof(double);
  if (a.done && *a.done) return;
  const int lane = threadIdx.x;
  const LT L(lane);
  const int N = a.N, Nc = a.Nc, i = a.as_perm ? __builtin_amdgcn_readfirstlane(a.as_perm[blockIdx.x]) : (int)blockIdx.x, g = L.g, c = L.c;
  const size_t pbase = (size_t)i * N;
  const bool gu = g < UD;
  const double *Z = a.zeros;
  // The stage matrix T = [[fx, fu], [K, 0]] ((XP + udim) x 16, kernel coordinates) as the A operand of the MFMA chain: this
  // lane supplies T[rho = c][kappa = g + 4r], r = 0 .. KS (kappa < XP: state column, original index KS g + r; kappa = XP + g:
  // control column g).  State rows rho < XP read fx / fu (original row pi(rho) = L.oc), gain rows rho = XP + b read row b of
  // the factor record (record[kernel column + 16 b]); every other lane reads the zero buffer through a zero stride.
  constexpr int MB = (int)sizeof(MT);  // bytes per entry of fx, fu and the factor record (float: the fp32-storage mode)
  const char *pA = (const char *)Z, *pB = (const char *)Z;
  int sAr = 0, sAj = 0, sBj = 0;  // byte strides: between the steps r < KS, between stages (A), between stages (B: step KS)
  if (L.cxv) {
    pA = (const char *)((const MT *)a.fx + pbase * (XD * XD) + (size_t)XD * (KS * g) + L.oc);
    sAr = XD * MB; sAj = XD * XD * MB;
    if (gu) { pB = (const char *)((const MT *)a.fu + pbase * (XD * UD) + (size_t)XD * g + L.oc); sBj = XD * UD * MB; }
  } else if (L.cu) {
    pA = (const char *)((const MT *)a.K + pbase * 64 + 16 * L.cb + g);
    sAr = 4 * MB; sAj = 64 * MB;
  }
  bool vA[KS];  // (padding: state column KS g + r beyond xdim -> zero; the record holds zeros there already)
#pragma unroll
  for (int r = 0; r < KS; r++) vA[r] = !L.cxv || !PADX || (KS * g + r < XD);
  const double *Xb = DEFECT ? a.X_prev : a.Xb, *Ub = DEFECT ? a.U_prev : a.Ub;
  // sign tolerance of the multipliers, in units of THIS particle's cost weight (a down-weighted particle of the cone path has
  // proportionally small multipliers: an absolute tolerance would freeze its weakly active bounds)
  const double pwi = a.pw ? a.pw[i] : 1.0;
  const double tol_l = (a.as_ctl ? a.as_ctl->tol_l : a.as_tol_l) * pwi;
  // uniform stage bases + per-lane constant byte offsets (see k_bwd_as)
  const unsigned lug = (unsigned)((gu ? g : 0) * D8);
  const bool own_x = c < KS && (!PADX || L.row0 + c < XD);  // this lane owns state row row0 + c
  const unsigned lx1 = own_x ? (unsigned)((L.row0 + c) * D8) : 0u;
  const long long px = (long long)(pbase * XD) * D8, pu = (long long)(pbase * UD) * D8;
  const double *Xb_ = ubase(Xb, px), *f_ = ubase(a.f, px), *Xo_ = ubase(a.Xo, px);
  const double *act_ = ubase((const double *)a.as_act, pu >> 1);
  const double *T_ = SENS ? ubase(a.as_T, (long long)(pbase * 64) * D8) : Z;
  const unsigned lT = (unsigned)((((c - 4) & 3) * 16 + 4 * g) * D8);  // this lane's 32 bytes of a stage's sensitivity record (lanes c = 4 .. 4 + UD - 1)
  auto xoff = [&](int jj) { return (long long)(jj * (int)(XD * D8)); };
  auto uoff = [&](int jj) { return (long long)(jj * (int)(UD * D8)); };
  // per-control inputs of control g: lanes c = 0..3 of the k-group read {U base, feed-forward, lower, upper}[c], lane 7 the status
  const char *pG = (const char *)Z;
  int sG = 0;
  if (gu && c < 4) {
    const double *arr = c == 0 ? Ub : (c == 1 ? a.kff : (c == 2 ? a.as_lo : a.as_hi));
    pG = (const char *)(arr + pbase * UD + g);
    sG = UD * (int)D8;
  } else if (gu && c == 7) {  // 4-byte entries read as 8 bytes: the low word counts (the buffer ends with 8 spare bytes)
    pG = (const char *)(a.as_act + pbase * UD + g);
    sG = UD * (int)sizeof(int);
  }
  // outputs of control g: lane 0 -> new base control, lane 1 -> feed-forward of the next round
  char *pS = (char *)((c == 1 ? a.kff : ((CONE && c == 2) ? a.as_uraw : a.Uo)) + pbase * UD + (gu ? g : 0));
  const bool st_lane = gu && c < (CONE ? 3 : 2);
  const bool store_u = (c == 0) && gu;
  // what a stage needs when it starts: this lane's entries of T, the per-control word of this lane, this lane's base-state row
  struct Pipe { double T[KS + 1], grp, xb, f; };
  int jF = 0;  // stage the pointers point at
  auto fetch = [&](int jj, Pipe &q) {  // called in ascending stage order (a clamped repeat of the last stage leaves the pointers alone)
    if (jj > jF) { pA += sAj; pB += sBj; pG += sG; jF = jj; }
#pragma unroll
    for (int r = 0; r < KS; r++) {
      const double t = gldm<MT>(pA + (vA[r] ? r * sAr : 0));
      q.T[r] = vA[r] ? t : 0.0;
    }
    q.T[KS] = gldm<MT>(pB);
    q.grp = gld(pG);
    q.xb = ldo(ubase(Xb_, xoff(jj)), lx1);
    if (DEFECT) q.f = ldo(ubase(f_, xoff(jj)), lx1);
  };

  double V[KS];  // dx in the MFMA's B / C layout: V[r] = dx[kernel row g + 4r], the same in every lane of the k-group that counts (c < 4)
#pragma unroll
  for (int r = 0; r < KS; r++) V[r] = 0.0;
  int nrel = 0, nadd = 0, nbad = 0;
  int jh = -1;  // highest stage with a status change that counts (stages ascend: the last one seen)
  double vworst = 0.0;
  double dcap = 0.0;  // SENS: the step of shared control g as applied (lane c = 0 of k-group g)
  // settled: the stages behind the consensus stage elementwise (never in a DEFECT sweep: the first round of an attempt sweeps every particle)
  const bool elem = SENS && !DEFECT && a.as_settled_in && a.as_settled_in[i];
  TL_DECL(2);
  // consensus step of k-group g when this wave solves the consensus system itself (a.cons_G block partials; Nc == 1): lane e
  // sums entry e of [H | g] in block order — the same operations in every wave, so every particle applies the same step —,
  // then a Cholesky solve on lane-uniform values (arithmetic of k_cons_small's single-thread solve)
  double dcons = 0.0;
  bool frozen = false;
  if (a.cons_G) {
    constexpr int nH = UD * UD, E = nH + UD;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (lane < E) {
      const double *src = lane < nH ? a.cons_tH + lane : a.cons_tg + (lane - nH);
      const int stride = lane < nH ? nH : UD;
      int k = 0;
      for (; k + 3 < a.cons_G; k += 4) {
        a0 += src[(size_t)k * stride];
        a1 += src[(size_t)(k + 1) * stride];
        a2 += src[(size_t)(k + 2) * stride];
        a3 += src[(size_t)(k + 3) * stride];
      }
      for (; k < a.cons_G; k++) a0 += src[(size_t)k * stride];
    }
    const double acc = (a0 + a1) + (a2 + a3);
    double Lm[UD][UD], y[UD];
    bool cbad = false;
#pragma unroll
    for (int q = 0; q < UD; q++) {
      double d = readlane_d(acc, q + UD * q);
#pragma unroll
      for (int k = 0; k < q; k++) d -= Lm[q][k] * Lm[q][k];
      cbad |= !(d > 0.0);
      d = sqrt(cbad ? 1.0 : d);
      Lm[q][q] = d;
#pragma unroll
      for (int p = q + 1; p < UD; p++) {
        double v = readlane_d(acc, q + UD * p);  // upper triangle (the only one the sweeps fill)
#pragma unroll
        for (int k = 0; k < q; k++) v -= Lm[p][k] * Lm[q][k];
        Lm[p][q] = v / d;
      }
    }
#pragma unroll
    for (int p = 0; p < UD; p++) {
      double v = -readlane_d(acc, nH + p);
#pragma unroll
      for (int k = 0; k < p; k++) v -= Lm[p][k] * y[k];
      y[p] = v / Lm[p][p];
    }
#pragma unroll
    for (int p = UD - 1; p >= 0; p--) {
      double v = y[p];
#pragma unroll
      for (int k = p + 1; k < UD; k++) v -= Lm[k][p] * y[k];
      y[p] = v / Lm[p][p];
    }
    if (CONE && a.as_freeze_tol > 0.0) {
      // (a HELD shared control's entry is its multiplier over the 1e30 penalty: left alone.  Statuses and base value: this particle's own
      //  copy of the consensus stage — identical in every particle, and not yet rewritten by this wave)
      const int *act0 = a.as_act + pbase * UD;
      const double *ub0 = Ub + pbase * UD;
      double ym = 0.0;
      bool asks_release = false;  // a HELD shared control whose multiplier has the wrong sign: the consensus stage has a decision to take
#pragma unroll
      for (int p = 0; p < UD; p++) {
        if (act0[p] == 0) ym = fmax(ym, fabs(y[p]) / fmax(1.0, fabs(ub0[p])));
        else asks_release |= (act0[p] == 1 ? -a.as_big * y[p] : a.as_big * y[p]) < -tol_l;
      }
      frozen = ym <= a.as_freeze_tol && !asks_release;
      if (frozen) {
#pragma unroll
        for (int p = 0; p < UD; p++)
          if (act0[p] == 0) y[p] = 0.0;
      }
    }
    dcons = gu ? pick<UD>(y, g) : 0.0;
    if (cbad && i == 0 && lane == 0) *a.fail = 2;
  }
  if (CONE && frozen && a.as_settled_in && a.as_settled_in[i]) {
    // nothing of this particle changed in the last round (no factor sweep for it in this one) and the shared step is zero: its forward
    // sweep would write back what is there.  The counters say "no change".
    if (lane == 0) {
      a.as_cnt[3 * i + 0] = 0; a.as_cnt[3 * i + 1] = 0; a.as_cnt[3 * i + 2] = 0;
      if (a.as_settled_out) a.as_settled_out[i] = 1;
      if (a.as_jhi) a.as_jhi[i] = -1;
      a.as_open[i] = 0;
      if (a.as_viol) a.as_viol[i] = 0.0;
    }
    if (i == 0 && lane < UD) a.as_delta[lane] = 0.0;
    return;
  }
  const double inv_dual = 1.0 / ((a.as_ctl ? a.as_ctl->dual_scale : 1.0) * pwi);
  auto stage = [&](auto main_tag, const int j, const Pipe &cur) {
    constexpr bool MAIN = decltype(main_tag)::value;
    TL(0);  // stage entry
    // first pass: [A dx ; K dx] of the incoming state (stage 0 has none: A~_0 = 0 and nothing to feed back)
    v4d D = {0.0, 0.0, 0.0, 0.0};
    if (MAIN || j > 0) {
#pragma unroll
      for (int r = 0; r < KS; r++) D = mfma(cur.T[r], V[r], D);
    }
    TL(1);  // first-pass MFMAs issued (waited for this stage's loads)
    const double raw = D[KS];  // (K dx)[g] in every lane of k-group g
    // gather the per-control inputs in lane c = 0 (the other lanes of the k-group compute on whatever they get: ignored)
    const double ubc = cur.grp;                // lane 0's own word
    const double kc = dpp_d<0x55>(cur.grp);    // quad broadcast of lane 1
    const double loc = dpp_d<0xAA>(cur.grp);   // lane 2
    const double hic = dpp_d<0xFF>(cur.grp);   // lane 3
    const int actc = __builtin_amdgcn_mov_dpp(__double2loint(cur.grp), 0x141, 0xF, 0xF, true);  // half-row mirror: lane 7 -> lane 0
    // base control of this stage: in the first round of a warm start the caller's U_prev, which must already BE the base point
    // of the stored set (inside its box, exactly on the bound where held) — else the caller's promise does not hold
    TL(2);  // control word gathered
    if (DEFECT) {
      const double snapped = actc == 1 ? loc : (actc == 2 ? hic : fmin(fmax(ubc, loc), hic));
      nbad |= (store_u && !(snapped == ubc)) ? 1 : 0;  // (also catches an empty box and a NaN)
      // a shared control has ONE base value: the caller's U_prev must hold particle 0's in every particle
      if (!MAIN && j < Nc) nbad |= (store_u && !(ubc == Ub[(size_t)j * UD + g])) ? 1 : 0;
    } else {
      nbad |= (store_u && !(loc <= hic)) ? 1 : 0;  // an empty box ends the solve as the reference's does (NaN outputs)
    }
    // du[g] in every lane of k-group g: the shared consensus step (identical in every particle, so are the decisions)
    // resp. the feedback law on the (clamped) state
    double draw;
    if (!MAIN && j < Nc) draw = a.cons_G ? dcons : (gu ? a.duc[j * UD + g] : 0.0);
    else draw = -raw - kc;
    const bool cnt_here = store_u && (MAIN || j >= Nc || i == 0);
    const bool held = store_u && actc != 0;
    const double lam = actc == 1 ? -a.as_big * draw : a.as_big * draw;  // multiplier of the held side
    const bool release = held && lam < -tol_l;
    const double zt = ubc + draw;
    const bool vlo = store_u && !held && zt < loc - a.as_tol_p * fmax(1.0, fabs(loc));
    const bool vhi = store_u && !held && !vlo && zt > hic + a.as_tol_p * fmax(1.0, fabs(hic));
    const int anew = release ? 0 : (vlo ? 1 : (vhi ? 2 : actc));
    // held: no step (a released control starts the next round from its bound); a control that would leave its box is clamped
    // onto the bound and held from now on.  The new base control is exactly the bound on every held control.
    const double unew = held ? ubc : (vlo ? loc : (vhi ? hic : zt));
    const double dug = unew - ubc;
    nbad |= (store_u && !(draw == draw)) ? 1 : 0;
    nrel += (cnt_here && release) ? 1 : 0;
    nadd += (cnt_here && (vlo || vhi)) ? 1 : 0;
    jh = (cnt_here && (release || vlo || vhi)) ? j : jh;
    {  // size of the violation behind a change (diagnostic / acceptance of changes at round-off level)
      // (a diagnostic: the hardware reciprocal will do — two IEEE divisions per stage were 30 of the sweep's ~ 260 instructions)
      const double pv = (vlo ? loc - zt : (vhi ? zt - hic : 0.0)) * __builtin_amdgcn_rcp(fmax(1.0, fabs(vlo ? loc : hic)));
      const double dv = release ? -lam * inv_dual : 0.0;
      vworst = fmax(vworst, cnt_here ? fmax(pv, dv) : 0.0);
    }
    TL(3);  // decisions taken (the first use of `raw` waited for the MFMA chain)
    const double du_q = dpp_d<0x00>(dug);  // the decision of lane 0, in the four lanes of the k-group whose columns are kept
    if (SENS && !MAIN && j < Nc) dcap = dug;
    // feed-forward of the NEXT round if this particle stays settled (no factor sweep then): at base + step every free
    // control is stationary (k = 0) and a held one keeps its multiplier, k_b = -du_b
    const double knew = actc ? -draw : 0.0;
    if (MAIN || j >= Nc) {
      const double k1 = dpp_d<0x00>(knew);  // quad broadcast of lane 0
      const double z1 = CONE ? dpp_d<0x00>(zt) : 0.0;
      if (st_lane) gst(pS, (c == 0) ? unew : ((CONE && c == 2) ? z1 : k1));
    } else {
      // (one branch for both stores: the quad broadcast must see lane 0 enabled wherever the compiler puts it)
      const double z1 = CONE ? dpp_d<0x00>(zt) : 0.0;
      if (store_u || (CONE && gu && c == 2)) gst(pS, (c == 0) ? unew : z1);
      if (store_u && i == 0) a.as_delta[j * UD + g] = dug;  // the consensus step as applied (settled particles: g_i += H_i delta)
    }
    pS += UD * (int)D8;
    if (store_u) gsto_i(ubase(act_, uoff(j) >> 1), lug >> 1, anew);
    TL(4);  // control / status stores issued
    // second pass: + B du.  The C layout of the result IS the B layout of the next stage's state: nothing moves.
    D[KS] = 0.0;
    double bop = (c < 4 && gu) ? du_q : 0.0;
    if (SENS) {  // sensitivity columns c = 4 .. 4 + UD - 1: unit steps of the shared controls on the consensus stage, the feedback law behind it
      const bool sc = c >= 4 && c < 4 + UD && gu;
      const double bs = (!MAIN && j < Nc) ? ((g == c - 4) ? 1.0 : 0.0) : -raw;
      bop = sc ? bs : bop;
    }
    D = mfma(cur.T[KS], bop, D);
    if (SENS) {  // the stage's record: lane (4 + k, g) owns slots 4 g .. 4 g + 3 of column k (KS state rows, then control g): 32 bytes per lane
      if (c >= 4 && c < 4 + UD) {
        typedef double v2dd __attribute__((ext_vector_type(2)));
        const v2dd lo2 = {D[0], KS > 1 ? D[KS > 1 ? 1 : 0] : 0.0}, hi2 = {KS > 2 ? D[KS > 2 ? 2 : 0] : 0.0, gu ? bop : 0.0};
        __attribute__((address_space(1))) char *t = (__attribute__((address_space(1))) char *)(unsigned long long)ubase(T_, (long long)(j * (int)(64 * D8))) + lT;
        *(v2dd __attribute__((address_space(1))) *)t = lo2;
        *(v2dd __attribute__((address_space(1))) *)(t + 16) = hi2;
      }
    }
    if (DEFECT) {  // the defect r = f - x_prev of row KS g + r sits in lane c = r of the k-group: quad broadcasts
      const double dfo = own_x ? cur.f - cur.xb : 0.0;
      D[0] += dpp_d<0x00>(dfo);
      if (KS > 1) D[1] += dpp_d<0x55>(dfo);
      if (KS > 2) D[2] += dpp_d<0xAA>(dfo);
    }
    if (own_x) {
      double mine = D[0];
#pragma unroll
      for (int r = 1; r < KS; r++) mine = (c == r) ? D[r] : mine;
      gsto(ubase(Xo_, xoff(j)), lx1, cur.xb + mine);
    }
#pragma unroll
    for (int r = 0; r < KS; r++) V[r] = (c < (SENS ? 4 + UD : 4)) ? D[r] : 0.0;  // (the other columns of the tile carry nothing: kept at zero)
    TL(5);  // new state stored and in place
    TL_FLUSH(j);
  };

  // SENS: a settled particle's stages behind the consensus stage, elementwise from the sensitivity records (see the kernel's header
  // comment).  Lane (q4, s16) = stage q4 of a group of four x slot s16 of the record (slot 4 gs + rs: state row KS gs + rs for
  // rs < KS, control gs for rs = 3); SB groups per trip, every load of the trip issued before anything is stored.
  int nrel_e = 0, nadd_e = 0, nbad_e = 0, jh_e = -1;
  auto settled_update = [&](int j0) {
    double dl[UD];
#pragma unroll
    for (int k = 0; k < UD; k++) dl[k] = readlane_d(dcap, 16 * k);
    const int s16 = lane & 15, q4 = lane >> 4, gs = s16 >> 2, rs = s16 & 3;
    const bool is_x = rs < KS && KS * gs + rs < XD, is_u = rs == 3 && gs < UD;
    const size_t xo = (size_t)(is_x ? KS * gs + rs : 0), uo = (size_t)(is_u ? gs : 0);
    constexpr int SB = 4;
    for (int j4 = j0; j4 < N; j4 += 4 * SB) {
      double dv[SB], xv[SB], ub[SB], kf[SB], lo[SB], hi[SB];
      int act[SB];
#pragma unroll
      for (int q = 0; q < SB; q++) {
        const int jj = j4 + 4 * q + q4, jc = jj < N ? jj : N - 1;  // (clamped: the tail re-reads the last stage, its results are dropped)
        const double *t = a.as_T + (pbase + jc) * 64 + s16;
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < UD; k++) acc = fma(gld(t + 16 * k), dl[k], acc);
        dv[q] = acc;
        xv[q] = is_x ? gld(Xb + (pbase + jc) * XD + xo) : 0.0;
        const size_t e = (pbase + jc) * UD + uo;
        ub[q] = is_u ? gld(Ub + e) : 0.0;
        kf[q] = is_u ? gld(a.kff + e) : 0.0;
        lo[q] = is_u ? gld(a.as_lo + e) : 0.0;
        hi[q] = is_u ? gld(a.as_hi + e) : 0.0;
        act[q] = is_u ? a.as_act[e] : 0;
      }
#pragma unroll
      for (int q = 0; q < SB; q++) {
        const int jj = j4 + 4 * q + q4;
        const bool live = jj < N;
        if (is_x && live) gst(a.Xo + (pbase + jj) * XD + xo, xv[q] + dv[q]);
        if (is_u && live) {
          const size_t e = (pbase + jj) * UD + uo;
          if (act[q] == 0) {
            const double zt = ub[q] + dv[q];
            const bool vlo = zt < lo[q] - a.as_tol_p * fmax(1.0, fabs(lo[q])), vhi = !vlo && zt > hi[q] + a.as_tol_p * fmax(1.0, fabs(hi[q]));
            nadd_e += (vlo || vhi) ? 1 : 0;  // (outside its box and still free: the next round sweeps this particle and clamps it)
            nbad_e |= !(zt == zt) ? 1 : 0;
            gst(a.Uo + e, zt);
            if (CONE) gst(a.as_uraw + e, zt);  // the stage's own Newton step before clamping (what the cone pass updates its multipliers with)
          } else {
            const double kn = kf[q] - dv[q];  // kff holds -du_b of the held control: its multiplier is -/+ big du_b
            const double lam = act[q] == 1 ? a.as_big * kn : -a.as_big * kn;
            const bool release = lam < -tol_l;
            nrel_e += release ? 1 : 0;
            jh_e = (release && jj > jh_e) ? jj : jh_e;
            nbad_e |= !(kn == kn) ? 1 : 0;
            gst(a.kff + e, kn);
            if (release) a.as_act[e] = 0;
            if (a.Uo != Ub) gst(a.Uo + e, ub[q]);
            if (CONE) gst(a.as_uraw + e, ub[q] + dv[q]);
          }
        }
      }
    }
  };

  auto finish = [&]() {
    // counters of this particle: the store_u lanes (c == 0, g < udim) counted; sum / or over the k-groups
    double r = grp_allsum((double)nrel), d = grp_allsum((double)nadd), b = grp_allsum((double)nbad);
    if (SENS && elem) {  // the elementwise path counted in the control slots of every stage group: sum over the wave
      r += grp_allsum(row_allsum((double)nrel_e)); d += grp_allsum(row_allsum((double)nadd_e)); b += grp_allsum(row_allsum((double)nbad_e));
      int m = jh_e;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { const int v = __shfl_xor(m, o, 64); m = v > m ? v : m; }
      jh = m > jh ? m : jh;  // (every lane holds the maximum: lane 0's copy is the one written below)
    }
    if (a.as_viol) {  // max over the k-groups (the counting lanes are c == 0)
      double p01, p23, q0, q1;
      swap32_d(vworst, p01, p23);
      const double m = fmax(p01, p23);
      swap16_d(m, q0, q1);
      if (lane == 0) a.as_viol[i] = fmax(q0, q1);
    }
    if (a.as_jhi) {  // max over the k-groups' counting lanes (c == 0: lanes 0, 16, 32, 48)
      int m = jh;
#pragma unroll
      for (int k = 1; k < 4; k++) { const int o = __builtin_amdgcn_readlane(jh, 16 * k); m = o > m ? o : m; }
      if (lane == 0) a.as_jhi[i] = m;
    }
    if (lane == 0) {
      a.as_cnt[3 * i + 0] = (int)r;
      a.as_cnt[3 * i + 1] = (int)d;
      a.as_cnt[3 * i + 2] = b > 0.0 ? 1 : 0;
      if (a.as_settled_out) a.as_settled_out[i] = (r == 0.0 && d == 0.0 && !(b > 0.0)) ? 1 : 0;
      if (CONE) a.as_open[i] = 0;  // (counted by the cone pass that follows)
    }
  };

  const int jmin = Nc > 1 ? Nc : 1;  // MAIN covers the free stages jmin .. N-1
  auto clampN = [&](int jj) { return jj < N ? jj : N - 1; };  // (the last stages re-read the last one: no branch)
  int j = 0;
  if constexpr (SENS && !DEFECT) if (elem) {  // (a path of its own, ahead of the sweep's register sets: nothing of the sweep is live beside the elementwise loads)
    Pipe P;
    fetch(0, P);
    stage(std::false_type{}, 0, P);  // the consensus stage (Nc == 1): decisions of the shared controls, the step as applied
    settled_update(1);
    finish();
    return;
  }
  if (PF2) {
    // Prefetch distance: TWO stages.  With one wave per SIMD (small shards, the later rounds' few unsettled particles) a stage
    // of this sweep is shorter than the HBM latency, so data requested one stage ahead would still pin every stage to that
    // latency.  Three register sets in a ring; the main loop runs three stages per trip so that their roles are static (no moves).
    // (Three stages ahead, four sets, four stages per trip: measured 40 % SLOWER at every size — not kept.)
    Pipe P0, P1, P2;
    fetch(0, P0);
    fetch(clampN(1), P1);
    for (; j < jmin && j < N; j++) {  // consensus stages / stage 0: general body, ring rotated by moves
      fetch(clampN(j + 2), P2);
      stage(std::false_type{}, j, P0);
      P0 = P1;
      P1 = P2;
    }
    for (; j + 2 < N; j += 3) {
      fetch(clampN(j + 2), P2);
      stage(std::true_type{}, j, P0);
      fetch(clampN(j + 3), P0);
      stage(std::true_type{}, j + 1, P1);
      fetch(clampN(j + 4), P1);
      stage(std::true_type{}, j + 2, P2);
    }
    if (j < N) { stage(std::true_type{}, j, P0); j++; }  // (0 .. 2 stages left; their data is already in flight / landed:
    if (j < N) { stage(std::true_type{}, j, P1); j++; }  //  after a full trip the ring holds stage j in P0 and j + 1 in P1)
  } else {
    Pipe A, B;
    fetch(0, A);
    for (; j < jmin && j < N; j++) {
      fetch(clampN(j + 1), B);
      stage(std::false_type{}, j, A);
      A = B;
    }
    for (; j < N; j++) {
      fetch(clampN(j + 1), B);
      stage(std::true_type{}, j, A);
      A = B;
    }
  }
  finish();
}

// ------------------------------------------------------------------------------------------------
// round control (one block): particle sums of {released, activated, bad} + the solve's failure flag -> the device-side decision
// the host used to take after reading them back.  Single rank: called right behind the forward sweep with reduce = 1,
// decide = 1.  Sharded: reduce = 1 (local counters -> ctl->cnt), all-reduce of ctl->cnt, then decide = 1.
// ------------------------------------------------------------------------------------------------
// control block of a fresh attempt (one thread): replaces a host -> device copy and a memset per solve
__global__ void k_as_begin(AsCtl *ctl, int *fail, int max_rounds, double dual_scale, int stall_limit) {
  if (threadIdx.x == 0) {
    *fail = 0;
    AsCtl h = {};
    h.max_rounds = max_rounds;
    h.stall_limit = stall_limit;
    h.last_changes = 0x7fffffff;
    h.dual_scale = dual_scale;
    h.tol_l = dual_scale * 1e-11;
    *ctl = h;
  }
}

// `tail` (sharded runs, consensus horizon > 0): the four counters travel as doubles behind [Hc | gc] in the NEXT round's
// consensus all-reduce instead of in a collective of their own — reduce = 1 packs the local sums into tail[0..3], decide = 1
// reads the all-reduced values from there.
__global__ void __launch_bounds__(1024) k_as_ctl(AsCtl *ctl, const int *cnt_part, int M, const int *fail, int reduce, int decide, int last_of_batch,
                                                 AsCtl *mirror, unsigned long long *mirror_seq, unsigned long long seq, double *tail, const double *viol,
                                                 const int *open_part) {
  as_ctl_block(ctl, cnt_part, M, fail, reduce, decide, last_of_batch, mirror, mirror_seq, seq, tail, viol, open_part);
}

// Order of the particles for a later round's launches: the UNSETTLED ones first (ascending), then the settled ones.  A later round's sweeps
// do long work for the unsettled particles only (restarted factor sweep, forward sweep) and next to nothing for the settled ones (g_i += H_i
// delta; elementwise update).  Launched in index order the long waves land where their indices put them: with a quarter of 4096 particles
// unsettled, 6 % of the SIMDs hold three or four long waves and the launch takes what a full launch takes (measured at config E: 210 of
// 285 us with 27 % unsettled).  Workgroups are dispatched in order, so with the unsettled particles in front they spread one per SIMD.
// One block of 1024 threads: counts, exclusive scan, scatter.
__global__ void __launch_bounds__(1024) k_as_perm(const int *settled, int M, int *perm, const int *done) {
  if (done && *done) return;
  as_perm_block(settled, M, perm);
}

// CONE instantiations of the two sweeps: every compiled (xdim, udim) pair with udim >= 2 (ten more kernels each)
template <int XD, int UD>
constexpr bool cone_dims() { return UD >= 2; }
// (xdim, udim) pairs with fp32-storage instantiations (class MT = float) of the two sweeps
template <int XD, int UD>
constexpr bool f32_dims() { return (XD == 12 && UD == 4) || (XD == 6 && UD == 3) || (XD == 4 && UD == 2); }
// XBOX instantiations of the factor sweep: every compiled (xdim, udim) pair (five more kernels each)
template <int XD, int UD>
constexpr bool xbox_dims() { return true; }
template <int XD, int UD>
void launch_bwd_as_t(const LQArgs &a, hipStream_t s) {
  // waves per SIMD this launch brings (1024 SIMDs): <= 2 deep2, <= 3 deep, else lean (see k_bwd_as)
  static const int m2 = getenv("PMPC_AS_DEEP2_MAXM") ? atoi(getenv("PMPC_AS_DEEP2_MAXM")) : PMPC_AS_DEEP2_MAXM;
  static const int m1 = getenv("PMPC_AS_DEEP_MAXM") ? atoi(getenv("PMPC_AS_DEEP_MAXM")) : PMPC_AS_DEEP_WAVES * 1024;
  int mode = a.M <= m2 ? 2 : (a.M <= m1 ? 1 : 0);
  // the DEFECT instantiation of the deep variant needs 127 registers (4 waves per SIMD without help): never the lean one
  if (a.defect && mode == 0) mode = 1;
 const dim3 grd(a.M), blk(64);
  if (a.mat32) {  // fp32-storage mode: the deep variants, with or without stage cones
    if constexpr (f32_dims<XD, UD>()) {
      if (mode == 0) mode = 1;
#define PMPC_BWD32(MD, SK, DF, CN) hipLaunchKernelGGL((k_bwd_as<XD, UD, MD, SK, DF, CN, float>), grd, blk, 0, s, a)
#define PMPC_BWD32_C(MD, SK, DF) do { if (a.cone_H) PMPC_BWD32(MD, SK, DF, true); else PMPC_BWD32(MD, SK, DF, false); } while (0)
      if (a.defect) { if (mode == 2) PMPC_BWD32_C(2, false, true); else PMPC_BWD32_C(1, false, true); }
      else if (a.as_settled_in) PMPC_BWD32_C(2, true, false);
      else { if (mode == 2) PMPC_BWD32_C(2, false, false); else PMPC_BWD32_C(1, false, false); }
#undef PMPC_BWD32_C
#undef PMPC_BWD32
      return;
    } else {
      abort();  // (solver.hip asks f32_as_dims_supported first)
    }
  }
  if (a.cone_H) {  // stage cones: the deep variants only (one more register per prefetch set)
    if constexpr (cone_dims<XD, UD>()) {
      if (mode == 0) mode = 1;
      if (a.defect) {
        if (mode == 2) hipLaunchKernelGGL((k_bwd_as<XD, UD, 2, false, true, true>), grd, blk, 0, s, a);
        else hipLaunchKernelGGL((k_bwd_as<XD, UD, 1, false, true, true>), grd, blk, 0, s, a);
      } else if (a.as_settled_in) {
        hipLaunchKernelGGL((k_bwd_as<XD, UD, 2, true, false, true>), grd, blk, 0, s, a);
      } else {
        if (mode == 2) hipLaunchKernelGGL((k_bwd_as<XD, UD, 2, false, false, true>), grd, blk, 0, s, a);
        else hipLaunchKernelGGL((k_bwd_as<XD, UD, 1, false, false, true>), grd, blk, 0, s, a);
      }
      return;
    } else {
      abort();  // (solver.hip asks cone_as_dims_supported first)
    }
  }
  if (a.xb_D) {  // state boxes: the deep variants only
    if constexpr (xbox_dims<XD, UD>()) {
      if (mode == 0) mode = 1;
      if (a.defect) {
        if (mode == 2) hipLaunchKernelGGL((k_bwd_as<XD, UD, 2, false, true, 2>), grd, blk, 0, s, a);
        else hipLaunchKernelGGL((k_bwd_as<XD, UD, 1, false, true, 2>), grd, blk, 0, s, a);
      } else if (a.as_settled_in) {
        hipLaunchKernelGGL((k_bwd_as<XD, UD, 2, true, false, 2>), grd, blk, 0, s, a);
      } else {
        if (mode == 2) hipLaunchKernelGGL((k_bwd_as<XD, UD, 2, false, false, 2>), grd, blk, 0, s, a);
        else hipLaunchKernelGGL((k_bwd_as<XD, UD, 1, false, false, 2>), grd, blk, 0, s, a);
      }
      return;
    } else {
      abort();  // (solver.hip asks xbox_as_dims_supported first)
    }
  }
#define PMPC_BWD_AS(SK, DF)                                                                      \
  do {                                                                                           \
    if (mode == 2) hipLaunchKernelGGL((k_bwd_as<XD, UD, 2, SK, DF>), grd, blk, 0, s, a);         \
    else if (mode == 1) hipLaunchKernelGGL((k_bwd_as<XD, UD, 1, SK, DF>), grd, blk, 0, s, a);    \
    else hipLaunchKernelGGL((k_bwd_as<XD, UD, 0, SK, DF>), grd, blk, 0, s, a);                   \
  } while (0)
  if (a.defect) PMPC_BWD_AS(false, true);
  else if (a.as_settled_in) {  // few particles left: the latency regime at every M
    if (m2 > 0) hipLaunchKernelGGL((k_bwd_as<XD, UD, 2, true, false>), grd, blk, 0, s, a);
    else hipLaunchKernelGGL((k_bwd_as<XD, UD, 1, true, false>), grd, blk, 0, s, a);
  }
  else PMPC_BWD_AS(false, false);
#undef PMPC_BWD_AS
}
template <int XD, int UD>
void launch_fwd_as_t(const LQArgs &a, hipStream_t s) {
  // (two-stage prefetch everywhere since the merged loads freed the registers for 4 waves per SIMD: +0.8 % at 4096 particles;
  //  PMPC_AS_FWD_PF2_MAXM=<M> puts larger launches back on the one-stage variant)
  static const int m2 = getenv("PMPC_AS_FWD_PF2_MAXM") ? atoi(getenv("PMPC_AS_FWD_PF2_MAXM")) : (1 << 30);
  const dim3 grd(a.M), blk(64);
  if (a.mat32) {
    if constexpr (f32_dims<XD, UD>()) {
      if (a.as_uraw && a.as_T && a.Nc == 1) {
        if (a.defect) hipLaunchKernelGGL((k_fwd_as<XD, UD, true, false, true, float, true>), grd, blk, 0, s, a);
        else hipLaunchKernelGGL((k_fwd_as<XD, UD, false, true, true, float, true>), grd, blk, 0, s, a);
      } else if (a.as_uraw) {
        if (a.defect) hipLaunchKernelGGL((k_fwd_as<XD, UD, true, true, true, float>), grd, blk, 0, s, a);
        else hipLaunchKernelGGL((k_fwd_as<XD, UD, false, true, true, float>), grd, blk, 0, s, a);
      } else if (a.as_T && a.Nc == 1) {
        if (a.defect) hipLaunchKernelGGL((k_fwd_as<XD, UD, true, true, false, float, true>), grd, blk, 0, s, a);
        else hipLaunchKernelGGL((k_fwd_as<XD, UD, false, true, false, float, true>), grd, blk, 0, s, a);
      } else {
        if (a.defect) hipLaunchKernelGGL((k_fwd_as<XD, UD, true, true, false, float>), grd, blk, 0, s, a);
        else hipLaunchKernelGGL((k_fwd_as<XD, UD, false, true, false, float>), grd, blk, 0, s, a);
      }
      return;
    } else {
      abort();
    }
  }
  if (a.as_uraw) {
    if constexpr (cone_dims<XD, UD>()) {
      if (a.as_T && a.Nc == 1) {  // sensitivity records + elementwise update of the settled particles, with stage cones
        if (a.defect) hipLaunchKernelGGL((k_fwd_as<XD, UD, true, false, true, double, true>), grd, blk, 0, s, a);  // (one-stage ring: registers)
        else hipLaunchKernelGGL((k_fwd_as<XD, UD, false, true, true, double, true>), grd, blk, 0, s, a);
        return;
      }
      if (a.defect) hipLaunchKernelGGL((k_fwd_as<XD, UD, true, true, true>), grd, blk, 0, s, a);
      else hipLaunchKernelGGL((k_fwd_as<XD, UD, false, true, true>), grd, blk, 0, s, a);
      return;
    } else {
      abort();
    }
  }
  if (a.as_T && a.Nc == 1) {  // sensitivity records + elementwise update of the settled particles (solver.hip decides when)
    // (x12 u4, DEFECT: 128 registers + 2 spilled dwords outside the stage loop; the one-stage ring, 93 registers, measured 1 - 2 % slower)
    if (a.defect) hipLaunchKernelGGL((k_fwd_as<XD, UD, true, true, false, double, true>), grd, blk, 0, s, a);
    else hipLaunchKernelGGL((k_fwd_as<XD, UD, false, true, false, double, true>), grd, blk, 0, s, a);
    return;
  }
  if (a.M <= m2) {
    if (a.defect) hipLaunchKernelGGL((k_fwd_as<XD, UD, true, true>), grd, blk, 0, s, a);
    else hipLaunchKernelGGL((k_fwd_as<XD, UD, false, true>), grd, blk, 0, s, a);
  } else {
    if (a.defect) hipLaunchKernelGGL((k_fwd_as<XD, UD, true, false>), grd, blk, 0, s, a);
    else hipLaunchKernelGGL((k_fwd_as<XD, UD, false, false>), grd, blk, 0, s, a);
  }
}

}  // namespace

#ifdef PMPC_STAGE_TIMELINE
extern "C" int pmpc_debug_timeline_read(unsigned long long *out) {  // 3 x 128 x PMPC_TL_STAMPS stamps of the last launches
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pmpc_tl), sizeof(unsigned long long) * 3 * 128 * PMPC_TL_STAMPS);
}
#endif
bool f32_as_dims_supported(int x, int u) {
#define X(xd, ud) if (x == xd && u == ud) return f32_dims<xd, ud>();
  PMPC_FAST_DIMS(X)
#undef X
  return false;
}
bool xbox_as_dims_supported(int x, int u) {
#define X(xd, ud) if (x == xd && u == ud) return xbox_dims<xd, ud>();
  PMPC_FAST_DIMS(X)
#undef X
  return false;
}
bool cone_as_dims_supported(int x, int u) {
#define X(xd, ud) if (x == xd && u == ud) return cone_dims<xd, ud>();
  PMPC_FAST_DIMS(X)
#undef X
  return false;
}
void launch_bwd_as(const LQArgs &a, hipStream_t s) {
#define X(xd, ud) if (a.x == xd && a.u == ud) { launch_bwd_as_t<xd, ud>(a, s); return; }
  PMPC_FAST_DIMS(X)
#undef X
  abort();
}
void launch_fwd_as(const LQArgs &a, hipStream_t s) {
#define X(xd, ud) if (a.x == xd && a.u == ud) { launch_fwd_as_t<xd, ud>(a, s); return; }
  PMPC_FAST_DIMS(X)
#undef X
  abort();
}
void launch_as_ctl(AsCtl *ctl, const int *cnt_part, int M, const int *fail, int reduce, int decide, int last_of_batch, AsCtl *mirror,
                   unsigned long long *mirror_seq, unsigned long long seq, hipStream_t s, double *tail, const double *viol, const int *open_part) {
  hipLaunchKernelGGL(k_as_ctl, dim3(1), dim3(1024), 0, s, ctl, cnt_part, M, fail, reduce, decide, last_of_batch, mirror, mirror_seq, seq, tail, viol,
                     open_part);
}
void launch_as_perm(const int *settled, int M, int *perm, const int *done, hipStream_t s) {
  hipLaunchKernelGGL(k_as_perm, dim3(1), dim3(1024), 0, s, settled, M, perm, done);
}
void launch_as_begin(AsCtl *ctl, int *fail, int max_rounds, double dual_scale, hipStream_t s, int stall_limit) {
  hipLaunchKernelGGL(k_as_begin, dim3(1), dim3(64), 0, s, ctl, fail, max_rounds, dual_scale, stall_limit);
}
