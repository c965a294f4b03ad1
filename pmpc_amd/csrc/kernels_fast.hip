// kernels_fast.hip — register-resident MFMA path of the structured LQ solve (gfx950 / CDNA4).
//
// Scope: no slew penalties, consensus horizon Nc <= 1, symmetric cost blocks, and
// XP + udim <= 16 with udim <= 4, where XP = 4*ceil(xdim/4).  That covers the benchmark
// configs (unicycle x4 u2, quadrotor x12 u4 — for the quadrotor [fx | fu] is exactly one
// 12 x 16 tile and the stage Hessian F'SF exactly one 16 x 16 fp64 MFMA tile).
//
// One 64-lane wavefront owns one particle and walks its horizon; lane = (c, g), c = lane & 15 a
// tile COLUMN, g = lane >> 4 a k-group.  Everything lives in registers in the
// v_mfma_f64_16x16x4_f64 layouts:
//   C/D layout  lane (c,g), reg r  <->  M[g + 4r][c]
//   A operand   lane (i,g), step r <->  A[i][k = g + 4r]        B operand  lane (j,g) <-> B[k = g + 4r][j]
// so a C-layout tile is directly the B operand of the next product, a SYMMETRIC C-layout tile is
// directly an A operand, and F = [fx | fu] loaded once as Fr[r] = F[g + 4r][c] serves both as the B
// operand of G = S F and as the A operand of H = F' G.  Per stage: 2*KS + 2 MFMAs
//   G = S F;  H = F' G + blkdiag(Q~_{j-1}, R~_j);  K = Huu^-1 Hux;  S' = H_xx - H_xu K
// plus a 4x4 Cholesky done redundantly per lane on readlane-broadcast values.
//
// HBM access: the kernel state index rho = g + 4r is mapped to the ORIGINAL state index
// pi(rho) = KS*g + r, so each lane reads KS CONSECUTIVE doubles of one column and the 64 lanes
// together cover the contiguous [fx_j | fu_j] (resp. Q_j) block exactly once — fully coalesced
// streaming of the (M,N,xdim,xdim) stacks.  The permutation is applied wherever a global address
// is formed from a state index and nowhere else.
//
// Reference semantics: same Newton system as kernels_generic.hip (PMPC.jl/src/lqp_utils.jl:2-393).
#include "pmpc_dev.h"

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v4d mfma(double a, double b, v4d c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ double readlane_d(double v, int l) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
  int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
  int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
// sum over the 16 lanes of a DPP row (all lanes get the total): xor1, xor2, half-mirror, mirror
__device__ __forceinline__ double row_allsum(double v) {
  v += dpp_d<0xB1>(v);
  v += dpp_d<0x4E>(v);
  v += dpp_d<0x141>(v);
  v += dpp_d<0x140>(v);
  return v;
}
// sum over the 4 k-groups (lanes c, c+16, c+32, c+48), all lanes get the total
__device__ __forceinline__ double grp_allsum(double v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

template <int XD, int UD>
struct Lane {
  static constexpr int KS = (XD + 3) / 4, XP = 4 * KS;
  int i, c, g, oc, cb, row0;
  bool cxv, cu;
  __device__ explicit Lane(int lane) {
    i = blockIdx.x;
    c = lane & 15;
    g = lane >> 4;
    oc = (c & 3) * KS + (c >> 2);  // original state index of kernel column c
    cxv = c < XP && oc < XD;
    cb = c - XP;
    cu = cb >= 0 && cb < UD;
    row0 = KS * g;  // original index of kernel row g + 4r is row0 + r
  }
};

// Fr[r] = F[g + 4r][c], F = [A~ | B~] with A~ = fx_j (0 at stage 0), B~ = fu_j
template <int XD, int UD>
__device__ __forceinline__ void load_F(const LQArgs &a, const Lane<XD, UD> &L, int j, double *Fr) {
  constexpr int KS = Lane<XD, UD>::KS;
  const size_t blk = (size_t)L.i * a.N + j;
  const double *p = L.cxv ? a.fx + blk * (XD * XD) + XD * L.oc : a.fu + blk * (XD * UD) + XD * (L.cu ? L.cb : 0);
  const bool ld = (L.cxv && j > 0) || L.cu;
#pragma unroll
  for (int r = 0; r < KS; r++) Fr[r] = (ld && L.row0 + r < XD) ? p[L.row0 + r] : 0.0;
}

// C-layout registers of the (symmetric) state cost block Q_j
template <int XD, int UD>
__device__ __forceinline__ void load_Q(const LQArgs &a, const Lane<XD, UD> &L, int j, double *Qr) {
  constexpr int KS = Lane<XD, UD>::KS;
  const double *p = a.Q + ((size_t)L.i * a.N + j) * (XD * XD) + XD * (L.cxv ? L.oc : 0);
#pragma unroll
  for (int r = 0; r < KS; r++) Qr[r] = (L.cxv && L.row0 + r < XD) ? p[L.row0 + r] : 0.0;
}

// s_row[r] = s_col of the lane that owns kernel column g + 4r (same k-group)
template <int KS>
__device__ __forceinline__ void col_to_row(double s_col, int g, double *s_row) {
#pragma unroll
  for (int r = 0; r < KS; r++) s_row[r] = __shfl(s_col, (g + 4 * r) + 16 * g, 64);
}

template <int UD>
__device__ __forceinline__ double pick(const double (&v)[UD], int k) {
  double o = 0.0;
#pragma unroll
  for (int b = 0; b < UD; b++) o = (k == b) ? v[b] : o;
  return o;
}

// ------------------------------------------------------------------------------------------------
// backward sweep (see kernels_generic.hip for the FACTOR / vector-only protocol)
// ------------------------------------------------------------------------------------------------
template <int XD, int UD, bool FACTOR>
__global__ void __launch_bounds__(64) k_bwd_fast(LQArgs a) {
  typedef Lane<XD, UD> LT;
  constexpr int KS = LT::KS, XP = LT::XP;
  const int lane = threadIdx.x;
  const LT L(lane);
  const int N = a.N, Nc = a.Nc, i = L.i, g = L.g, c = L.c;
  const size_t pbase = (size_t)i * N;
  const bool own0 = (i == 0 && a.owner);

  double S[KS], s_row[KS], Qn[KS];
  double s_col;

  // ---- terminal: S = Q~_{N-1} (+Dx), s = g_x,N-1 ----------------------------------------------------
  {
    const int jj = N - 1;
    double gsum = 0.0;
    if (FACTOR) {
      load_Q<XD, UD>(a, L, jj, Qn);
      const double *X = a.X + (pbase + jj) * XD, *Xr = a.X_ref + (pbase + jj) * XD, *Xp = a.X_prev + (pbase + jj) * XD;
      double part = 0.0;
#pragma unroll
      for (int r = 0; r < KS; r++) {
        const int ro = L.row0 + r;
        const double xm = ro < XD ? X[ro] - Xr[ro] : 0.0;
        part += Qn[r] * xm;
        double d = 0.0;
        if (L.cxv && g + 4 * r == c) d = a.reg_x + (a.Dx ? a.Dx[(pbase + jj) * XD + L.oc] : 0.0);
        S[r] = Qn[r] + d;
      }
      gsum = grp_allsum(part);
      if (L.cxv) gsum += a.reg_x * (X[L.oc] - Xp[L.oc]);
    }
    if (L.cxv && a.wx) gsum += a.wx[(pbase + jj) * XD + L.oc];
    s_col = L.cxv ? gsum : 0.0;
    col_to_row<KS>(s_col, g, s_row);
  }

  for (int j = N - 1; j >= 0; j--) {
    const bool cons = j < Nc;
    double Fr[KS];
    load_F<XD, UD>(a, L, j, Fr);
    const double *Uj = a.U + (pbase + j) * UD;

    // ---- control-side gradient pieces -------------------------------------------------------------
    double Rraw = 0.0, hp = 0.0;
#pragma unroll
    for (int r = 0; r < KS; r++) hp += Fr[r] * s_row[r];
    if (FACTOR) {
      if (L.cu && g < UD) {
        Rraw = a.R[(pbase + j) * (UD * UD) + g + UD * L.cb];
        hp += Rraw * (Uj[g] - a.U_ref[(pbase + j) * UD + g]);
      }
    }
    double h_col = grp_allsum(hp);
    if (L.cu) {
      if (FACTOR) h_col += a.reg_u * (Uj[L.cb] - a.U_prev[(pbase + j) * UD + L.cb]);
      if (a.wu && (!cons || own0)) h_col += a.wu[(pbase + j) * UD + L.cb];
    }
    double hu[UD];
#pragma unroll
    for (int b = 0; b < UD; b++) hu[b] = readlane_d(h_col, XP + b);

    double Kreg = 0.0, Hinv[UD][UD];
    v4d H = {0.0, 0.0, 0.0, 0.0};
    if (FACTOR) {
      // ---- H = F' S F + blkdiag(Q~_{j-1}, R~_j) --------------------------------------------------
      if (j > 0) load_Q<XD, UD>(a, L, j - 1, Qn);
#pragma unroll
      for (int r = 0; r < KS; r++) {
        double v = 0.0;
        if (j > 0) {
          v = Qn[r];
          if (L.cxv && g + 4 * r == c) v += a.reg_x + (a.Dx ? a.Dx[(pbase + j - 1) * XD + L.oc] : 0.0);
        }
        H[r] = v;
      }
      {
        double v = Rraw;
        if (L.cu && g == L.cb) {
          v += a.reg_u;
          if (a.Du && !cons) v += a.Du[(pbase + j) * UD + L.cb];
        }
        H[KS] = v;
      }
      v4d G = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int r = 0; r < KS; r++) G = mfma(S[r], Fr[r], G);
#pragma unroll
      for (int r = 0; r < KS; r++) H = mfma(Fr[r], G[r], H);
    }

    if (cons) {
      // ---- consensus stage (Nc == 1, j == 0): export the per-particle condensed (H_i, g_i) -------
      if (FACTOR) {
        double Huu[UD][UD];
#pragma unroll
        for (int p = 0; p < UD; p++)
#pragma unroll
          for (int q = 0; q < UD; q++) Huu[p][q] = readlane_d(H[KS], (XP + q) + 16 * p);
        if (lane < UD * UD) {
          double v = 0.0;
#pragma unroll
          for (int p = 0; p < UD; p++)
#pragma unroll
            for (int q = 0; q < UD; q++) v = (lane == p + UD * q) ? Huu[p][q] : v;
          if (own0 && a.Du && (lane % UD) == (lane / UD)) v += a.Du[(pbase + j) * UD + lane % UD];
          a.Hc_part[(size_t)i * (UD * UD) + lane] = v;
        }
      }
      if (lane < UD) a.gc_part[(size_t)i * UD + lane] = pick<UD>(hu, lane);
      break;
    }

    if (FACTOR) {
      // ---- Huu^-1 by Cholesky on lane-uniform values ---------------------------------------------
      double Lc[UD][UD], Li[UD][UD];
#pragma unroll
      for (int p = 0; p < UD; p++)
#pragma unroll
        for (int q = 0; q <= p; q++) Lc[p][q] = readlane_d(H[KS], (XP + q) + 16 * p);
      bool bad = false;
#pragma unroll
      for (int q = 0; q < UD; q++) {
        double d = Lc[q][q];
#pragma unroll
        for (int k = 0; k < q; k++) d -= Lc[q][k] * Lc[q][k];
        bad |= !(d > 0.0);
        d = sqrt(d);
        const double inv = 1.0 / d;
        Lc[q][q] = d;
        Li[q][q] = inv;
#pragma unroll
        for (int p = q + 1; p < UD; p++) {
          double v = Lc[p][q];
#pragma unroll
          for (int k = 0; k < q; k++) v -= Lc[p][k] * Lc[q][k];
          Lc[p][q] = v * inv;
        }
      }
      if (bad && lane == 0) *a.fail = 2;
#pragma unroll
      for (int q = 0; q < UD; q++)
#pragma unroll
        for (int p = q + 1; p < UD; p++) {
          double v = 0.0;
#pragma unroll
          for (int k = q; k < p; k++) v += Lc[p][k] * Li[k][q];
          Li[p][q] = -v * Li[p][p];
        }
#pragma unroll
      for (int p = 0; p < UD; p++)
#pragma unroll
        for (int q = 0; q <= p; q++) {
          double v = 0.0;
#pragma unroll
          for (int k = p; k < UD; k++) v += Li[k][p] * Li[k][q];
          Hinv[p][q] = v;
          Hinv[q][p] = v;
        }
      // ---- K = Huu^-1 Hux, S' = Hxx - Hxu K -------------------------------------------------------
      double av = 0.0;
#pragma unroll
      for (int p = 0; p < UD; p++)
#pragma unroll
        for (int q = 0; q < UD; q++) av = (c == p && g == q) ? Hinv[p][q] : av;
      v4d Kacc = {0.0, 0.0, 0.0, 0.0};
      Kacc = mfma(av, H[KS], Kacc);
      Kreg = (L.cxv && g < UD) ? Kacc[0] : 0.0;
      v4d Sn = mfma(H[KS], -Kacc[0], H);
#pragma unroll
      for (int r = 0; r < KS; r++) S[r] = Sn[r];
      if (L.cxv && g < UD) a.K[(pbase + j) * (UD * XD) + g + UD * L.oc] = Kreg;
      if (lane < UD * UD) {
        double v = 0.0;
#pragma unroll
        for (int p = 0; p < UD; p++)
#pragma unroll
          for (int q = 0; q < UD; q++) v = (lane == p + UD * q) ? Hinv[p][q] : v;
        a.Hinv[(pbase + j) * (UD * UD) + lane] = v;
      }
    } else {
      if (L.cxv && g < UD) Kreg = a.K[(pbase + j) * (UD * XD) + g + UD * L.oc];
      const double *Hg = a.Hinv + (pbase + j) * (UD * UD);
#pragma unroll
      for (int p = 0; p < UD; p++)
#pragma unroll
        for (int q = 0; q < UD; q++) Hinv[p][q] = Hg[p + UD * q];
    }

    // ---- feed-forward k = Huu^-1 hu and s_{j-1} = h_x - K' hu + g_x,j-1 ---------------------------
    if (lane < UD) {
      double kv[UD];
#pragma unroll
      for (int p = 0; p < UD; p++) {
        double v = 0.0;
#pragma unroll
        for (int q = 0; q < UD; q++) v += Hinv[p][q] * hu[q];
        kv[p] = v;
      }
      a.kff[(pbase + j) * UD + lane] = pick<UD>(kv, lane);
    }
    if (j == 0) break;
    double p2 = -Kreg * pick<UD>(hu, g);
    double add = 0.0;
    if (FACTOR) {
      const double *X = a.X + (pbase + j - 1) * XD, *Xr = a.X_ref + (pbase + j - 1) * XD, *Xp = a.X_prev + (pbase + j - 1) * XD;
#pragma unroll
      for (int r = 0; r < KS; r++) {
        const int ro = L.row0 + r;
        const double xm = ro < XD ? X[ro] - Xr[ro] : 0.0;
        p2 += Qn[r] * xm;
      }
      if (L.cxv) add = a.reg_x * (X[L.oc] - Xp[L.oc]);
    }
    if (L.cxv && a.wx) add += a.wx[(pbase + j - 1) * XD + L.oc];
    const double red2 = grp_allsum(p2);
    s_col = L.cxv ? h_col + red2 + add : 0.0;
    col_to_row<KS>(s_col, g, s_row);
  }
}

// ------------------------------------------------------------------------------------------------
// forward sweep
// ------------------------------------------------------------------------------------------------
template <int XD, int UD>
__global__ void __launch_bounds__(64) k_fwd_fast(LQArgs a) {
  typedef Lane<XD, UD> LT;
  constexpr int KS = LT::KS;
  const int lane = threadIdx.x;
  const LT L(lane);
  const int N = a.N, Nc = a.Nc, i = L.i, g = L.g, c = L.c;
  const size_t pbase = (size_t)i * N;
  double xcol = 0.0;  // dx[oc] on valid state columns
  for (int j = 0; j < N; j++) {
    double Fr[KS];
    load_F<XD, UD>(a, L, j, Fr);
    double du[UD];
    if (j < Nc) {
#pragma unroll
      for (int b = 0; b < UD; b++) du[b] = a.duc[j * UD + b];
    } else {
      double Kreg = 0.0;
      if (L.cxv && g < UD) Kreg = a.K[(pbase + j) * (UD * XD) + g + UD * L.oc];
      const double sum = row_allsum(Kreg * xcol);
      const double dug = g < UD ? -sum - a.kff[(pbase + j) * UD + g] : 0.0;
#pragma unroll
      for (int b = 0; b < UD; b++) du[b] = readlane_d(dug, 16 * b);
    }
    const double ycol = L.cxv ? xcol : (L.cu ? pick<UD>(du, L.cb) : 0.0);
    double xr[KS];
#pragma unroll
    for (int r = 0; r < KS; r++) xr[r] = row_allsum(Fr[r] * ycol);
    if (c == 0) {
#pragma unroll
      for (int r = 0; r < KS; r++)
        if (L.row0 + r < XD) {
          double *o = a.dX + (pbase + j) * XD + L.row0 + r;
          *o = a.accumulate ? *o + xr[r] : xr[r];
        }
    }
    if (lane < UD) {
      double *o = a.dU + (pbase + j) * UD + lane;
      const double v = pick<UD>(du, lane);
      *o = a.accumulate ? *o + v : v;
    }
    // next column-distributed state: kernel column c lives in k-group c & 3, register c >> 2
    double nx = 0.0;
#pragma unroll
    for (int r = 0; r < KS; r++) {
      const double t = __shfl(xr[r], 16 * (c & 3), 64);
      nx = ((c >> 2) == r) ? t : nx;
    }
    xcol = L.cxv ? nx : 0.0;
  }
}

template <int XD, int UD>
void launch_bwd_t(const LQArgs &a, bool factor, hipStream_t s) {
  if (factor) hipLaunchKernelGGL((k_bwd_fast<XD, UD, true>), dim3(a.M), dim3(64), 0, s, a);
  else hipLaunchKernelGGL((k_bwd_fast<XD, UD, false>), dim3(a.M), dim3(64), 0, s, a);
}
template <int XD, int UD>
void launch_fwd_t(const LQArgs &a, hipStream_t s) {
  hipLaunchKernelGGL((k_fwd_fast<XD, UD>), dim3(a.M), dim3(64), 0, s, a);
}

}  // namespace

// (xdim, udim) pairs with compiled instances
#define PMPC_FAST_DIMS(X) X(12, 4) X(4, 2) X(2, 1) X(3, 2) X(5, 3) X(6, 2) X(8, 4) X(4, 4) X(6, 3) X(10, 4) X(8, 2) X(4, 1) X(3, 1)

bool lq_fast_supported(const LQArgs &a) {
  if (a.w != 0 || a.any_slew || a.Nc > 1 || !a.sym_cost) return false;
#define X(xd, ud) if (a.x == xd && a.u == ud) return true;
  PMPC_FAST_DIMS(X)
#undef X
  return false;
}

void launch_bwd_fast(const LQArgs &a, bool factor, hipStream_t s) {
#define X(xd, ud) if (a.x == xd && a.u == ud) { launch_bwd_t<xd, ud>(a, factor, s); return; }
  PMPC_FAST_DIMS(X)
#undef X
  abort();
}

void launch_fwd_fast(const LQArgs &a, hipStream_t s) {
#define X(xd, ud) if (a.x == xd && a.u == ud) { launch_fwd_t<xd, ud>(a, s); return; }
  PMPC_FAST_DIMS(X)
#undef X
  abort();
}
