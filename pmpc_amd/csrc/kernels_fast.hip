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
// operand of G = S F and as the A operand of H = F' G.  Per stage: 2*KS + 1 MFMAs
//   G = S F;   H = F' G + blkdiag(Q~_{j-1}, R~_j);   S' = H_xx - H_xu K
// and K = Huu^-1 Hux by gathering the udim control rows column-wise (4 bpermute shuffles) and an
// in-lane Cholesky substitution per column; the udim x udim Cholesky itself runs redundantly in
// every lane on readlane-broadcast values (rsq + Newton, no divides).
//
// HBM access: the kernel state index rho = g + 4r is mapped to the ORIGINAL state index
// pi(rho) = KS*g + r, so each lane reads KS CONSECUTIVE doubles of one column and the 64 lanes
// together cover the contiguous [fx_j | fu_j] (resp. Q_j) block exactly once — fully coalesced
// streaming of the (M,N,xdim,xdim) stacks; the next stage's matrices are prefetched while the
// current stage computes.  The permutation is applied wherever a global address is formed from a
// state index and nowhere else.
//
// Factor storage of this path (private to it): a.K = gains (udim x xdim, col-major),
// a.Hinv = Cholesky factor of Huu, col-major lower triangle with RECIPROCAL diagonal.
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
// 1/sqrt(d) to full double precision: v_rsq_f64 seed + two Newton steps
__device__ __forceinline__ double rsqrt_d(double d) {
  double r = __builtin_amdgcn_rsq(d);
  double h = 0.5 * d;
  r = r * (1.5 - h * r * r);
  r = r * (1.5 - h * r * r);
  return r;
}
__device__ __forceinline__ const double *badd(const double *p, long long bytes) {
  return (const double *)((const char *)p + bytes);
}
__device__ __forceinline__ double ldo(const double *base, unsigned boff) {  // uniform base + 32-bit byte offset
  return *(const double *)((const char *)base + boff);
}

// unconditional load from a per-lane VALID address, zeroed by a select (no exec-mask branch);
// `rv` guards padding rows (only when xdim is not a multiple of 4, where p[r] could leave the block)
template <bool PADX>
__device__ __forceinline__ double ldsel(const double *p, bool keep, bool rv) {
  if (PADX) return (keep && rv) ? *p : 0.0;
  const double t = *p;
  return keep ? t : 0.0;
}

template <int XD, int UD>
struct Lane {
  static constexpr int KS = (XD + 3) / 4, XP = 4 * KS;
  int c, g, oc, cb, row0;
  bool cxv, cu;
  __device__ explicit Lane(int lane) {
    c = lane & 15;
    g = lane >> 4;
    oc = (c & 3) * KS + (c >> 2);  // original state index of kernel column c
    cxv = c < XP && oc < XD;
    cb = c - XP;
    cu = cb >= 0 && cb < UD;
    row0 = KS * g;  // original index of kernel row g + 4r is row0 + r
  }
};

// s_row[r] = s_col of the lane that owns kernel column g + 4r (same k-group)
template <int KS>
__device__ __forceinline__ void col_to_row(double s_col, int g, double *s_row) {
#pragma unroll
  for (int r = 0; r < KS; r++) s_row[r] = __shfl(s_col, (g + 4 * r) + 16 * g, 64);
}

template <int UD>
__device__ __forceinline__ double pick(const double (&v)[UD], int k) {
  double o = v[0];
#pragma unroll
  for (int b = 1; b < UD; b++) o = (k == b) ? v[b] : o;
  return o;
}

// y = (L L')^-1 y with L given as strict lower part + reciprocal diagonal
template <int UD>
__device__ __forceinline__ void chol_solve(const double (&Lc)[UD][UD], const double (&Ld)[UD], double (&y)[UD]) {
#pragma unroll
  for (int p = 0; p < UD; p++) {
    double v = y[p];
#pragma unroll
    for (int k = 0; k < p; k++) v -= Lc[p][k] * y[k];
    y[p] = v * Ld[p];
  }
#pragma unroll
  for (int p = UD - 1; p >= 0; p--) {
    double v = y[p];
#pragma unroll
    for (int k = p + 1; k < UD; k++) v -= Lc[k][p] * y[k];
    y[p] = v * Ld[p];
  }
}

// ------------------------------------------------------------------------------------------------
// backward sweep (see kernels_generic.hip for the FACTOR / vector-only protocol)
//   HXB: state bounds active (Dx, wx valid)   HUB: control bounds active (Du, wu valid)
// ------------------------------------------------------------------------------------------------
template <int XD, int UD, bool FACTOR, bool HXB, bool HUB>
__global__ void __launch_bounds__(64) k_bwd_fast(LQArgs a) {
  typedef Lane<XD, UD> LT;
  constexpr int KS = LT::KS, XP = LT::XP;
  constexpr bool PADX = (XD != XP);
  const int lane = threadIdx.x;
  const LT L(lane);
  const int N = a.N, Nc = a.Nc, i = blockIdx.x, g = L.g, c = L.c;
  const size_t pbase = (size_t)i * N;
  const bool own0 = (i == 0 && a.owner);
  const bool gu = g < UD;
  const bool diag_x = L.cxv && ((c & 3) == g);  // kernel row g + 4r == c for r = c >> 2
  const int diag_r = c >> 2;

  // ---- per-lane pointers / byte offsets at stage N-1, decremented by constant strides ----------------
  // matrices: 64-bit per-lane pointers (stacks may exceed 4 GB); every lane gets a VALID address and
  // the value is zeroed by a select where the lane has no entry.
  const long long sF = -(long long)sizeof(double) * (L.cxv ? XD * XD : XD * UD);
  const double *pF = L.cxv ? a.fx + (pbase + N - 1) * (XD * XD) + XD * L.oc + L.row0
                           : a.fu + (pbase + N - 1) * (XD * UD) + XD * (L.cu ? L.cb : 0) + L.row0;
  const bool ldF = L.cxv || L.cu;
  const double *pQ = a.Q + (pbase + N - 1) * (XD * XD) + XD * (L.cxv ? L.oc : 0) + L.row0;
  const double *pR = a.R + (pbase + N - 1) * (UD * UD) + (gu ? g : 0) + UD * (L.cu ? L.cb : 0);
  double *pK = a.K + (pbase + N - 1) * (UD * XD) + (gu ? g : 0) + UD * (L.cxv ? L.oc : 0);
  double *pL = a.Hinv + (pbase + N - 1) * (UD * UD);
  // vectors: uniform base + 32-bit byte offsets (lq_fast_supported bounds the array sizes)
  unsigned ox_row = (unsigned)(((pbase + N - 1) * XD + L.row0) * sizeof(double));
  unsigned ox_col = (unsigned)(((pbase + N - 1) * XD + (L.cxv ? L.oc : 0)) * sizeof(double));
  unsigned ou_g = (unsigned)(((pbase + N - 1) * UD + (gu ? g : 0)) * sizeof(double));
  unsigned ou_c = (unsigned)(((pbase + N - 1) * UD + (L.cu ? L.cb : 0)) * sizeof(double));
  unsigned ou_0 = (unsigned)(((pbase + N - 1) * UD) * sizeof(double));
  constexpr unsigned SX = XD * sizeof(double), SU = UD * sizeof(double);

  double S[KS], s_row[KS], Qn[KS], Fn[KS];
  double s_col;

  // ---- terminal: S = Q~_{N-1} (+Dx), s = g_x,N-1 ; prefetch F_{N-1} -----------------------------------
  {
    {
      const bool ld0 = (L.cxv && N - 1 > 0) || L.cu;
#pragma unroll
      for (int r = 0; r < KS; r++) Fn[r] = ldsel<PADX>(pF + r, ld0, L.row0 + r < XD);
    }
    (void)ldF;
    double gsum = 0.0;
    if (FACTOR) {
      double part = 0.0;
#pragma unroll
      for (int r = 0; r < KS; r++) {
        const bool rv = !PADX || (L.row0 + r < XD);
        Qn[r] = ldsel<PADX>(pQ + r, L.cxv, rv);
        const unsigned o = rv ? ox_row + r * 8u : ox_row;
        const double xm = rv ? ldo(a.X, o) - ldo(a.X_ref, o) : 0.0;
        part += Qn[r] * xm;
      }
      double dd = a.reg_x;
      if (HXB) dd += ldo(a.Dx, ox_col);
#pragma unroll
      for (int r = 0; r < KS; r++) S[r] = Qn[r] + ((diag_x && diag_r == r) ? dd : 0.0);
      gsum = grp_allsum(part) + a.reg_x * (ldo(a.X, ox_col) - ldo(a.X_prev, ox_col));
    }
    if (HXB) gsum += ldo(a.wx, ox_col);
    s_col = L.cxv ? gsum : 0.0;
    col_to_row<KS>(s_col, g, s_row);
  }

  for (int j = N - 1; j >= 0; j--) {
    const bool cons = j < Nc;
    double Fr[KS];
#pragma unroll
    for (int r = 0; r < KS; r++) Fr[r] = Fn[r];

    // ---- loads of this stage: control-side vectors, R_j, and the state side of stage j-1 ------------
    double Rraw = 0.0, um_g = 0.0, ud_c = 0.0, Du_c = 0.0;
    if (FACTOR) {
      Rraw = (L.cu && gu) ? *pR : 0.0;
      um_g = gu ? ldo(a.U, ou_g) - ldo(a.U_ref, ou_g) : 0.0;
      ud_c = a.reg_u * (ldo(a.U, ou_c) - ldo(a.U_prev, ou_c));
      if (HUB) Du_c = ldo(a.Du, ou_c);
    }
    if (HUB) ud_c += ((!cons || own0) ? ldo(a.wu, ou_c) : 0.0);
    // prefetch next stage's dynamics (F_{j-1}; its state columns are zero at stage 0)
    if (j > 0) {
      pF = badd(pF, sF);
      const bool ldn = (L.cxv && j - 1 > 0) || L.cu;
#pragma unroll
      for (int r = 0; r < KS; r++) Fn[r] = ldsel<PADX>(pF + r, ldn, L.row0 + r < XD);
    }
    double xm_row[KS], xd_c = 0.0, Dx_c = 0.0;
    if (j > 0) {
      ox_row -= SX;
      ox_col -= SX;
      if (FACTOR) {
        pQ -= XD * XD;
#pragma unroll
        for (int r = 0; r < KS; r++) {
          const bool rv = !PADX || (L.row0 + r < XD);
          Qn[r] = ldsel<PADX>(pQ + r, L.cxv, rv);
          const unsigned o = rv ? ox_row + r * 8u : ox_row;
          xm_row[r] = rv ? ldo(a.X, o) - ldo(a.X_ref, o) : 0.0;
        }
        xd_c = a.reg_x * (ldo(a.X, ox_col) - ldo(a.X_prev, ox_col));
        if (HXB) Dx_c = ldo(a.Dx, ox_col);
      }
      if (HXB) xd_c += ldo(a.wx, ox_col);
    }

    // ---- h = F' s (+ control gradient) -----------------------------------------------------------------
    double hp = Rraw * um_g;
#pragma unroll
    for (int r = 0; r < KS; r++) hp += Fr[r] * s_row[r];
    double h_col = grp_allsum(hp);
    if (L.cu) h_col += ud_c;
    double hu[UD];
#pragma unroll
    for (int b = 0; b < UD; b++) hu[b] = readlane_d(h_col, XP + b);

    double Kreg = 0.0, Lc[UD][UD], Ld[UD];
    v4d H = {0.0, 0.0, 0.0, 0.0};
    if (FACTOR) {
      // ---- H = F' S F + blkdiag(Q~_{j-1}, R~_j) ------------------------------------------------------
      if (j > 0) {
        const double dd = a.reg_x + Dx_c;
#pragma unroll
        for (int r = 0; r < KS; r++) H[r] = Qn[r] + ((diag_x && diag_r == r) ? dd : 0.0);
      }
      H[KS] = Rraw + ((L.cu && g == L.cb) ? a.reg_u + (cons ? 0.0 : Du_c) : 0.0);
      v4d G = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int r = 0; r < KS; r++) G = mfma(S[r], Fr[r], G);
#pragma unroll
      for (int r = 0; r < KS; r++) H = mfma(Fr[r], G[r], H);
    }

    if (cons) {
      // ---- consensus stage (Nc == 1, j == 0): export the per-particle condensed (H_i, g_i) ----------
      if (FACTOR) {
        // Hc_part[i][p + UD q] = Huu[p][q] lives in lane (XP + q, p), register KS
        double v = H[KS];
        if (own0 && HUB && L.cu && g == L.cb) v += Du_c;
        if (L.cu && gu) a.Hc_part[(size_t)i * (UD * UD) + g + UD * L.cb] = v;
      }
      if (lane < UD) a.gc_part[(size_t)i * UD + lane] = pick<UD>(hu, lane);
      break;
    }

    if (FACTOR) {
      // ---- Cholesky of Huu on lane-uniform values (readlane broadcast of the lower triangle) --------
      bool bad = false;
#pragma unroll
      for (int q = 0; q < UD; q++) {
#pragma unroll
        for (int p = q; p < UD; p++) {
          double v = readlane_d(H[KS], (XP + q) + 16 * p);  // Huu[p][q]
#pragma unroll
          for (int k = 0; k < q; k++) v -= Lc[p][k] * Lc[q][k];
          if (p == q) {
            bad |= !(v > 0.0);
            Ld[q] = rsqrt_d(v);
          } else {
            Lc[p][q] = v * Ld[q];
          }
        }
      }
      if (bad && lane == 0) *a.fail = 2;
      // ---- K = Huu^-1 Hux: gather the control rows column-wise, substitute in-lane ------------------
      double col[UD];
#pragma unroll
      for (int k = 0; k < UD; k++) col[k] = __shfl(H[KS], c + 16 * k, 64);  // H[XP + k][c]
      chol_solve<UD>(Lc, Ld, col);
      const double Kg = pick<UD>(col, g);
      Kreg = (L.cxv && gu) ? Kg : 0.0;
      v4d Sn = mfma(H[KS], gu ? -Kg : 0.0, H);  // S' = Hxx - Hxu K (columns >= XP are never used)
#pragma unroll
      for (int r = 0; r < KS; r++) S[r] = Sn[r];
      if (L.cxv && gu) *pK = Kreg;
      if (lane == 0) {
#pragma unroll
        for (int q = 0; q < UD; q++)
#pragma unroll
          for (int p = q; p < UD; p++) pL[p + UD * q] = (p == q) ? Ld[q] : Lc[p][q];
      }
    } else {
      Kreg = (L.cxv && gu) ? *pK : 0.0;
#pragma unroll
      for (int q = 0; q < UD; q++)
#pragma unroll
        for (int p = q; p < UD; p++) {
          const double v = pL[p + UD * q];
          if (p == q) Ld[q] = v;
          else Lc[p][q] = v;
        }
    }

    // ---- feed-forward k = Huu^-1 hu and s_{j-1} = h_x - K' hu + g_x,j-1 -----------------------------
    const double hug = pick<UD>(hu, g);
    chol_solve<UD>(Lc, Ld, hu);
    if (lane == 0) {
#pragma unroll
      for (int b = 0; b < UD; b++) *(double *)((char *)a.kff + ou_0 + b * 8u) = hu[b];
    }
    if (j == 0) break;
    double p2 = -Kreg * hug;
    if (FACTOR) {
#pragma unroll
      for (int r = 0; r < KS; r++) p2 += Qn[r] * xm_row[r];
    }
    const double red2 = grp_allsum(p2);
    s_col = L.cxv ? h_col + red2 + xd_c : 0.0;
    col_to_row<KS>(s_col, g, s_row);
    pR -= UD * UD;
    pK -= UD * XD;
    pL -= UD * UD;
    ou_g -= SU;
    ou_c -= SU;
    ou_0 -= SU;
  }
}

// ------------------------------------------------------------------------------------------------
// forward sweep (ROLLOUT: absolute linear rollout X from U, PMPC.jl/src/types.jl:161-173)
// ------------------------------------------------------------------------------------------------
template <int XD, int UD, bool ROLLOUT>
__global__ void __launch_bounds__(64) k_fwd_fast(LQArgs a, const double *Uin, double *Xout) {
  typedef Lane<XD, UD> LT;
  constexpr int KS = LT::KS;
  constexpr bool PADX = (XD != LT::XP);
  const int lane = threadIdx.x;
  const LT L(lane);
  const int N = a.N, Nc = a.Nc, i = blockIdx.x, g = L.g, c = L.c;
  const size_t pbase = (size_t)i * N;
  const bool gu = g < UD;
  const long long sF = (long long)sizeof(double) * (L.cxv ? XD * XD : XD * UD);
  const double *pF = L.cxv ? a.fx + pbase * (XD * XD) + XD * L.oc + L.row0
                           : a.fu + pbase * (XD * UD) + XD * (L.cu ? L.cb : 0) + L.row0;
  const bool ldF = L.cxv || L.cu;
  const double *pK = a.K + pbase * (UD * XD) + (gu ? g : 0) + UD * (L.cxv ? L.oc : 0);
  unsigned ox_row = (unsigned)((pbase * XD + L.row0) * sizeof(double));
  unsigned ou_g = (unsigned)((pbase * UD + (gu ? g : 0)) * sizeof(double));
  unsigned ou_c = (unsigned)((pbase * UD + (L.cu ? L.cb : 0)) * sizeof(double));
  unsigned ou_0 = (unsigned)((pbase * UD) * sizeof(double));
  constexpr unsigned SX = XD * sizeof(double), SU = UD * sizeof(double);

  double xcol = 0.0;  // dx[oc] on valid state columns (ROLLOUT: X_{j-1} - X_prev_{j-1})
  double Fn[KS], Kn = 0.0, kn = 0.0;  // next stage's F, gains and feed-forward (prefetched one stage ahead)
#pragma unroll
  for (int r = 0; r < KS; r++) Fn[r] = ldsel<PADX>(pF + r, L.cu, L.row0 + r < XD);  // stage 0: state columns are zero
  if (!ROLLOUT && Nc == 0) {
    Kn = (L.cxv && gu) ? *pK : 0.0;
    kn = gu ? ldo(a.kff, ou_g) : 0.0;
  }
  for (int j = 0; j < N; j++) {
    double Fr[KS];
#pragma unroll
    for (int r = 0; r < KS; r++) Fr[r] = Fn[r];
    const double Kreg = Kn, kreg = kn;
    if (j + 1 < N) {
      pF = badd(pF, sF);
#pragma unroll
      for (int r = 0; r < KS; r++) Fn[r] = ldsel<PADX>(pF + r, ldF, L.row0 + r < XD);
      if (!ROLLOUT) {
        pK += UD * XD;
        Kn = (L.cxv && gu && j + 1 >= Nc) ? *pK : 0.0;
        kn = (gu && j + 1 >= Nc) ? ldo(a.kff, ou_g + SU) : 0.0;
      }
    }
    double ycol;
    double du[UD];
    if (ROLLOUT) {
      ycol = L.cxv ? xcol : (L.cu ? ldo(Uin, ou_c) - ldo(a.U_prev, ou_c) : 0.0);
    } else {
      if (j < Nc) {
#pragma unroll
        for (int b = 0; b < UD; b++) du[b] = a.duc[j * UD + b];
      } else {
        const double sum = row_allsum(Kreg * xcol);
        const double dug = gu ? -sum - kreg : 0.0;
#pragma unroll
        for (int b = 0; b < UD; b++) du[b] = readlane_d(dug, 16 * b);
      }
      ycol = L.cxv ? xcol : (L.cu ? pick<UD>(du, L.cb) : 0.0);
    }
    double xr[KS];
#pragma unroll
    for (int r = 0; r < KS; r++) xr[r] = row_allsum(Fr[r] * ycol);
    if (ROLLOUT) {
      // X_j = f_j + fx (X_{j-1} - Xp_{j-1}) + fu (U_j - Up_j); next xcol needs X_j - X_prev_j
#pragma unroll
      for (int r = 0; r < KS; r++) {
        const bool rv = L.row0 + r < XD;
        const unsigned o = rv ? ox_row + r * 8u : ox_row;
        const double xj = xr[r] + ldo(a.f, o);
        if (c == 0 && rv) *(double *)((char *)Xout + o) = xj;
        xr[r] = xj - ldo(a.X_prev, o);
      }
    } else {
      if (c == 0) {
#pragma unroll
        for (int r = 0; r < KS; r++)
          if (L.row0 + r < XD) {
            double *o = (double *)((char *)a.dX + ox_row + r * 8u);
            *o = a.accumulate ? *o + xr[r] : xr[r];
          }
      }
      if (lane == 0) {
#pragma unroll
        for (int b = 0; b < UD; b++) {
          double *o = (double *)((char *)a.dU + ou_0 + b * 8u);
          *o = a.accumulate ? *o + du[b] : du[b];
        }
      }
    }
    // next column-distributed state: kernel column c lives in k-group c & 3, register c >> 2
    double nx = 0.0;
#pragma unroll
    for (int r = 0; r < KS; r++) {
      const double t = __shfl(xr[r], 16 * (c & 3), 64);
      nx = ((c >> 2) == r) ? t : nx;
    }
    xcol = L.cxv ? nx : 0.0;
    ox_row += SX;
    ou_g += SU;
    ou_c += SU;
    ou_0 += SU;
  }
}

template <int XD, int UD>
void launch_bwd_t(const LQArgs &a, bool factor, hipStream_t s) {
  const bool xb = a.wx != nullptr, ub = a.wu != nullptr;
  const dim3 grd(a.M), blk(64);
#define PMPC_BWD(F, XB, UB) hipLaunchKernelGGL((k_bwd_fast<XD, UD, F, XB, UB>), grd, blk, 0, s, a)
  if (factor) {
    if (xb && ub) PMPC_BWD(true, true, true);
    else if (xb) PMPC_BWD(true, true, false);
    else if (ub) PMPC_BWD(true, false, true);
    else PMPC_BWD(true, false, false);
  } else {
    if (xb && ub) PMPC_BWD(false, true, true);
    else if (xb) PMPC_BWD(false, true, false);
    else if (ub) PMPC_BWD(false, false, true);
    else PMPC_BWD(false, false, false);
  }
#undef PMPC_BWD
}
template <int XD, int UD>
void launch_fwd_t(const LQArgs &a, hipStream_t s) {
  hipLaunchKernelGGL((k_fwd_fast<XD, UD, false>), dim3(a.M), dim3(64), 0, s, a, (const double *)nullptr, (double *)nullptr);
}
template <int XD, int UD>
void launch_rollout_t(const LQArgs &a, const double *U, double *X, hipStream_t s) {
  hipLaunchKernelGGL((k_fwd_fast<XD, UD, true>), dim3(a.M), dim3(64), 0, s, a, U, X);
}

}  // namespace

// (xdim, udim) pairs with compiled instances
#define PMPC_FAST_DIMS(X) X(12, 4) X(4, 2) X(2, 1) X(3, 2) X(5, 3) X(6, 2) X(8, 4)

bool lq_fast_supported(const LQArgs &a) {
  if (a.w != 0 || a.any_slew || a.Nc > 1 || !a.sym_cost) return false;
  // vector arrays are addressed with 32-bit byte offsets
  if ((size_t)a.M * a.N * (size_t)(a.x > a.u ? a.x : a.u) * sizeof(double) >= (1ull << 31)) return false;
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

void launch_rollout_fast(const LQArgs &a, const double *U, double *X, hipStream_t s) {
#define X(xd, ud) if (a.x == xd && a.u == ud) { launch_rollout_t<xd, ud>(a, U, X, s); return; }
  PMPC_FAST_DIMS(X)
#undef X
  abort();
}
