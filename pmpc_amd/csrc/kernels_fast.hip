// kernels_fast.hip — register-resident MFMA path of the structured LQ solve (gfx950 / CDNA4).
//
// Scope: no slew penalties, symmetric cost blocks, any consensus horizon (Nc > 1: k_cond_fast), and
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
// Factor storage of this path (private to it): ONE 64-double record per (particle, stage) in a.K, indexed by
// the lane that owns the value — record[c + 16 g] = K[g][pi(c)] for state columns c, Huu^-1[g][c - XP] for
// control columns — so the factor sweep writes it with a single fully coalesced 512-byte store and the
// vector / forward sweeps read gains and Huu^-1 back with a single coalesced load (per-lane scattered
// 8/16-byte stores of K, chol(Huu), k cost 25 % of the factor sweep before).  a.kff = feed-forward k.
//
// Reference semantics: same Newton system as kernels_generic.hip (PMPC.jl/src/lqp_utils.jl:2-393).
#include "fast_common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// backward sweep (see kernels_generic.hip for the FACTOR / vector-only protocol)
//   HXB: state bounds active (Dx, wx valid)   HUB: control bounds active (Du, wu valid)
// FACTOR reads the gradient pre-pass arrays (xm, xd, um, ud; launch_grad_prep), the vector-only sweep
// reads the IPM shifts (wx, wu) directly.  Everything a stage needs EARLY (F_j, R_j, um_j, ud_j, Du_j,
// Q_{j-1}, Dx_{j-1} resp. K_j, chol(Huu_j)) is loaded one full stage ahead; what it needs LATE
// (xm_{j-1}, xd_{j-1}) is issued at the top of the stage.  Lanes without an entry in an array read a
// zero buffer through a zero-stride pointer, so no load is predicated and no value needs a select.
// ------------------------------------------------------------------------------------------------
// DEEP: also the stage's mid/late data (Q_{j-1}, xm, gx, D) is loaded a full stage ahead — lowest per-stage
// latency (few particles per GPU) at 144 VGPRs / 3 waves per SIMD; !DEEP issues those at the top of their own stage
// and fits 4 waves per SIMD (128 VGPRs), which wins once there are > 3 waves per SIMD to run.
// (the sweeps of the active-set rounds — no gradient pre-pass, settled particles skipped, dynamics defect of a no-rollout warm
// start — are kernels of their own: kernels_as.hip)
template <int XD, int UD, bool FACTOR, bool HXB, bool HUB, bool DEEP>
__global__ void __launch_bounds__(64) k_bwd_fast(LQArgs a) {
  typedef Lane<XD, UD> LT;
  constexpr int KS = LT::KS, XP = LT::XP;
  constexpr bool PADX = (XD != XP);
  constexpr long long D8 = sizeof(double);
  const int lane = threadIdx.x;
  const LT L(lane);
  const int N = a.N, Nc = a.Nc, i = blockIdx.x, g = L.g, c = L.c;
  const size_t pbase = (size_t)i * N;
  const bool own0 = (i == 0 && a.owner);
  const bool gu = g < UD;
  const double *Z = a.zeros;
  const double pwt = a.pw ? a.pw[i] : 1.0;  // cost weight of this particle (gradient arrays come pre-weighted)
  const double regx = pwt * a.reg_x, regu = pwt * a.reg_u;

  // per-lane pointers at stage N-1 and per-lane byte strides (0 for lanes that read the zero buffer)
  const bool fF = L.cxv || L.cu;
  const double *pF = L.cxv ? a.fx + (pbase + N - 1) * (XD * XD) + XD * L.oc + L.row0
                           : (L.cu ? a.fu + (pbase + N - 1) * (XD * UD) + XD * L.cb + L.row0 : Z);
  const int sF = fF ? -(int)D8 * (L.cxv ? XD * XD : XD * UD) : 0;
  const double *pQ = L.cxv ? a.Q + (pbase + N - 1) * (XD * XD) + XD * L.oc + L.row0 : Z;
  const int sQ = L.cxv ? -(int)D8 * (XD * XD) : 0;
  const bool fR = L.cu && gu;
  const double *pR = fR ? a.R + (pbase + N - 1) * (UD * UD) + g + UD * L.cb : Z;
  const int sR = fR ? -(int)D8 * (UD * UD) : 0;
  // control-side gradient (ud resp. wu) and Du on the control columns
  const double *gu_src = FACTOR ? a.ud : a.wu;
  const bool fgu = L.cu && (FACTOR || HUB);
  const double *pgu = fgu ? gu_src + (pbase + N - 1) * UD + L.cb : Z;
  const int sgu = fgu ? -(int)D8 * UD : 0;
  // Du: one diagonal entry per control (lanes (XP + b, b)), or — du_full — a full u x u block laid out like R
  const bool fDu = L.cu && FACTOR && HUB && (a.du_full ? gu : g == L.cb);
  const int eDu = a.du_full ? UD * UD : UD;
  const double *pDu = fDu ? a.Du + (pbase + N - 1) * eDu + (a.du_full ? g + UD * L.cb : L.cb) : Z;
  const int sDu = fDu ? -(int)D8 * eDu : 0;
  // state-side gradient (xd resp. wx) and Dx on the diagonal lanes
  const double *gx_src = FACTOR ? a.xd : a.wx;
  const bool fgx = L.cxv && (FACTOR || HXB);
  const double *pgx = fgx ? gx_src + (pbase + N - 1) * XD + L.oc : Z;
  const int sgx = fgx ? -(int)D8 * XD : 0;
  const bool diag_x = L.cxv && ((c & 3) == g);  // kernel row g + 4r == c for r = c >> 2
  const bool fDx = diag_x && FACTOR && HXB;
  const double *pDx = fDx ? a.Dx + (pbase + N - 1) * XD + L.oc : Z;
  const int sDx = fDx ? -(int)D8 * XD : 0;
  bool dmask[KS];  // lane masks (SGPR pairs), not VGPRs
#pragma unroll
  for (int r = 0; r < KS; r++) dmask[r] = diag_x && (c >> 2) == r;
  const bool umask = L.cu && g == L.cb;
  // row-distributed vectors (every lane has an entry unless xdim is padded): 32-bit byte offsets
  unsigned ox_row = (unsigned)(((pbase + N - 1) * XD + L.row0) * D8);
  unsigned ou_g = (unsigned)(((pbase + N - 1) * UD + (gu ? g : 0)) * D8);
  unsigned ou_0 = (unsigned)(((pbase + N - 1) * UD) * D8);
  constexpr unsigned SX = XD * D8, SU = UD * D8;
  double *pRec = a.K + (pbase + N - 1) * 64 + lane;  // factor record of stage N-1, this lane's slot
  const bool frec = (L.cxv || L.cu) && gu;           // lanes whose slot carries a value

  double S[KS], s_row[KS];
  double s_col;
  // "next stage" registers
  double Fn[KS], Qn[KS], xmn[KS], gxn = 0.0, Dun = 0.0, Dxn = 0.0, Rn = 0.0, umn = 0.0, gun, recn = 0.0;

  auto load_row = [&](const double *p, double *dst) {
#pragma unroll
    for (int r = 0; r < KS; r++) dst[r] = (!PADX || L.row0 + r < XD || p == Z) ? p[r] : 0.0;
  };
  // ---- terminal: S = Q~_{N-1} (+Dx), s = g_x,N-1 ; first "next" loads (stage N-1) -----------------------
  {
    double Q0[KS], part = 0.0;
    if (FACTOR) {
      load_row(pQ, Q0);
      double dd = regx;
      if (HXB) dd += *pDx;
#pragma unroll
      for (int r = 0; r < KS; r++) {
        const bool rv = !PADX || (L.row0 + r < XD);
        const double xm = rv ? ldo(a.xm, ox_row + (rv ? r * 8u : 0u)) : 0.0;
        part += Q0[r] * xm;
        S[r] = fma(pwt, Q0[r], dmask[r] ? dd : 0.0);
      }
      part = grp_allsum(part);
    }
    s_col = part + *pgx;  // zero on lanes without a state column
    col_to_row<KS>(s_col, g, s_row);
    // stage N-1 data needed the moment the stage starts (everything else is issued at the top of its stage)
    load_row(pF, Fn);
    gun = *pgu;
    if (FACTOR) {
      Rn = *pR;
      umn = gu ? ldo(a.um, ou_g) : 0.0;
      if (DEEP && HUB) { Dun = *pDu; pDu = badd(pDu, sDu); }
    } else {
      recn = *pRec;
    }
    if (DEEP && N > 1) {  // state side of stage N-2 (consumed during / at the end of stage N-1)
      ox_row -= SX;
      pgx = badd(pgx, sgx);
      gxn = *pgx;
      if (FACTOR) {
        pQ = badd(pQ, sQ);
        load_row(pQ, Qn);
        if (HXB) { pDx = badd(pDx, sDx); Dxn = *pDx; }
#pragma unroll
        for (int r = 0; r < KS; r++) {
          const bool rv = !PADX || (L.row0 + r < XD);
          xmn[r] = rv ? ldo(a.xm, ox_row + (rv ? r * 8u : 0u)) : 0.0;
        }
      }
    }
  }

  for (int j = N - 1; j >= 0; j--) {
    const bool cons = j < Nc;
    // ---- rotate the pipeline registers ----------------------------------------------------------------
    double Fr[KS], Qc[KS];
#pragma unroll
    for (int r = 0; r < KS; r++) Fr[r] = Fn[r];
    const double Rc = Rn, um_g = umn;
    double gu_c = gun, rec = recn;  // rec: this lane's slot of the stage's factor record
    if (!FACTOR && HUB && cons && !own0) gu_c = 0.0;  // consensus shift counted once, on the owner's particle 0
    if (j == 0) {  // stage 0 has no incoming state: A~_0 = 0
#pragma unroll
      for (int r = 0; r < KS; r++) Fr[r] = L.cxv ? 0.0 : Fr[r];
    }
    double xm_row[KS], gx_c = 0.0, Du_c = 0.0, Dx_c = 0.0;
    if (DEEP) {
      // ---- rotate the remaining pipeline registers; prefetch the state side of stage j-2 ---------------
      gx_c = gxn; Du_c = Dun; Dx_c = Dxn;
#pragma unroll
      for (int r = 0; r < KS; r++) { xm_row[r] = xmn[r]; Qc[r] = Qn[r]; }
      if (j > 0 && FACTOR && HUB) { Dun = *pDu; pDu = badd(pDu, sDu); }
      if (j > 1) {
        ox_row -= SX;
        pgx = badd(pgx, sgx);
        gxn = *pgx;
        if (FACTOR) {
          pQ = badd(pQ, sQ);
          load_row(pQ, Qn);
          if (HXB) { pDx = badd(pDx, sDx); Dxn = *pDx; }
#pragma unroll
          for (int r = 0; r < KS; r++) {
            const bool rv = !PADX || (L.row0 + r < XD);
            xmn[r] = rv ? ldo(a.xm, ox_row + (rv ? r * 8u : 0u)) : 0.0;
          }
        }
      }
    } else {
      // ---- same-stage loads (consumed after the G products resp. at the end of the stage) --------------
      if (FACTOR && HUB) { Du_c = *pDu; pDu = badd(pDu, sDu); }
      if (j > 0) {
        ox_row -= SX;
        pgx = badd(pgx, sgx);
        gx_c = *pgx;
        if (FACTOR) {
          pQ = badd(pQ, sQ);
          load_row(pQ, Qc);
          if (HXB) { pDx = badd(pDx, sDx); Dx_c = *pDx; }
#pragma unroll
          for (int r = 0; r < KS; r++) {
            const bool rv = !PADX || (L.row0 + r < XD);
            xm_row[r] = rv ? ldo(a.xm, ox_row + (rv ? r * 8u : 0u)) : 0.0;
          }
        }
      }
    }
    if (j > 0) {
      // ---- prefetch what stage j-1 needs the moment it starts ------------------------------------------
      pF = badd(pF, sF);
      load_row(pF, Fn);
      pgu = badd(pgu, sgu);
      gun = *pgu;
      ou_g -= SU;
      if (FACTOR) {
        pR = badd(pR, sR);
        Rn = *pR;
        umn = gu ? ldo(a.um, ou_g) : 0.0;
      } else {
        pRec -= 64;
        recn = *pRec;
      }
    }

    // ---- h = F' s (+ control gradient) -----------------------------------------------------------------
    double hp = Rc * um_g;
#pragma unroll
    for (int r = 0; r < KS; r++) hp = fma(Fr[r], s_row[r], hp);
    const double h_col = grp_allsum(hp) + gu_c;
    double hu[UD];
#pragma unroll
    for (int b = 0; b < UD; b++) hu[b] = readlane_d(h_col, XP + b);

    v4d H = {0.0, 0.0, 0.0, 0.0};
    if (FACTOR) {
      // ---- H = F' S F + blkdiag(Q~_{j-1}, R~_j) ------------------------------------------------------
      if (j > 0) {
        const double dd = regx + Dx_c;
#pragma unroll
        for (int r = 0; r < KS; r++) H[r] = fma(pwt, Qc[r], dmask[r] ? dd : 0.0);
      }
      H[KS] = fma(pwt, Rc, (umask ? regu : 0.0) + (cons ? 0.0 : Du_c));  // Du_c is zero on lanes without an entry
      v4d G = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int r = 0; r < KS; r++) G = mfma(S[r], Fr[r], G);
#pragma unroll
      for (int r = 0; r < KS; r++) H = mfma(Fr[r], G[r], H);
    }

    if (cons) {
      // ---- consensus stage: no minimisation.  Export the per-particle condensed gradient block and the
      //      diagonal Hessian block Huu; for Nc > 1 keep Y_j = H_ux in the factor record (k_cond_fast builds the
      //      off-diagonal blocks Y_j Gamma_{j-1} from it) and carry P_{j-1} = H_xx on as the cost-to-go ------------
      if (FACTOR) {
        // Hc_part[i][(j UD + p) + nc (j UD + q)] = Huu[p][q] lives in lane (XP + q, p), register KS
        double v = H[KS];
        if (own0 && HUB) v += Du_c;
        const int nc = Nc * UD;
        if (L.cu && gu) a.Hc_part[(size_t)i * nc * nc + (size_t)(j * UD + g) + (size_t)nc * (j * UD + L.cb)] = v;
        *pRec = (L.cxv && gu) ? H[KS] : 0.0;
        pRec -= 64;
#pragma unroll
        for (int r = 0; r < KS; r++) S[r] = H[r];
      }
      if (lane < UD) a.gc_part[(size_t)i * (Nc * UD) + j * UD + lane] = pick<UD>(hu, lane);
      if (j == 0) break;
      double p2 = 0.0;
      if (FACTOR) {
#pragma unroll
        for (int r = 0; r < KS; r++) p2 = fma(Qc[r], xm_row[r], p2);
        p2 = grp_allsum(p2);
      }
      s_col = L.cxv ? h_col + p2 + gx_c : 0.0;
      col_to_row<KS>(s_col, g, s_row);
      ou_0 -= SU;
      continue;
    }

    if (FACTOR) {
      // ---- Cholesky of Huu on lane-uniform values (readlane broadcast of the lower triangle) --------
      double Lc[UD][UD], Ld[UD], col[UD];
      bool bad = false;
#pragma unroll
      for (int q = 0; q < UD; q++) {
#pragma unroll
        for (int pp = q; pp < UD; pp++) {
          double v = readlane_d(H[KS], (XP + q) + 16 * pp);  // Huu[pp][q]
#pragma unroll
          for (int k = 0; k < q; k++) v -= Lc[pp][k] * Lc[q][k];
          if (pp == q) {
            bad |= !(v > 0.0);
            Ld[q] = rsqrt_d(v);
          } else {
            Lc[pp][q] = v * Ld[q];
          }
        }
      }
      if (bad && lane == 0) *a.fail = 2;
      // ---- gather the control rows column-wise and substitute in-lane: state columns give K[:, c], the
      //      control columns get unit right-hand sides and give Huu^-1[:, c - XP] -------------------------
      double rows4[4];
      grp_gather(H[KS], rows4);  // rows4[k] = H[XP + k][c]
#pragma unroll
      for (int k = 0; k < UD; k++) col[k] = L.cu ? (L.cb == k ? 1.0 : 0.0) : rows4[k];
      chol_solve<UD>(Lc, Ld, col);
      const double Kg = pick<UD>(col, g);
      rec = frec ? Kg : 0.0;
      v4d Sn = mfma(H[KS], (L.cxv && gu) ? -Kg : 0.0, H);  // S' = Hxx - Hxu K (columns >= XP are never used)
#pragma unroll
      for (int r = 0; r < KS; r++) S[r] = Sn[r];
      *pRec = rec;  // one coalesced 512-byte store per stage
      pRec -= 64;
    }
    const double Kreg = L.cxv ? rec : 0.0;
    // ---- feed-forward k = Huu^-1 hu: the control quad of k-group g holds row g of Huu^-1 ---------------
    const double hug = pick<UD>(hu, g);
    double kq = L.cu ? rec * pick<UD>(hu, L.cb) : 0.0;
    kq += dpp_d<0xB1>(kq);
    kq += dpp_d<0x4E>(kq);  // sum over the aligned quad c = XP .. XP+3
    if (c == XP && gu) *(double *)((char *)a.kff + ou_0 + g * 8u) = kq;
    if (j == 0) break;
    // ---- s_{j-1} = h_x - K' hu + g_x,j-1 -------------------------------------------------------------
    double p2 = -Kreg * hug;
    if (FACTOR) {
#pragma unroll
      for (int r = 0; r < KS; r++) p2 = fma(Qc[r], xm_row[r], p2);
    }
    const double red2 = grp_allsum(p2);
    s_col = L.cxv ? h_col + red2 + gx_c : 0.0;
    col_to_row<KS>(s_col, g, s_row);
    ou_0 -= SU;
  }
}

// gradient pre-pass of a factor solve: xm = pw (X - X_ref), xd = pw reg_x (X - X_prev) + wx,
// um = pw (U - U_ref), ud = pw reg_u (U - U_prev) + wu (consensus stages: wu only on the owner's particle 0)
__global__ void __launch_bounds__(256) k_grad_prep(LQArgs a) {
  const long long nx = (long long)a.M * a.N * a.x, nu = (long long)a.M * a.N * a.u;
  const long long stride = (long long)gridDim.x * 256;
  const long long perx = (long long)a.N * a.x, peru = (long long)a.N * a.u;
  for (long long k = blockIdx.x * 256LL + threadIdx.x; k < nx; k += stride) {
    const double X = a.X[k];
    const double pw = a.pw ? a.pw[k / perx] : 1.0;
    a.xm[k] = pw * (X - a.X_ref[k]);
    a.xd[k] = pw * a.reg_x * (X - a.X_prev[k]) + (a.wx ? a.wx[k] : 0.0);
  }
  for (long long k = blockIdx.x * 256LL + threadIdx.x; k < nu; k += stride) {
    const double U = a.U[k];
    const double pw = a.pw ? a.pw[k / peru] : 1.0;
    a.um[k] = pw * (U - a.U_ref[k]);
    double w = 0.0;
    if (a.wu) {
      const int j = (int)((k / a.u) % a.N);
      const long long i = k / peru;
      if (j >= a.Nc || (i == 0 && a.owner)) w = a.wu[k];
    }
    a.ud[k] = pw * a.reg_u * (U - a.U_prev[k]) + w;
  }
}

// per-particle cost J_i = 1/2 z'P z + q'z + r of the reference's single-particle QP (PMPC.jl/src/qp_utils.jl:60-162):
//   1/2 sum_j [(x-xr)'Q(x-xr) + reg_x |x-xp|^2 + (u-ur)'R(u-ur) + reg_u |u-up|^2]
//   + 1/2 s sum_{j>=1} |u_j - u_{j-1}|^2 + 1/2 s0 |u_0|^2 - s0 u_0'u_{-1}      (the slew constant 1/2 s0 |u_{-1}|^2 is
//   absent from the reference's r, :140-160).  One 256-thread block per particle, coalesced over the Q / R stacks.
__global__ void __launch_bounds__(256) k_particle_cost(LQArgs a, const double *X, const double *U, double *J) {
  __shared__ double sh[256];
  const int i = blockIdx.x, tid = threadIdx.x, x = a.x, u = a.u, N = a.N;
  const size_t pb = (size_t)i * N;
  const double *Xi = X + pb * x, *Ui = U + pb * u;
  const double *Xr = a.X_ref + pb * x, *Ur = a.U_ref + pb * u, *Xp = a.X_prev + pb * x, *Up = a.U_prev + pb * u;
  double acc = 0.0;
  const double *Q = a.Q + pb * x * x, *R = a.R + pb * u * u;
  // (stage j, column c, row r) of entry e = tid + 256 k by carrying, not by dividing: the two integer divisions per entry were most of
  // this kernel's instructions (124 us at config D against 87 us for its bytes; same entries per thread in the same order: same sums)
  // the deviations x - x_ref, u - u_ref of this particle once into LDS (when they fit: a.cost_lds doubles were given to the launch): an
  // entry of Q then costs one global load and two LDS reads instead of five global loads
  extern __shared__ double dev[];
  const bool staged = a.cost_lds >= N * (x + u);
  if (staged) {
    for (int e = tid; e < N * x; e += 256) dev[e] = Xi[e] - Xr[e];
    for (int e = tid; e < N * u; e += 256) dev[N * x + e] = Ui[e] - Ur[e];
    __syncthreads();
  }
  auto quad = [&](const double *Mx, const double *Z, const double *Zr, const double *D, int d) {
    const int dd = d * d, qa = 256 / d, qb = 256 - qa * d;
    int j = tid / dd, rc = tid - j * dd, c = rc / d, r = rc - c * d;
    const int tot = N * dd;
    int e = tid;
    // four entries per trip, their loads issued together (one load in flight per thread left the kernel at 3.1 TB/s: latency, not bandwidth;
    // same entries per thread in the same order: same sums)
    for (; staged && e + 768 < tot; e += 1024) {
      const double q0 = Mx[e], q1 = Mx[e + 256], q2 = Mx[e + 512], q3 = Mx[e + 768];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const double q = k == 0 ? q0 : (k == 1 ? q1 : (k == 2 ? q2 : q3));
        acc += 0.5 * q * D[j * d + r] * D[j * d + c];
        r += qb; c += qa;
        if (r >= d) { r -= d; c++; }
        while (c >= d) { c -= d; j++; }
      }
    }
    for (; e < tot; e += 256) {
      if (staged) acc += 0.5 * Mx[e] * D[j * d + r] * D[j * d + c];
      else acc += 0.5 * Mx[e] * (Z[j * d + r] - Zr[j * d + r]) * (Z[j * d + c] - Zr[j * d + c]);
      r += qb; c += qa;
      if (r >= d) { r -= d; c++; }
      while (c >= d) { c -= d; j++; }
    }
  };
  quad(Q, Xi, Xr, dev, x);
  quad(R, Ui, Ur, dev + N * x, u);
  for (int e = tid; e < N * x; e += 256) {
    const double d = Xi[e] - Xp[e];
    acc += 0.5 * a.reg_x * d * d;
  }
  const double sl = a.slew[i], sl0 = a.slew0[i];
  for (int e = tid; e < N * u; e += 256) {
    const double d = Ui[e] - Up[e];
    acc += 0.5 * a.reg_u * d * d;
    if (e >= u) {
      const double dv = Ui[e] - Ui[e - u];
      acc += 0.5 * sl * dv * dv;
    } else {
      acc += 0.5 * sl0 * Ui[e] * Ui[e] - sl0 * Ui[e] * a.um1[(size_t)i * u + e];
    }
  }
  sh[tid] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) sh[tid] += sh[tid + o];
    __syncthreads();
  }
  if (tid == 0) J[i] = sh[0];
}

// ------------------------------------------------------------------------------------------------
// forward sweep (ROLLOUT: absolute linear rollout X from U, PMPC.jl/src/types.jl:161-173).
// Stage data (F_j, K_j, k_j resp. U_j - U_prev_j, f_j, X_prev_j) is loaded one stage ahead; lanes
// without an entry read the zero buffer through a zero-stride pointer.
// ------------------------------------------------------------------------------------------------
template <int XD, int UD, bool ROLLOUT>
__global__ void __launch_bounds__(64) k_fwd_fast(LQArgs a, const double *Uin, double *Xout) {
  typedef Lane<XD, UD> LT;
  constexpr int KS = LT::KS, XP = LT::XP;
  constexpr bool PADX = (XD != XP);
  constexpr long long D8 = sizeof(double);
  const int lane = threadIdx.x;
  const LT L(lane);
  const int N = a.N, Nc = a.Nc, i = blockIdx.x, g = L.g, c = L.c;
  const size_t pbase = (size_t)i * N;
  const bool gu = g < UD;
  const double *Z = a.zeros;
  const bool fF = L.cxv || L.cu;
  const double *pF = L.cxv ? a.fx + pbase * (XD * XD) + XD * L.oc + L.row0
                           : (L.cu ? a.fu + pbase * (XD * UD) + XD * L.cb + L.row0 : Z);
  const int sF = fF ? (int)D8 * (L.cxv ? XD * XD : XD * UD) : 0;
  const bool fK = L.cxv && gu && !ROLLOUT;
  const double *pK = fK ? a.K + pbase * 64 + lane : Z;  // this lane's slot of the factor records
  const int sK = fK ? (int)D8 * 64 : 0;
  // control-column source: feed-forward k_j (per k-group) resp. rollout inputs (per control column)
  const bool fk = gu && !ROLLOUT;
  const double *pk = fk ? a.kff + pbase * UD + g : Z;
  const int sk = fk ? (int)D8 * UD : 0;
  const bool fc = L.cu;
  const double *pu = (ROLLOUT && fc) ? Uin + pbase * UD + L.cb : Z;
  const double *pup = (ROLLOUT && fc) ? a.U_prev + pbase * UD + L.cb : Z;
  const double *pdc = (!ROLLOUT && fc) ? a.duc + L.cb : Z;  // shared consensus step, stage by stage
  const int sdc = (!ROLLOUT && fc) ? (int)D8 * UD : 0;
  const int su = (ROLLOUT && fc) ? (int)D8 * UD : 0;
  unsigned ox_row = (unsigned)((pbase * XD + L.row0) * D8);
  unsigned ou_g = (unsigned)((pbase * UD + (gu ? g : 0)) * D8);
  constexpr unsigned SX = XD * D8, SU = UD * D8;
  const bool store_x = (c == 0), store_u = (c == 0) && gu;
  const int src_grp = 16 * (c & 3);

  auto load_row = [&](const double *p, double *dst) {
#pragma unroll
    for (int r = 0; r < KS; r++) dst[r] = (!PADX || L.row0 + r < XD || p == Z) ? p[r] : 0.0;
  };

  double xcol = 0.0;  // dx[oc] on valid state columns (ROLLOUT: X_{j-1} - X_prev_{j-1})
  double Fn[KS], Kn, kn, un, fn[KS], xpn[KS];
  load_row(pF, Fn);
  Kn = *pK;
  kn = *pk;
  un = ROLLOUT ? *pu - *pup : 0.0;
  if (ROLLOUT) {
#pragma unroll
    for (int r = 0; r < KS; r++) {
      const bool rv = !PADX || (L.row0 + r < XD);
      fn[r] = rv ? ldo(a.f, ox_row + (rv ? r * 8u : 0u)) : 0.0;
      xpn[r] = rv ? ldo(a.X_prev, ox_row + (rv ? r * 8u : 0u)) : 0.0;
    }
  }
  for (int j = 0; j < N; j++) {
    double Fr[KS], fr[KS], xpr[KS];
#pragma unroll
    for (int r = 0; r < KS; r++) { Fr[r] = Fn[r]; fr[r] = fn[r]; xpr[r] = xpn[r]; }
    const double Kreg = Kn, kreg = kn, ureg = un;
    if (j == 0) {  // A~_0 = 0
#pragma unroll
      for (int r = 0; r < KS; r++) Fr[r] = L.cxv ? 0.0 : Fr[r];
    }
    if (j + 1 < N) {  // prefetch stage j + 1
      pF = badd(pF, sF);
      load_row(pF, Fn);
      if (ROLLOUT) {
        pu = badd(pu, su);
        pup = badd(pup, su);
        un = *pu - *pup;
#pragma unroll
        for (int r = 0; r < KS; r++) {
          const bool rv = !PADX || (L.row0 + r < XD);
          fn[r] = rv ? ldo(a.f, ox_row + SX + (rv ? r * 8u : 0u)) : 0.0;
          xpn[r] = rv ? ldo(a.X_prev, ox_row + SX + (rv ? r * 8u : 0u)) : 0.0;
        }
      } else {
        pK = badd(pK, sK);
        Kn = *pK;
        pk = badd(pk, sk);
        kn = *pk;
      }
    }
    double ycol;
    if (ROLLOUT) {
      ycol = L.cxv ? xcol : ureg;  // ureg is zero off the control columns
    } else {
      double du_c;
      if (j < Nc) {
        du_c = *pdc;  // shared consensus step (zero off the control columns)
        pdc = badd(pdc, sdc);
        if (store_u) {
          double *o = (double *)((char *)a.dU + ou_g);
          const double v = a.duc[j * UD + g];
          *o = v;
        }
      } else {
        const double sum = row_allsum(Kreg * xcol);
        const double dug = -sum - kreg;  // du[g] in every lane of k-group g (zero for g >= udim)
        const double t = __shfl(dug, 16 * (L.cu ? L.cb : 0), 64);  // all lanes take part: the source lanes are not control columns
        du_c = L.cu ? t : 0.0;
        if (store_u) {
          double *o = (double *)((char *)a.dU + ou_g);
          *o = dug;
        }
      }
      ycol = L.cxv ? xcol : du_c;
    }
    double xr[KS];
#pragma unroll
    for (int r = 0; r < KS; r++) xr[r] = row_allsum(Fr[r] * ycol);
    if (ROLLOUT) {
      // X_j = f_j + fx (X_{j-1} - Xp_{j-1}) + fu (U_j - Up_j); the next column state is X_j - X_prev_j
#pragma unroll
      for (int r = 0; r < KS; r++) {
        const bool rv = !PADX || (L.row0 + r < XD);
        const double xj = xr[r] + fr[r];
        if (store_x && rv) *(double *)((char *)Xout + ox_row + r * 8u) = xj;
        xr[r] = xj - xpr[r];
      }
    } else if (store_x) {
#pragma unroll
      for (int r = 0; r < KS; r++)
        if (!PADX || L.row0 + r < XD) {
          double *o = (double *)((char *)a.dX + ox_row + r * 8u);
          *o = xr[r];
        }
    }
    // next column-distributed state: kernel column c lives in k-group c & 3, register c >> 2
    double nx = 0.0;
#pragma unroll
    for (int r = 0; r < KS; r++) {
      const double t = __shfl(xr[r], src_grp, 64);
      nx = ((c >> 2) == r) ? t : nx;
    }
    xcol = L.cxv ? nx : 0.0;
    ox_row += SX;
    ou_g += SU;
  }
}

// counters[3] <- failure flag (fail != null), and / or publication of counters[0..3] (mirror_cnt != null)
__global__ void k_as_publish(int *counters, const int *fail, int *mirror_cnt, unsigned long long *mirror_seq, unsigned long long seq) {
  if (threadIdx.x == 0) {
    if (fail) counters[3] = *fail;
    if (mirror_cnt) {
      for (int k = 0; k < 4; k++) mirror_cnt[k] = counters[k];
      __threadfence_system();
      *(volatile unsigned long long *)mirror_seq = seq;
    }
  }
}


// ------------------------------------------------------------------------------------------------
// condensed consensus Hessian, off-diagonal blocks (Nc > 1).  With P_j the no-elimination cost-to-go of the
// consensus stages (P_{Nc-1} = S of the free stages, P_{j-1} = Q~_{j-1} + A_j' P_j A_j) and
// Gamma_{j,l} = A_j ... A_{l+1} B_l the sensitivity of x_j to the shared control u_l,
//   Hc[j][l] = B_j' P_j Gamma_{j,l} = Y_j Gamma_{j-1,l}   (l < j),   Y_j = B_j' P_j A_j = H_ux of stage j,
// which k_bwd_fast<FACTOR> left in the factor record; the diagonal blocks are its H_uu.  O(Nc^2) small
// products instead of the O(Nc^3) of forming Phi' M Phi (kernels_generic.hip).  The columns of Gamma are
// independent, so one wave owns COND_TPW column tiles (16 consensus variables each) of one particle, keeps
// them in registers in the MFMA B/C layout and walks the stages: 2*KS MFMAs per tile and stage.
// Output goes to the UPPER triangle Hc_part[i][q + nc (j UD + p)] (q < j UD): 16 lanes store 16 consecutive doubles.
// ------------------------------------------------------------------------------------------------
constexpr int COND_TPW = 4;
template <int XD, int UD>
__global__ void __launch_bounds__(64) k_cond_fast(LQArgs a) {
  typedef Lane<XD, UD> LT;
  constexpr int KS = LT::KS;
  const int lane = threadIdx.x;
  const LT L(lane);
  const int N = a.N, Nc = a.Nc, nc = Nc * UD, i = blockIdx.x, g = L.g, c = L.c;
  if (a.done && *a.done) return;  // (active-set rounds enqueued ahead: the set has settled)
  if (a.as_settled_in && a.as_settled_in[i]) return;
  const size_t pbase = (size_t)i * N;
  const int t0 = blockIdx.y * COND_TPW;
  const int jstart = (16 * t0) / UD;  // first stage at which one of this wave's columns starts
  double *Hc = a.Hc_part + (size_t)i * nc * nc;
  double Gm[COND_TPW][KS];
#pragma unroll
  for (int k = 0; k < COND_TPW; k++)
#pragma unroll
    for (int r = 0; r < KS; r++) Gm[k][r] = 0.0;
  bool kv[KS];  // k index g + 4r is a real state
#pragma unroll
  for (int r = 0; r < KS; r++) kv[r] = (KS * g + r) < XD;
  // no-rollout warm start (a.defect: the base point is the linearisation point, off the linearised dynamics by r_j = f_j -
  // x_prev_j): the defects of the EARLIER consensus stages reach stage j's state as d_{j-1}, d_j = A~_j d_{j-1} + r_j, and the
  // coupling block Y_j turns that into a term Y_j d_{j-1} of the condensed gradient which no backward sweep can form.  One
  // more column (c = 0) through the same products, by the wave that owns the first tile.
  const bool dd = a.defect != nullptr && blockIdx.y == 0;
  double Dm[KS];
#pragma unroll
  for (int r = 0; r < KS; r++) Dm[r] = 0.0;

  for (int j = jstart; j < Nc; j++) {
    // A operand of A~_j: lane (c, g), step r <-> fx_j[pi(c)][pi(g + 4r)];  of Y_j: record[(g + 4r) + 16 c], rows c < UD
    static_assert(LT::XP + UD <= 16 && KS <= 3, "fused [A; Y] operand: the state rows and the control rows share one 16-row tile");
  // ONE A operand for both products: rows c < XP are A~_j (kernel order), rows XP + b row b of Y_j (XP + UD <= 16 for every compiled
    // pair) — the result's registers r < KS are A~_j Gamma (the next Gamma), register KS is Y_j Gamma in row XP + g: half the MFMAs
    double Top[KS];
    const double *fx = a.fx + (pbase + j) * (XD * XD), *rec = a.K + (pbase + j) * 64;
#pragma unroll
    for (int r = 0; r < KS; r++) Top[r] = j > 0 ? ((L.cxv && kv[r]) ? fx[(KS * g + r) * XD + L.oc] : (L.cu ? rec[(g + 4 * r) + 16 * L.cb] : 0.0)) : 0.0;
    const double *fu = a.fu + (pbase + j) * (XD * UD);
    if (dd) {
      v4d n = {0.0, 0.0, 0.0, 0.0};
      if (j > 0) {
#pragma unroll
        for (int r = 0; r < KS; r++) n = mfma(Top[r], Dm[r], n);
        if (c == 0 && g < UD) a.gc_part[(size_t)i * nc + j * UD + g] += n[KS];
      }
#pragma unroll
      for (int r = 0; r < KS; r++) {
        const bool mine = c == 0 && kv[r];
        const size_t e = (pbase + j) * XD + KS * g + r;
        Dm[r] = mine ? n[r] + (a.defect[mine ? e : 0] - a.X_prev[mine ? e : 0]) : 0.0;
      }
    }
#pragma unroll
    for (int k = 0; k < COND_TPW; k++) {
      const int qt = 16 * (t0 + k);
      if (qt >= nc) continue;  // wave-uniform
      const int q = qt + c;
      if (qt < j * UD) {  // some column of this tile started before stage j (wave-uniform)
        v4d n = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < KS; r++) n = mfma(Top[r], Gm[k][r], n);
        if (g < UD && q < j * UD) Hc[(size_t)q + (size_t)nc * (j * UD + g)] = n[KS];
#pragma unroll
        for (int r = 0; r < KS; r++) Gm[k][r] = n[r];
      }
      if (q < nc && q / UD == j) {  // Gamma_{j,j} = B_j
        const int b = q - j * UD;
#pragma unroll
        for (int r = 0; r < KS; r++) Gm[k][r] = kv[r] ? fu[b * XD + KS * g + r] : 0.0;
      }
    }
  }
}

// The same walk with the particles' blocks SUMMED before they leave the chip (a.Hc_grp): COND_GRP waves = COND_GRP particles per
// workgroup, every stage's blocks reduced across the waves through LDS, one slab of partial sums per workgroup instead of one per
// particle.  At full consensus (Nc = N = 50, u = 4, M = 4096) the per-particle Hessians are 1.3 GB written here and read again by the
// reduction — the two largest kernels of that workload —; the group sums are 1/COND_GRP of that.  The diagonal blocks (written into
// Hc_part by the factor sweep) are folded in on the way, so the slab is the complete upper triangle.  For solves that never look at
// one particle's H_i again (no settled-particle refresh, no consensus weights, no epigraph problem on the host).
#ifndef PMPC_COND_GRP
#define PMPC_COND_GRP 8
#endif
constexpr int COND_GRP = PMPC_COND_GRP;
template <int XD, int UD>
__global__ void __launch_bounds__(64 * COND_GRP) k_cond_fast_grouped(LQArgs a) {
  typedef Lane<XD, UD> LT;
  constexpr int KS = LT::KS;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const LT L(lane);
  const int N = a.N, Nc = a.Nc, nc = Nc * UD, g = L.g, c = L.c;
  if (a.done && *a.done) return;
  const int ip = blockIdx.x * COND_GRP + wv;
  const bool live = ip < a.M;
  const int i = live ? ip : 0;  // (a wave beyond the last particle walks particle 0's data and contributes zeros: it must reach the barriers)
  const size_t pbase = (size_t)i * N;
  const int t0 = blockIdx.y * COND_TPW;
  const int jstart = (16 * t0) / UD;
  const double *Hd = a.Hc_part + (size_t)i * nc * nc;
  double *Hg = a.Hc_grp + (size_t)blockIdx.x * nc * nc;
  __shared__ double red[2][COND_TPW][COND_GRP][64];
  double Gm[COND_TPW][KS];
#pragma unroll
  for (int k = 0; k < COND_TPW; k++)
#pragma unroll
    for (int r = 0; r < KS; r++) Gm[k][r] = 0.0;
  bool kv[KS];
#pragma unroll
  for (int r = 0; r < KS; r++) kv[r] = (KS * g + r) < XD;
  const bool dd = a.defect != nullptr && blockIdx.y == 0;
  double Dm[KS];
#pragma unroll
  for (int r = 0; r < KS; r++) Dm[r] = 0.0;

  for (int j = jstart; j < Nc; j++) {
    static_assert(LT::XP + UD <= 16 && KS <= 3, "fused [A; Y] operand: the state rows and the control rows share one 16-row tile");
  // ONE A operand for both products: rows c < XP are A~_j (kernel order), rows XP + b row b of Y_j (XP + UD <= 16 for every compiled
    // pair) — the result's registers r < KS are A~_j Gamma (the next Gamma), register KS is Y_j Gamma in row XP + g: half the MFMAs
    double Top[KS];
    const double *fx = a.fx + (pbase + j) * (XD * XD), *rec = a.K + (pbase + j) * 64;
#pragma unroll
    for (int r = 0; r < KS; r++) Top[r] = j > 0 ? ((L.cxv && kv[r]) ? fx[(KS * g + r) * XD + L.oc] : (L.cu ? rec[(g + 4 * r) + 16 * L.cb] : 0.0)) : 0.0;
    const double *fu = a.fu + (pbase + j) * (XD * UD);
    if (dd) {
      v4d n = {0.0, 0.0, 0.0, 0.0};
      if (j > 0) {
#pragma unroll
        for (int r = 0; r < KS; r++) n = mfma(Top[r], Dm[r], n);
        if (live && c == 0 && g < UD) a.gc_part[(size_t)i * nc + j * UD + g] += n[KS];
      }
#pragma unroll
      for (int r = 0; r < KS; r++) {
        const bool mine = c == 0 && kv[r];
        const size_t e = (pbase + j) * XD + KS * g + r;
        Dm[r] = mine ? n[r] + (a.defect[mine ? e : 0] - a.X_prev[mine ? e : 0]) : 0.0;
      }
    }
    const int buf = j & 1;
#pragma unroll
    for (int k = 0; k < COND_TPW; k++) {
      const int qt = 16 * (t0 + k);
      if (qt >= nc) continue;  // block-uniform
      const int q = qt + c;
      double out = 0.0;
      if (qt < j * UD) {
        v4d n = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < KS; r++) n = mfma(Top[r], Gm[k][r], n);
        if (g < UD && q < j * UD) out = n[KS];
#pragma unroll
        for (int r = 0; r < KS; r++) Gm[k][r] = n[r];
      }
      // diagonal block of stage j, upper triangle (the factor sweep left H_uu there)
      if (g < UD && q < nc && q / UD == j && q <= j * UD + g) out = Hd[(size_t)q + (size_t)nc * (j * UD + g)];
      if (qt < (j + 1) * UD) red[buf][k][wv][lane] = live ? out : 0.0;
      if (q < nc && q / UD == j) {  // Gamma_{j,j} = B_j
        const int b = q - j * UD;
#pragma unroll
        for (int r = 0; r < KS; r++) Gm[k][r] = kv[r] ? fu[b * XD + KS * g + r] : 0.0;
      }
    }
    __syncthreads();
    // wave k sums tile k over the particles of the group (the next stage writes the other buffer: one barrier per stage)
    if (wv < COND_TPW) {
      const int qt = 16 * (t0 + wv), q = qt + c;
      if (qt < nc && qt < (j + 1) * UD && g < UD && q < nc && q <= j * UD + g) {
        double acc = 0.0;
#pragma unroll
        for (int w = 0; w < COND_GRP; w++) acc += red[buf][wv][w][lane];
        Hg[(size_t)q + (size_t)nc * (j * UD + g)] = acc;
      }
    }
  }
}

template <int XD, int UD>
void launch_bwd_t(const LQArgs &a, bool factor, hipStream_t s) {
  const bool xb = a.wx != nullptr, ub = a.wu != nullptr;
  // > 3 waves per SIMD (1024 SIMDs on MI355X) to run: occupancy 4 beats the deeper pipeline
  const bool deep = a.M <= 3 * 1024;
  const dim3 grd(a.M), blk(64);
#define PMPC_BWD(F, XB, UB, DP) hipLaunchKernelGGL((k_bwd_fast<XD, UD, F, XB, UB, DP>), grd, blk, 0, s, a)
#define PMPC_BWD2(F, XB, UB) do { if (deep) PMPC_BWD(F, XB, UB, true); else PMPC_BWD(F, XB, UB, false); } while (0)
  if (factor) {
    if (xb && ub) PMPC_BWD2(true, true, true);
    else if (xb) PMPC_BWD2(true, true, false);
    else if (ub) PMPC_BWD2(true, false, true);
    else PMPC_BWD2(true, false, false);
  } else {
    if (xb && ub) PMPC_BWD2(false, true, true);
    else if (xb) PMPC_BWD2(false, true, false);
    else if (ub) PMPC_BWD2(false, false, true);
    else PMPC_BWD2(false, false, false);
  }
#undef PMPC_BWD2
#undef PMPC_BWD
}
template <int XD, int UD>
void launch_cond_t(const LQArgs &a, hipStream_t s) {
  const int ntiles = (a.Nc * UD + 15) / 16;
  if (a.Hc_grp) hipLaunchKernelGGL((k_cond_fast_grouped<XD, UD>), dim3((a.M + COND_GRP - 1) / COND_GRP, (ntiles + COND_TPW - 1) / COND_TPW), dim3(64 * COND_GRP), 0, s, a);
  else hipLaunchKernelGGL((k_cond_fast<XD, UD>), dim3(a.M, (ntiles + COND_TPW - 1) / COND_TPW), dim3(64), 0, s, a);
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

bool lq_fast_supported(const LQArgs &a) {
  if (a.w != 0 || a.any_slew || !a.sym_cost) return false;
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

int cond_fast_groups(int M) { return (M + COND_GRP - 1) / COND_GRP; }
void launch_cond_fast(const LQArgs &a, hipStream_t s) {
  if (a.Nc <= 1) return;
#define X(xd, ud) if (a.x == xd && a.u == ud) { launch_cond_t<xd, ud>(a, s); return; }
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

void launch_grad_prep(const LQArgs &a, hipStream_t s) {
  long long n = (long long)a.M * a.N * a.x;
  long long b = (n + 255) / 256;
  if (b > 2048) b = 2048;
  hipLaunchKernelGGL(k_grad_prep, dim3((unsigned)b), dim3(256), 0, s, a);
}

void launch_as_publish(int *counters, const int *fail, int *mirror_cnt, unsigned long long *mirror_seq, unsigned long long seq, hipStream_t s) {
  hipLaunchKernelGGL(k_as_publish, dim3(1), dim3(64), 0, s, counters, fail, mirror_cnt, mirror_seq, seq);
}

void launch_particle_cost(const LQArgs &a, const double *X, const double *U, double *J, hipStream_t s) {
  LQArgs b = a;
  const size_t want = (size_t)a.N * (a.x + a.u);
  b.cost_lds = want * sizeof(double) <= 48 * 1024 ? (int)want : 0;
  hipLaunchKernelGGL(k_particle_cost, dim3(a.M), dim3(256), (size_t)b.cost_lds * sizeof(double), s, b, X, U, J);
}

void launch_rollout_fast(const LQArgs &a, const double *U, double *X, hipStream_t s) {
#define X(xd, ud) if (a.x == xd && a.u == ud) { launch_rollout_t<xd, ud>(a, U, X, s); return; }
  PMPC_FAST_DIMS(X)
#undef X
  abort();
}
