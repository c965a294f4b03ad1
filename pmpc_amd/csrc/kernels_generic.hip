// kernels_generic.hip — structured LQ solve for ARBITRARY (xdim, udim, N, Nc, slew) on gfx950.
//
// This is the always-available device path: one 64-lane wavefront per particle, stage matrices
// staged in LDS, lanes striding over output entries.  It handles what the register-resident
// MFMA path (kernels_fast.hip) does not: slew penalties (stage state augmented with the previous
// control, n = xdim + udim), any consensus horizon Nc (forward-sensitivity condensing) and any
// dimensions.  The problem it solves is the Newton system of the joint QP the reference assembles
// in PMPC.jl/src/lqp_utils.jl:2-393:
//
//     min 1/2 dz'(P + D)dz + (g + w)'dz   s.t.  A dz = 0
//
// P: block-diagonal Q_j + reg_x I (:130-141), R_j + reg_u I + slew tridiagonal (:17-102);
// A: the linearised dynamics chain (:219-303); consensus controls (first Nc stages) are shared
// decision variables (:17-61, :231-244) -> per-particle condensed (H_i, g_i) summed over particles.
#include "pmpc_dev.h"
#include <cstdlib>
#include "as_ctl_dev.h"

namespace {

constexpr int WV = 64;

__device__ __forceinline__ size_t vofs(int i, int j, int N, int d) { return ((size_t)i * N + j) * (size_t)d; }
__device__ __forceinline__ size_t mofs(int i, int j, int N, int r, int c) {
  return ((size_t)i * N + j) * (size_t)r * (size_t)c;
}
// slew diagonal of control stage j: PMPC.jl/src/lqp_utils.jl:31-39 / :81-88
__device__ __forceinline__ double slew_diag(double s0, double s, int j, int N) {
  return j == 0 ? s0 + s : (j == N - 1 ? s : 2.0 * s);
}
// OSQP.jl keeps triu(P): effective symmetric entry of a column-major d x d block
__device__ __forceinline__ double symu(const double *B, int d, int r, int t) {
  return r <= t ? B[r + d * t] : B[t + d * r];
}

struct LdsMap {
  double *S, *F, *G, *H, *Lc, *Kt, *Hi, *sv, *hv, *tv, *Qr, *Rr, *gxs;
};
__host__ __device__ inline size_t lds_doubles(int x, int u, int n) {
  int nt = n + u;
  return (size_t)n * n + 2 * (size_t)n * nt + (size_t)nt * nt + 2 * (size_t)u * u + (size_t)u * n + n + 2 * nt +
         (size_t)x * x + (size_t)u * u + x;
}
__device__ inline LdsMap lds_map(double *p, int x, int u, int n) {
  int nt = n + u;
  LdsMap m;
  m.S = p;  p += n * n;
  m.F = p;  p += n * nt;
  m.G = p;  p += n * nt;
  m.H = p;  p += nt * nt;
  m.Lc = p; p += u * u;
  m.Kt = p; p += u * n;
  m.Hi = p; p += u * u;
  m.sv = p; p += n;
  m.hv = p; p += nt;
  m.tv = p; p += nt;
  m.Qr = p; p += x * x;
  m.Rr = p; p += u * u;
  m.gxs = p;
  return m;
}

// F = [A~ | B~]  (n x (n+u), column-major): A~ = [fx 0; 0 0] (zero at stage 0), B~ = [fu; I_w]
__device__ inline void build_F(const LQArgs &a, int i, int j, double *F, int lane) {
  const int x = a.x, u = a.u, n = a.n, nt = a.n + a.u;
  const double *fx = a.fx + mofs(i, j, a.N, x, x);
  const double *fu = a.fu + mofs(i, j, a.N, x, u);
  for (int e = lane; e < n * nt; e += WV) {
    int r = e % n, c = e / n;
    double v = 0.0;
    if (c < x) {
      if (r < x && j > 0) v = fx[r + x * c];
    } else if (c >= n) {
      int t = c - n;
      if (r < x) v = fu[r + x * t];
      else v = (r - x == t) ? 1.0 : 0.0;
    }
    F[e] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// backward sweep.  FACTOR: matrix Riccati + condensing, gradient = smooth gradient (P z + q) + w.
// !FACTOR: vector sweep only with the stored gains, gradient = w alone (the corrector solves for the
// DIFFERENCE between the corrector and predictor steps, so the smooth gradient is never re-read).
// ------------------------------------------------------------------------------------------------
template <bool FACTOR>
__global__ void __launch_bounds__(WV) k_bwd_generic(LQArgs a) {
  extern __shared__ double lds[];
  const int i = blockIdx.x, lane = threadIdx.x;
  const int x = a.x, u = a.u, n = a.n, w = a.w, N = a.N, Nc = a.Nc, nt = n + u, nc = Nc * u;
  LdsMap L = lds_map(lds, x, u, n);
  const double pw = a.pw ? a.pw[i] : 1.0;  // cost weight: scales Q, R, reg and the slew penalties of this particle
  const double sl = pw * a.slew[i], sl0 = pw * a.slew0[i], regx = pw * a.reg_x, regu = pw * a.reg_u;

  // gxs <- effective state gradient of stage jj (smooth part computed here when FACTOR)
  auto state_grad = [&](int jj) {
    if (FACTOR) {
      const double *Qg = a.Q + mofs(i, jj, N, x, x);
      for (int e = lane; e < x * x; e += WV) L.Qr[e] = pw * Qg[e];
      __syncthreads();
      if (lane < x) {
        const double *X = a.X + vofs(i, jj, N, x), *Xp = a.X_prev + vofs(i, jj, N, x), *Xr = a.X_ref + vofs(i, jj, N, x);
        double g = regx * (X[lane] - Xp[lane]);
        for (int t = 0; t < x; t++) g += symu(L.Qr, x, lane, t) * X[t] - L.Qr[lane + x * t] * Xr[t];
        L.gxs[lane] = g + (a.wx ? a.wx[vofs(i, jj, N, x) + lane] : 0.0);
      }
    } else if (lane < x) {  // vector-only sweep: the gradient is the IPM shift alone
      L.gxs[lane] = a.wx ? a.wx[vofs(i, jj, N, x) + lane] : 0.0;
    }
    __syncthreads();
  };
  // tv[0..u) <- smooth control gradient of stage j (Rr holds raw R_j afterwards when FACTOR)
  auto ctrl_grad = [&](int j) {
    if (FACTOR) {
      const double *Rg = a.R + mofs(i, j, N, u, u);
      for (int e = lane; e < u * u; e += WV) L.Rr[e] = pw * Rg[e];
      __syncthreads();
      if (lane < u) {
        const double *U = a.U + vofs(i, j, N, u), *Up = a.U_prev + vofs(i, j, N, u), *Ur = a.U_ref + vofs(i, j, N, u);
        double g = regu * (U[lane] - Up[lane]) + slew_diag(sl0, sl, j, N) * U[lane];
        for (int t = 0; t < u; t++) g += symu(L.Rr, u, lane, t) * U[t] - L.Rr[lane + u * t] * Ur[t];
        if (j > 0) g -= sl * a.U[vofs(i, j - 1, N, u) + lane];
        if (j + 1 < N) g -= sl * a.U[vofs(i, j + 1, N, u) + lane];
        L.tv[lane] = g;
      }
    } else if (lane < u) {
      L.tv[lane] = 0.0;
    }
    __syncthreads();
  };

  // ---- terminal condition: S = blkdiag(Qd_{N-1}, 0), s = [g_x,N-1 ; 0] ---------------------------
  state_grad(N - 1);
  if (FACTOR) {
    for (int e = lane; e < n * n; e += WV) {
      int r = e % n, c = e / n;
      double v = 0.0;
      if (r < x && c < x) {
        v = symu(L.Qr, x, r, c);
        if (r == c) v += regx + (a.Dx ? a.Dx[vofs(i, N - 1, N, x) + r] : 0.0);
      }
      L.S[e] = v;
    }
  }
  for (int e = lane; e < n; e += WV) L.sv[e] = e < x ? L.gxs[e] : 0.0;
  __syncthreads();

  // ---- free stages -------------------------------------------------------------------------------
  for (int j = N - 1; j >= Nc; j--) {
    build_F(a, i, j, L.F, lane);
    ctrl_grad(j);  // tv[0..u) = smooth g_u,j ; Rr = raw R_j (FACTOR)
    if (FACTOR) {
      // G = S F
      for (int e = lane; e < n * nt; e += WV) {
        int r = e % n, c = e / n;
        double acc = 0.0;
        for (int k = 0; k < n; k++) acc += L.S[r + n * k] * L.F[k + n * c];
        L.G[e] = acc;
      }
      __syncthreads();
      // H = F' G + cost blocks
      for (int e = lane; e < nt * nt; e += WV) {
        int r = e % nt, c = e / nt;
        double acc = 0.0;
        for (int k = 0; k < n; k++) acc += L.F[k + n * r] * L.G[k + n * c];
        if (r >= n && c >= n) {
          int rr = r - n, cc = c - n;
          acc += symu(L.Rr, u, rr, cc);
          if (rr == cc) acc += regu + slew_diag(sl0, sl, j, N) + ((a.Du && !a.du_full) ? a.Du[vofs(i, j, N, u) + rr] : 0.0);
          if (a.Du && a.du_full) acc += a.Du[mofs(i, j, N, u, u) + rr + u * cc];
        }
        if (w && j > 0) {  // slew cross term -s u_j' u_{j-1}
          if (r >= n && c >= x && c < n && r - n == c - x) acc -= sl;
          if (c >= n && r >= x && r < n && c - n == r - x) acc -= sl;
        }
        L.H[e] = acc;
      }
      __syncthreads();
      // Cholesky of Huu (lane 0), then K = Huu^-1 Hux and Huu^-1 by substitution (one column per lane)
      for (int e = lane; e < u * u; e += WV) L.Lc[e] = L.H[(n + e % u) + nt * (n + e / u)];
      __syncthreads();
      if (lane == 0) {
        for (int c = 0; c < u; c++) {
          double d = L.Lc[c + u * c];
          for (int k = 0; k < c; k++) d -= L.Lc[c + u * k] * L.Lc[c + u * k];
          if (!(d > 0.0)) { *a.fail = 2; d = 1.0; }
          d = sqrt(d);
          L.Lc[c + u * c] = d;
          for (int r = c + 1; r < u; r++) {
            double v = L.Lc[r + u * c];
            for (int k = 0; k < c; k++) v -= L.Lc[r + u * k] * L.Lc[c + u * k];
            L.Lc[r + u * c] = v / d;
          }
        }
      }
      __syncthreads();
      for (int c = lane; c < n + u; c += WV) {
        double *col = c < n ? L.Kt + u * c : L.Hi + u * (c - n);
        for (int r = 0; r < u; r++) col[r] = c < n ? L.H[(n + r) + nt * c] : (r == c - n ? 1.0 : 0.0);
        for (int r = 0; r < u; r++) {  // L y = b
          double v = col[r];
          for (int k = 0; k < r; k++) v -= L.Lc[r + u * k] * col[k];
          col[r] = v / L.Lc[r + u * r];
        }
        for (int r = u - 1; r >= 0; r--) {  // L' k = y
          double v = col[r];
          for (int k = r + 1; k < u; k++) v -= L.Lc[k + u * r] * col[k];
          col[r] = v / L.Lc[r + u * r];
        }
      }
      __syncthreads();
      double *Kg = a.K + mofs(i, j, N, u, n), *Hg = a.Hinv + mofs(i, j, N, u, u);
      for (int e = lane; e < u * n; e += WV) Kg[e] = L.Kt[e];
      for (int e = lane; e < u * u; e += WV) Hg[e] = L.Hi[e];
      // S_new = Hxx - Hux' K  (into G), then symmetrise into S
      for (int e = lane; e < n * n; e += WV) {
        int r = e % n, c = e / n;
        double acc = L.H[r + nt * c];
        for (int k = 0; k < u; k++) acc -= L.H[(n + k) + nt * r] * L.Kt[k + u * c];
        L.G[e] = acc;
      }
      __syncthreads();
      for (int e = lane; e < n * n; e += WV) {
        int r = e % n, c = e / n;
        L.S[e] = 0.5 * (L.G[r + n * c] + L.G[c + n * r]);
      }
      __syncthreads();
    } else {
      const double *Kg = a.K + mofs(i, j, N, u, n), *Hg = a.Hinv + mofs(i, j, N, u, u);
      for (int e = lane; e < u * n; e += WV) L.Kt[e] = Kg[e];
      for (int e = lane; e < u * u; e += WV) L.Hi[e] = Hg[e];
      __syncthreads();
    }
    // ---- vector sweep: h = F's ; hu = h_u + g_u ; k = Huu^-1 hu ; s = h_x - K'hu (+ g_x,j-1) ----
    for (int c = lane; c < nt; c += WV) {
      double acc = 0.0;
      for (int r = 0; r < n; r++) acc += L.F[r + n * c] * L.sv[r];
      if (c >= n) acc += L.tv[c - n] + (a.wu ? a.wu[vofs(i, j, N, u) + (c - n)] : 0.0);
      L.hv[c] = acc;
    }
    __syncthreads();
    if (lane < u) {
      double acc = 0.0;
      for (int t = 0; t < u; t++) acc += L.Hi[lane + u * t] * L.hv[n + t];
      a.kff[vofs(i, j, N, u) + lane] = acc;
    }
    if (j > 0) state_grad(j - 1); else __syncthreads();  // gxs / Qr <- stage j-1
    for (int c = lane; c < n; c += WV) {
      double acc = L.hv[c];
      for (int r = 0; r < u; r++) acc -= L.Kt[r + u * c] * L.hv[n + r];
      if (j > 0 && c < x) acc += L.gxs[c];
      L.sv[c] = acc;
    }
    if (FACTOR && j > 0) {
      for (int e = lane; e < x * x; e += WV) {
        int r = e % x, c = e / x;
        double v = symu(L.Qr, x, r, c);
        if (r == c) v += regx + (a.Dx ? a.Dx[vofs(i, j - 1, N, x) + r] : 0.0);
        L.S[r + n * c] += v;
      }
    }
    __syncthreads();
  }

  // ---- consensus stages: adjoint sweep (no minimisation), per-particle condensed gradient --------
  for (int j = Nc - 1; j >= 0; j--) {
    build_F(a, i, j, L.F, lane);
    ctrl_grad(j);
    for (int c = lane; c < nt; c += WV) {
      double acc = 0.0;
      for (int r = 0; r < n; r++) acc += L.F[r + n * c] * L.sv[r];
      L.hv[c] = acc;
    }
    __syncthreads();
    if (lane < u) {
      double g = L.hv[n + lane] + L.tv[lane];
      if (i == 0 && a.owner && a.wu) g += a.wu[vofs(0, j, N, u) + lane];
      if (FACTOR && j == 0) g -= sl0 * a.um1[(size_t)i * u + lane];  // lqp_utils.jl:165 (kept only when Nc >= 1)
      a.gc_part[(size_t)i * nc + j * u + lane] = g;
    }
    if (j > 0) state_grad(j - 1); else __syncthreads();
    for (int c = lane; c < n; c += WV) {
      double acc = L.hv[c];
      if (j > 0 && c < x) acc += L.gxs[c];
      L.sv[c] = acc;
    }
    __syncthreads();
  }

  // ---- condensed consensus Hessian H_i = sum_j Phi_j' M_j Phi_j + blkdiag(Rt_j) - slew coupling --
  if (FACTOR && Nc > 0) {
    double *Pa = a.scratch + (size_t)i * 3 * n * nc, *Pb = Pa + (size_t)n * nc, *T = Pb + (size_t)n * nc;
    double *Hc = a.Hc_part + (size_t)i * nc * nc;
    for (int e = lane; e < nc * nc; e += WV) Hc[e] = 0.0;
    for (int e = lane; e < n * nc; e += WV) Pa[e] = 0.0;
    __syncthreads();
    for (int j = 0; j < Nc; j++) {
      const double *fx = a.fx + mofs(i, j, N, x, x), *fu = a.fu + mofs(i, j, N, x, u);
      // Phi_j = A~_j Phi_{j-1} + B~_j E_j
      for (int e = lane; e < n * nc; e += WV) {
        int r = e % n, c = e / n;
        bool mine = (c / u) == j;
        double v = 0.0;
        if (r < x) {
          if (j > 0) for (int k = 0; k < x; k++) v += fx[r + x * k] * Pa[k + n * c];
          if (mine) v += fu[r + x * (c - j * u)];
        } else if (mine && (r - x) == (c - j * u)) v = 1.0;
        Pb[e] = v;
      }
      // M_j: S (stage Nc-1) or blkdiag(Qd_j, 0)
      const bool last = (j == Nc - 1);
      if (!last) {
        const double *Qg = a.Q + mofs(i, j, N, x, x);
        for (int e = lane; e < x * x; e += WV) L.Qr[e] = pw * Qg[e];
      }
      const double *Rg = a.R + mofs(i, j, N, u, u);
      for (int e = lane; e < u * u; e += WV) L.Rr[e] = pw * Rg[e];
      __syncthreads();
      for (int e = lane; e < n * nc; e += WV) {
        int r = e % n, c = e / n;
        double v = 0.0;
        if (last) {
          for (int k = 0; k < n; k++) v += L.S[r + n * k] * Pb[k + n * c];
        } else if (r < x) {
          for (int k = 0; k < x; k++) v += symu(L.Qr, x, r, k) * Pb[k + n * c];
          v += (regx + (a.Dx ? a.Dx[vofs(i, j, N, x) + r] : 0.0)) * Pb[r + n * c];
        }
        T[e] = v;
      }
      __syncthreads();
      for (int e = lane; e < nc * nc; e += WV) {
        int r = e % nc, c = e / nc;
        double acc = 0.0;
        for (int k = 0; k < n; k++) acc += Pb[k + n * r] * T[k + n * c];
        if (r / u == j && c / u == j) {
          int rr = r - j * u, cc = c - j * u;
          acc += symu(L.Rr, u, rr, cc);
          if (i == 0 && a.owner && a.Du && a.du_full) acc += a.Du[mofs(0, j, N, u, u) + rr + u * cc];
          if (rr == cc) {
            acc += regu + slew_diag(sl0, sl, j, N);
            if (i == 0 && a.owner && a.Du && !a.du_full) acc += a.Du[vofs(0, j, N, u) + rr];
          }
        }
        if (j > 0 && (r % u) == (c % u) && ((r / u == j && c / u == j - 1) || (r / u == j - 1 && c / u == j))) acc -= sl;
        Hc[e] += acc;
      }
      __syncthreads();
      double *tmp = Pa; Pa = Pb; Pb = tmp;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// forward sweep: consensus stages take the shared step, free stages apply the Riccati gains
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(WV) k_fwd_generic(LQArgs a) {
  extern __shared__ double lds[];
  const int i = blockIdx.x, lane = threadIdx.x;
  const int x = a.x, u = a.u, n = a.n, N = a.N, Nc = a.Nc;
  double *xi = lds, *xn = xi + n, *du = xn + n;
  for (int e = lane; e < n; e += WV) xi[e] = 0.0;
  __syncthreads();
  for (int j = 0; j < N; j++) {
    if (lane < u) {
      double v;
      if (j < Nc) v = a.duc[j * u + lane];
      else {
        const double *Kg = a.K + mofs(i, j, N, u, n);
        v = -a.kff[vofs(i, j, N, u) + lane];
        for (int c = 0; c < n; c++) v -= Kg[lane + u * c] * xi[c];
      }
      du[lane] = v;
      a.dU[vofs(i, j, N, u) + lane] = v;
    }
    __syncthreads();
    const double *fx = a.fx + mofs(i, j, N, x, x), *fu = a.fu + mofs(i, j, N, x, u);
    for (int r = lane; r < n; r += WV) {
      double v = 0.0;
      if (r < x) {
        if (j > 0) for (int c = 0; c < x; c++) v += fx[r + x * c] * xi[c];
        for (int t = 0; t < u; t++) v += fu[r + x * t] * du[t];
        a.dX[vofs(i, j, N, x) + r] = v;
      } else v = du[r - x];
      xn[r] = v;
    }
    __syncthreads();
    for (int e = lane; e < n; e += WV) xi[e] = xn[e];
    __syncthreads();
  }
}

// linear rollout of the reference (PMPC.jl/src/types.jl:161-173): X from U
__global__ void __launch_bounds__(WV) k_rollout(LQArgs a, const double *U, double *X) {
  extern __shared__ double lds[];
  const int i = blockIdx.x, lane = threadIdx.x;
  const int x = a.x, u = a.u, N = a.N;
  double *dxp = lds, *dup = dxp + x;
  for (int j = 0; j < N; j++) {
    if (lane < x) dxp[lane] = j > 0 ? X[vofs(i, j - 1, N, x) + lane] - a.X_prev[vofs(i, j - 1, N, x) + lane] : 0.0;
    if (lane < u) dup[lane] = U[vofs(i, j, N, u) + lane] - a.U_prev[vofs(i, j, N, u) + lane];
    __syncthreads();
    const double *fx = a.fx + mofs(i, j, N, x, x), *fu = a.fu + mofs(i, j, N, x, u);
    for (int r = lane; r < x; r += WV) {
      double v = a.f[vofs(i, j, N, x) + r];
      if (j > 0) for (int c = 0; c < x; c++) v += fx[r + x * c] * dxp[c];
      for (int t = 0; t < u; t++) v += fu[r + x * t] * dup[t];
      X[vofs(i, j, N, x) + r] = v;
    }
    __syncthreads();
  }
}

// deterministic sum over particles: src [M][E] -> dst [gridDim.y][E]
__global__ void __launch_bounds__(256) k_reduce_particles(const double *src, double *dst, int M, int E) {
  __shared__ double red[4][64];
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int e = blockIdx.x * 64 + tx;
  double acc = 0.0;
  if (e < E)
    for (int i = blockIdx.y * 4 + ty; i < M; i += 4 * gridDim.y) acc += src[(size_t)i * E + e];
  red[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && e < E) dst[(size_t)blockIdx.y * E + e] = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
}

// dense SPD solve of the reduced consensus system (single workgroup; nc = Nc*udim)
__global__ void __launch_bounds__(256) k_cons_solve(const double *Hc, double *Lc, const double *gc, double *duc, int nc,
                                                    int factor, int *fail) {
  extern __shared__ double y[];
  const int tid = threadIdx.x, nth = blockDim.x;
  if (factor) {
    // the lower triangle of L is built from the UPPER triangle of Hc (the fast path only fills that one)
    for (int e = tid; e < nc * nc; e += nth) {
      const int r = e % nc, cc = e / nc;
      Lc[e] = r >= cc ? Hc[cc + (size_t)nc * r] : 0.0;
    }
    __syncthreads();
    for (int k = 0; k < nc; k++) {
      if (tid == 0) {
        double d = Lc[k + (size_t)nc * k];
        if (!(d > 0.0)) { *fail = 2; d = 1.0; }
        Lc[k + (size_t)nc * k] = sqrt(d);
      }
      __syncthreads();
      const double dk = Lc[k + (size_t)nc * k];
      for (int r = k + 1 + tid; r < nc; r += nth) Lc[r + (size_t)nc * k] /= dk;
      __syncthreads();
      const int rem = nc - k - 1;
      for (int e = tid; e < rem * rem; e += nth) {
        int r = k + 1 + e % rem, c = k + 1 + e / rem;
        if (r >= c) Lc[r + (size_t)nc * c] -= Lc[r + (size_t)nc * k] * Lc[c + (size_t)nc * k];
      }
      __syncthreads();
    }
  }
  for (int e = tid; e < nc; e += nth) y[e] = -gc[e];
  __syncthreads();
  for (int k = 0; k < nc; k++) {  // L y = b
    if (tid == 0) y[k] /= Lc[k + (size_t)nc * k];
    __syncthreads();
    const double yk = y[k];
    for (int r = k + 1 + tid; r < nc; r += nth) y[r] -= Lc[r + (size_t)nc * k] * yk;
    __syncthreads();
  }
  for (int k = nc - 1; k >= 0; k--) {  // L' x = y
    if (tid == 0) y[k] /= Lc[k + (size_t)nc * k];
    __syncthreads();
    const double yk = y[k];
    for (int r = tid; r < k; r += nth) y[r] -= Lc[k + (size_t)nc * r] * yk;
    __syncthreads();
  }
  for (int e = tid; e < nc; e += nth) duc[e] = y[e];
}

#ifndef PMPC_CONS_BLOCKED_THREADS
#define PMPC_CONS_BLOCKED_THREADS 256  // (r04 A/B at config D with Nc = N, nc = 200: 1024 threads are SLOWER — consensus class 1.02 -> 1.57 ms per step: barrier-bound)
#endif
// ---- blocked variant (nc <= PMPC_CONS_BLOCKED_MAX): right-looking Cholesky with 16-column panels ----------------
// Per panel: the 16 x 16 diagonal block is factored by ONE wave in registers (lane l = row l, pivots and
// multipliers broadcast with v_readlane — no barriers), the rows below are solved one per thread against it, and
// the trailing lower triangle gets its rank-16 update in 4 x 4 register blocks with the panel staged in LDS
// (k-major, conflict-free).  3 barriers per panel instead of 3 per COLUMN; the two substitutions use the same
// blocking (wave-level 16 x 16 triangular solve + one matrix-vector update per block: 2 barriers per block).
__device__ __forceinline__ double rl_d(double v, int l) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

__global__ void __launch_bounds__(PMPC_CONS_BLOCKED_THREADS) k_cons_solve_blocked(const double *Hc, double *Lc, const double *gc, double *duc, int nc,
                                                            int factor, int *fail) {
  extern __shared__ double sm[];
  double *P = sm;                      // panel, k-major: P[k * nc + r]
  double *y = P + (size_t)16 * nc;     // right-hand side / solution
  double *Dg = y + nc;                 // diagonal block, Dg[l * 17 + c]
  const int tid = threadIdx.x, nth = blockDim.x, lane = tid & 63;
  if (factor) {
    for (int e = tid; e < nc * nc; e += nth) {
      const int r = e % nc, cc = e / nc;
      Lc[e] = r >= cc ? Hc[cc + (size_t)nc * r] : 0.0;  // lower triangle of L from the UPPER triangle of Hc
    }
    __syncthreads();
    for (int kb = 0; kb < nc; kb += 16) {
      const int bs = min(16, nc - kb), rem = nc - kb - bs;
      if (tid < 64) {
        double a[16];
#pragma unroll
        for (int cc = 0; cc < 16; cc++)
          a[cc] = (lane < bs && cc <= lane) ? Lc[(kb + lane) + (size_t)nc * (kb + cc)] : ((cc == lane) ? 1.0 : 0.0);
        bool bad = false;
#pragma unroll
        for (int p = 0; p < 16; p++) {
          double d = rl_d(a[p], p);
          if (!(d > 0.0)) { bad = true; d = 1.0; }
          const double sd = sqrt(d), rd = 1.0 / sd;
          a[p] = (lane == p) ? sd : a[p] * rd;
#pragma unroll
          for (int cc = p + 1; cc < 16; cc++) {
            const double t = rl_d(a[p], cc);
            a[cc] = (lane >= cc) ? fma(-a[p], t, a[cc]) : a[cc];
          }
        }
        if (bad && lane == 0) *fail = 2;
        if (lane < 16) {
#pragma unroll
          for (int cc = 0; cc < 16; cc++) {
            Dg[lane * 17 + cc] = a[cc];
            if (lane < bs && cc <= lane) Lc[(kb + lane) + (size_t)nc * (kb + cc)] = a[cc];
          }
        }
      }
      __syncthreads();
      for (int r = tid; r < rem; r += nth) {
        const int R = kb + bs + r;
        double xv[16];
#pragma unroll
        for (int cc = 0; cc < 16; cc++) {
          double v = 0.0;
          if (cc < bs) {
            v = Lc[R + (size_t)nc * (kb + cc)];
#pragma unroll
            for (int k = 0; k < cc; k++) v = fma(-xv[k], Dg[cc * 17 + k], v);
            v /= Dg[cc * 17 + cc];
            Lc[R + (size_t)nc * (kb + cc)] = v;
          }
          xv[cc] = v;
          P[cc * nc + r] = v;
        }
      }
      __syncthreads();
      const int nb4 = (rem + 3) / 4, total = nb4 * (nb4 + 1) / 2, base = kb + bs;
      for (int e = tid; e < total; e += nth) {
        int br = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
        while (br * (br + 1) / 2 > e) br--;
        while ((br + 1) * (br + 2) / 2 <= e) br++;
        const int bc = e - br * (br + 1) / 2, r0 = 4 * br, c0 = 4 * bc;
        double acc[4][4];
#pragma unroll
        for (int ii = 0; ii < 4; ii++)
#pragma unroll
          for (int jj = 0; jj < 4; jj++) acc[ii][jj] = 0.0;
        for (int k = 0; k < 16; k++) {
          double pr[4], pc[4];
#pragma unroll
          for (int ii = 0; ii < 4; ii++) {
            pr[ii] = (r0 + ii < rem) ? P[k * nc + r0 + ii] : 0.0;
            pc[ii] = (c0 + ii < rem) ? P[k * nc + c0 + ii] : 0.0;
          }
#pragma unroll
          for (int ii = 0; ii < 4; ii++)
#pragma unroll
            for (int jj = 0; jj < 4; jj++) acc[ii][jj] = fma(pr[ii], pc[jj], acc[ii][jj]);
        }
#pragma unroll
        for (int ii = 0; ii < 4; ii++)
#pragma unroll
          for (int jj = 0; jj < 4; jj++) {
            const int rr = r0 + ii, cc = c0 + jj;
            if (rr < rem && cc <= rr) Lc[(base + rr) + (size_t)nc * (base + cc)] -= acc[ii][jj];
          }
      }
      __syncthreads();
    }
  }
  for (int e = tid; e < nc; e += nth) y[e] = -gc[e];
  __syncthreads();
  for (int kb = 0; kb < nc; kb += 16) {  // L z = b
    const int bs = min(16, nc - kb), rem = nc - kb - bs;
    if (tid < 64) {
      double a[16];
#pragma unroll
      for (int cc = 0; cc < 16; cc++)
        a[cc] = (lane < bs && cc <= lane) ? Lc[(kb + lane) + (size_t)nc * (kb + cc)] : ((cc == lane) ? 1.0 : 0.0);
      double yl = lane < bs ? y[kb + lane] : 0.0;
#pragma unroll
      for (int p = 0; p < 16; p++) {
        const double zp = rl_d(yl, p) / rl_d(a[p], p);
        yl = (lane == p) ? zp : ((lane > p) ? fma(-a[p], zp, yl) : yl);
      }
      if (lane < bs) y[kb + lane] = yl;
    }
    __syncthreads();
    for (int r = tid; r < rem; r += nth) {
      const int R = kb + bs + r;
      double v = y[R];
      for (int k = 0; k < bs; k++) v = fma(-Lc[R + (size_t)nc * (kb + k)], y[kb + k], v);
      y[R] = v;
    }
    __syncthreads();
  }
  for (int kb = ((nc - 1) / 16) * 16; kb >= 0; kb -= 16) {  // L' x = z
    const int bs = min(16, nc - kb);
    if (tid < 64) {
      double b[16];  // lane l holds COLUMN l of the diagonal block
#pragma unroll
      for (int cc = 0; cc < 16; cc++)
        b[cc] = (cc < bs && lane <= cc) ? Lc[(kb + cc) + (size_t)nc * (kb + lane)] : ((cc == lane) ? 1.0 : 0.0);
      double yl = lane < bs ? y[kb + lane] : 0.0;
#pragma unroll
      for (int p = 15; p >= 0; p--) {
        const double xp = rl_d(yl, p) / rl_d(b[p], p);
        yl = (lane == p) ? xp : ((lane < p) ? fma(-b[p], xp, yl) : yl);
      }
      if (lane < bs) y[kb + lane] = yl;
    }
    __syncthreads();
    for (int r = tid; r < kb; r += nth) {
      double v = y[r];
      for (int k = 0; k < bs; k++) v = fma(-Lc[(kb + k) + (size_t)nc * r], y[kb + k], v);
      y[r] = v;
    }
    __syncthreads();
  }
  for (int e = tid; e < nc; e += nth) duc[e] = y[e];
}

// small consensus systems (nc^2 + nc <= 32, e.g. Nc = 1): the per-particle (H_i, g_i) are summed in a fixed
// order by gridDim.x slices (stage 1: outH/outg hold one partial per slice) and a final single block
// (stage 2, gridDim.x == 1, solve_now) sums the slices and solves the dense system.  with_H = 0: gradient
// only, stored factor.
// LDS-resident variant of k_cons_solve_blocked for 16 < nc <= 88 (the whole factor in LDS, 1024 threads, no global
// read-modify-write between panels) with the inverses of the 16 x 16 diagonal blocks kept next to the factor: the panel solve
// of the factorisation and both substitutions become small matrix products instead of 16-step dependent chains.
// Lc: nc x nc factor (column-major lower) followed by ceil(nc / 16) inverse blocks of 16 x 17 doubles (for the vector-only
// solves that follow a factorisation).  Same arithmetic order for every caller: all ranks get the same bits.
__global__ void __launch_bounds__(1024) k_cons_solve_lds(const double *Hc, double *Lc, const double *gc, double *duc, int nc, int factor,
                                                         int *fail) {
  extern __shared__ double sm[];
  double *Ls = sm;                          // factor, column-major, lower triangle
  double *y = Ls + (size_t)nc * nc;         // right-hand side / solution
  double *Dg = y + nc;                      // current diagonal block L_kk, Dg[l * 17 + c]
  const int npan = (nc + 15) / 16;
  double *Di = Dg + 16 * 17;                // inverses of ALL diagonal blocks, Di[p * 272 + l * 17 + c]
  double *DiG = Lc + (size_t)nc * nc;
  const int tid = threadIdx.x, nth = blockDim.x, lane = tid & 63;
  if (factor) {
    for (int e = tid; e < nc * nc; e += nth) {
      const int r = e % nc, cc = e / nc;
      Ls[e] = r >= cc ? Hc[cc + (size_t)nc * r] : 0.0;  // lower triangle of L from the UPPER triangle of Hc
    }
    __syncthreads();
    for (int kb = 0, pn = 0; kb < nc; kb += 16, pn++) {
      const int bs = min(16, nc - kb), rem = nc - kb - bs;
      if (tid < 64) {  // diagonal block: one wave, registers + readlane (as in k_cons_solve_blocked)
        double a[16];
#pragma unroll
        for (int cc = 0; cc < 16; cc++) a[cc] = (lane < bs && cc <= lane) ? Ls[(kb + lane) + (size_t)nc * (kb + cc)] : ((cc == lane) ? 1.0 : 0.0);
        bool bad = false;
        double rdv[16];  // reciprocal diagonal of the block's factor (lane-uniform)
#pragma unroll
        for (int p = 0; p < 16; p++) {
          double d = rl_d(a[p], p);
          if (!(d > 0.0)) { bad = true; d = 1.0; }
          // 1 / sqrt(d) by the hardware seed + two Newton steps (full double accuracy; sqrt + divide on this dependent chain
          // were most of the kernel's time for small systems)
          double rd = __builtin_amdgcn_rsq(d);
          rd = fma(rd, fma(-0.5 * d * rd, rd, 0.5), rd);
          rd = fma(rd, fma(-0.5 * d * rd, rd, 0.5), rd);
          const double sd = d * rd;
          rdv[p] = rd;
          a[p] = (lane == p) ? sd : a[p] * rd;
#pragma unroll
          for (int cc = p + 1; cc < 16; cc++) {
            const double t = rl_d(a[p], cc);
            a[cc] = (lane >= cc) ? fma(-a[p], t, a[cc]) : a[cc];
          }
        }
        if (bad && lane == 0) *fail = 2;
        if (lane < 16) {
#pragma unroll
          for (int cc = 0; cc < 16; cc++) {
            Dg[lane * 17 + cc] = a[cc];
            if (lane < bs && cc <= lane) Ls[(kb + lane) + (size_t)nc * (kb + cc)] = a[cc];
          }
        }
        // inverse of the block: lane c < 16 solves L x = e_c (column c of L^-1) by forward substitution on its own registers,
        // rows of L broadcast by readlane from the lanes that hold them
        double xcol[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
          double v = (r == lane) ? 1.0 : 0.0;
#pragma unroll
          for (int k = 0; k < r; k++) v = fma(-rl_d(a[k], r), xcol[k], v);  // a[k] in lane r = L[r][k]
          xcol[r] = v * rdv[r];
        }
        if (lane < 16) {
#pragma unroll
          for (int r = 0; r < 16; r++) Di[pn * 272 + r * 17 + lane] = (r >= lane) ? xcol[r] : 0.0;
        }
      }
      __syncthreads();
      // panel below the block: X = A L_kk^-T, i.e. X[R][c] = sum_{k <= c} A[R][k] Linv[c][k]: independent entries
      for (int e = tid; e < rem * 16; e += nth) {
        const int r = e % rem, cc = e / rem;
        if (cc < bs) {
          const int R = kb + bs + r;
          double v = 0.0;
          for (int k = 0; k <= cc; k++) v = fma(Ls[R + (size_t)nc * (kb + k)], Di[pn * 272 + cc * 17 + k], v);
          // results go to a scratch column block first (the inputs of other entries of the same row are still being read)
          sm[(size_t)nc * nc + nc + 16 * 17 + npan * 272 + (size_t)cc * nc + r] = v;
        }
      }
      __syncthreads();
      for (int e = tid; e < rem * bs; e += nth) {
        const int r = e % rem, cc = e / rem;
        Ls[(kb + bs + r) + (size_t)nc * (kb + cc)] = sm[(size_t)nc * nc + nc + 16 * 17 + npan * 272 + (size_t)cc * nc + r];
      }
      __syncthreads();
      // trailing update, 4 x 4 register tiles over the lower triangle
      const int nb4 = (rem + 3) / 4, total = nb4 * (nb4 + 1) / 2, base = kb + bs;
      for (int e = tid; e < total; e += nth) {
        int br = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
        while (br * (br + 1) / 2 > e) br--;
        while ((br + 1) * (br + 2) / 2 <= e) br++;
        const int bc = e - br * (br + 1) / 2, r0 = 4 * br, c0 = 4 * bc;
        double acc[4][4];
#pragma unroll
        for (int ii = 0; ii < 4; ii++)
#pragma unroll
          for (int jj = 0; jj < 4; jj++) acc[ii][jj] = 0.0;
        for (int k = 0; k < bs; k++) {
          double pr[4], pc[4];
#pragma unroll
          for (int ii = 0; ii < 4; ii++) {
            pr[ii] = (r0 + ii < rem) ? Ls[(base + r0 + ii) + (size_t)nc * (kb + k)] : 0.0;
            pc[ii] = (c0 + ii < rem) ? Ls[(base + c0 + ii) + (size_t)nc * (kb + k)] : 0.0;
          }
#pragma unroll
          for (int ii = 0; ii < 4; ii++)
#pragma unroll
            for (int jj = 0; jj < 4; jj++) acc[ii][jj] = fma(pr[ii], pc[jj], acc[ii][jj]);
        }
#pragma unroll
        for (int ii = 0; ii < 4; ii++)
#pragma unroll
          for (int jj = 0; jj < 4; jj++) {
            const int rr = r0 + ii, cc = c0 + jj;
            if (rr < rem && cc <= rr) Ls[(base + rr) + (size_t)nc * (base + cc)] -= acc[ii][jj];
          }
      }
      __syncthreads();
    }
    for (int e = tid; e < nc * nc; e += nth) Lc[e] = Ls[e];
    for (int e = tid; e < npan * 272; e += nth) DiG[e] = Di[e];
  } else {
    for (int e = tid; e < nc * nc; e += nth) Ls[e] = Lc[e];
    for (int e = tid; e < npan * 272; e += nth) Di[e] = DiG[e];
  }
  for (int e = tid; e < nc; e += nth) y[e] = -gc[e];
  __syncthreads();
  for (int kb = 0, pn = 0; kb < nc; kb += 16, pn++) {  // L z = b
    const int bs = min(16, nc - kb), rem = nc - kb - bs;
    double v = 0.0;
    if (tid < bs)
      for (int k = 0; k <= tid; k++) v = fma(Di[pn * 272 + tid * 17 + k], y[kb + k], v);
    __syncthreads();
    if (tid < bs) y[kb + tid] = v;
    __syncthreads();
    for (int r = tid; r < rem; r += nth) {
      const int R = kb + bs + r;
      double t = y[R];
      for (int k = 0; k < bs; k++) t = fma(-Ls[R + (size_t)nc * (kb + k)], y[kb + k], t);
      y[R] = t;
    }
    __syncthreads();
  }
  for (int pn = npan - 1; pn >= 0; pn--) {  // L' x = z
    const int kb = pn * 16, bs = min(16, nc - kb);
    double v = 0.0;
    if (tid < bs)
      for (int k = tid; k < bs; k++) v = fma(Di[pn * 272 + k * 17 + tid], y[kb + k], v);  // (L_kk^-T)[tid][k] = Linv[k][tid]
    __syncthreads();
    if (tid < bs) y[kb + tid] = v;
    __syncthreads();
    for (int r = tid; r < kb; r += nth) {
      double t = y[r];
      for (int k = 0; k < bs; k++) t = fma(-Ls[(kb + k) + (size_t)nc * r], y[kb + k], t);
      y[r] = t;
    }
    __syncthreads();
  }
  for (int e = tid; e < nc; e += nth) duc[e] = y[e];
}

// ---- register-resident variant for the systems that do not fit LDS (80 < nc <= 255: Nc = N at config D is 200) ----------------
// The lower triangle lives in the REGISTERS of one 512-thread workgroup as 16 x 16 blocks in the accumulator layout of
// v_mfma_f64_16x16x4_f64 (lane (c, g), register r <-> entry [g + 4r][c]; blocks dealt to the 8 waves round-robin in column-major
// order, so the blocks still active at any panel are spread over all waves) for the whole factorisation; LDS carries the current
// 16-column panel (raw, then solved; row-major, stride 17) and the inverses of the diagonal blocks.  k_cons_solve_blocked keeps the
// matrix in global memory and pays a read-modify-write round trip per trailing tile and panel (0.33 ms at nc = 200, a chain of 13
// panels).  A panel here: (A) the owners of the panel's blocks publish them, (B) one wave factors the 16 x 16 diagonal block in
// registers (readlane broadcasts, 1/sqrt by seed + Newton) and inverts it, (C) every block of the panel becomes A L_kk^-T by 4 MFMAs
// and is published again, (D) every trailing block takes its rank-16 update by 4 MFMAs.  The right-hand side rides along as row nc
// of the matrix (L z = -g falls out of (C)), and L' x = z runs right-looking from the blocks with the four per-lane-group partial sums
// added in a FIXED order: every rank that solves the same system gets the same bits.
// Lc gets the factor in k_cons_solve_blocked's layout (its vector-only solves follow a factorisation done here).
typedef double cv4d __attribute__((ext_vector_type(4)));
constexpr int CONS_REG_NTH = 512, CONS_REG_NW = CONS_REG_NTH / 64;
constexpr int CONS_REG_NCP = 256;   // padded row count of the LDS panels (16 block rows)
constexpr int CONS_REG_LD = 17;     // row stride of the LDS panels
__host__ __device__ inline bool cons_reg_fits(int nc) { return nc > 16 && (nc + 1 + 15) / 16 <= CONS_REG_NCP / 16; }
constexpr size_t CONS_REG_LDS = ((size_t)2 * CONS_REG_NCP * CONS_REG_LD + (size_t)(CONS_REG_NCP / 16) * 272 + 272 + 256 + CONS_REG_NCP + 4 * CONS_REG_NCP) * sizeof(double);

// (B) of k_cons_solve_reg, one wave: Cholesky factor of the 16 x 16 diagonal block (`rows`, stride CONS_REG_LD; identity beyond bs) and the
// inverse of the factor.  All 64 lanes work on the block in the accumulator layout (lane (c, g), register r <-> entry [g + 4r][c]): per pivot
// the scaled column goes through LDS — which leaves the factor there, Dg[p * 16 + (row & 3) * 4 + (row >> 2)] = L[row][p] — and comes back
// as four row values + one column value per lane for the rank-one update (4 FMAs per lane); row p of the inverse (lane = its column) follows
// in the same step from the rows already published.  The lane-per-row form with readlane broadcasts was 1 800 instructions for ONE wave
// (5.5 us per panel, 71 of the kernel's 131 us at nc = 200).  Its own function, not inlined: next to the 96 block registers of the kernel
// the chain went to spill code.
__device__ __attribute__((noinline)) void cons_diag_block(const double *rows, double *Dg, double *Dinv, int bs, int *fail) {
  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  double m[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int row = g + 4 * r;
    m[r] = (row < bs && c <= row) ? rows[row * CONS_REG_LD + c] : ((row == c) ? 1.0 : 0.0);
  }
  const int vidx = (c & 3) * 4 + (c >> 2);  // where L[c][p] sits in a published column
  double xc[16];
  bool bad = false;
#pragma unroll
  for (int p = 0; p < 16; p++) {
    double d = rl_d(m[p >> 2], p + 16 * (p & 3));
    bad |= !(d > 0.0);
    d = (d > 0.0) ? d : 1.0;
    double rd = __builtin_amdgcn_rsq(d);
    rd = fma(rd, fma(-0.5 * d * rd, rd, 0.5), rd);
    rd = fma(rd, fma(-0.5 * d * rd, rd, 0.5), rd);
    // (no branches in the chain: the lanes that do not hold the pivot column write to a scratch line of their own, so that the
    //  whole factorisation is ONE basic block and the scheduler can put the inverse's dot products into the chain's stalls)
    double *dst = (c == p) ? Dg + p * 16 + g * 4 : Dg + 272 + lane * 4;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      m[r] = (c == p) ? m[r] * rd : m[r];  // (the pivot itself: d * rd = sqrt(d); rows above it: never read)
      dst[r] = m[r];
    }
    __builtin_amdgcn_wave_barrier();
    const double vl = Dg[p * 16 + vidx], v = (c > p) ? vl : 0.0;
#pragma unroll
    for (int r = 0; r < 4; r++) m[r] = fma(-Dg[p * 16 + g * 4 + r], v, m[r]);
    // row p of the inverse: x[p] = (e_p - sum_{k < p} L[p][k] x[k]) / L[p][p], this lane's column (the four lane groups alike)
    double sx = (p == c) ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < p; k++) sx = fma(-Dg[k * 16 + (p & 3) * 4 + (p >> 2)], xc[k], sx);
    xc[p] = sx * rd;
    Dinv[p * 17 + c] = (p >= c) ? xc[p] : 0.0;
  }
  if (bad && lane == 0) *fail = 2;
}

// (a uniform value the optimiser may not reason about: keeps the per-slot LDS addresses from being hoisted out of the panel loop —
//  12 slots x 8 addresses held in registers next to the 96 of the blocks spilled)
__device__ __forceinline__ int launder_s(int v) {
  asm volatile("" : "+s"(v));
  return v;
}
template <int MAXQ>
__global__ void __launch_bounds__(CONS_REG_NTH) k_cons_solve_reg(const double *Hc, double *Lc, const double *gc, double *duc, int nc, int *fail) {
  extern __shared__ double sm[];
  constexpr int NCP = CONS_REG_NCP, LD = CONS_REG_LD, NTH = CONS_REG_NTH, NW = CONS_REG_NW;
  double *Praw = sm;                      // the panel as its owners hold it: Praw[row * LD + k]
  double *P = Praw + NCP * LD;            // the panel solved against the diagonal block (the block's own rows included)
  double *Di = P + NCP * LD;              // inverse of every diagonal block's factor, Di[p * 272 + r * 17 + c]
  double *Dg = Di + (NCP / 16) * 272;     // the current diagonal block's factor, column by column (layout: cons_diag_block)
  double *zs = Dg + 272 + 256;            // z, then x  (behind Dg: the 64 x 4 scratch lines of cons_diag_block)
  double *part = zs + NCP;                // backward substitution: part[lane group][column]
  const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n1 = nc + 1, nb = (n1 + 15) / 16, nblk = nb * (nb + 1) / 2, npan = (nc + 15) / 16;
  cv4d blk[MAXQ];
  int bI[MAXQ], bJ[MAXQ];
#pragma unroll
  for (int q = 0; q < MAXQ; q++) {
    const int e = q * NW + w;
    int J = 0, st = 0;
    while (J < nb && e >= st + nb - J) { st += nb - J; J++; }
    bI[q] = e < nblk ? J + e - st : -1;  // (-1: no block)
    bJ[q] = e < nblk ? J : -1;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int R = 16 * bI[q] + g + 4 * r, C = 16 * bJ[q] + c;
      double v = 0.0;
      if (bI[q] >= 0 && C < nc) {
        if (R < nc && C <= R) v = Hc[C + (size_t)nc * R];  // lower triangle of L from the UPPER triangle of Hc
        else if (R == nc) v = -gc[C];
      }
      blk[q][r] = v;
    }
  }
  for (int e = tid; e < NCP; e += NTH) zs[e] = 0.0;
  for (int p = 0; p < npan; p++) {
    const int kb = 16 * p, bs = min(16, nc - kb);
    // (A) the panel's blocks -> Praw
#pragma unroll
    for (int q = 0; q < MAXQ; q++)
      if (bJ[q] == p) {
        const int I = launder_s(bI[q]);
#pragma unroll
        for (int r = 0; r < 4; r++) Praw[(16 * I + g + 4 * r) * LD + c] = blk[q][r];
      }
    __syncthreads();
    // (B) diagonal block: one wave, lane l = row kb + l; then its inverse, lane l = column l, the factor read back from LDS
    if (tid < 64) cons_diag_block(Praw + kb * LD, Dg, Di + p * 272, bs, fail);
    __syncthreads();
    // (C) the panel's blocks: X = A L_kk^-T (the diagonal block's own rows: the factor), published to P
#pragma unroll
    for (int q = 0; q < MAXQ; q++)
      if (bJ[q] == p) {
        const int I = launder_s(bI[q]);
        cv4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 4; kk++)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Praw[(16 * I + c) * LD + 4 * kk + g], Di[p * 272 + c * 17 + 4 * kk + g], acc, 0, 0, 0);
        const int rr = nc - 16 * I;  // (the right-hand side's row inside this block row, if 0 <= rr < 16)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          if (I == p && g + 4 * r < bs) acc[r] = (c <= g + 4 * r) ? Dg[c * 16 + g * 4 + r] : 0.0;
          P[(16 * I + g + 4 * r) * LD + c] = acc[r];
          if (g + 4 * r == rr && c < bs) zs[kb + c] = acc[r];
        }
        blk[q] = acc;
        __builtin_amdgcn_sched_barrier(0);  // (without it the loads of every slot are hoisted to the top: 256 registers and spills)
      }
    __syncthreads();
    // (D) trailing blocks: rank-16 update
#pragma unroll
    for (int q = 0; q < MAXQ; q++)
      if (bJ[q] > p) {
        const int I = launder_s(bI[q]), J = launder_s(bJ[q]);
#pragma unroll
        for (int kk = 0; kk < 4; kk++)
          blk[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(-P[(16 * I + c) * LD + 4 * kk + g], P[(16 * J + c) * LD + 4 * kk + g], blk[q], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    // (the next panel's (A) writes Praw, last read in (C); its (B) and (C) write Dg / Di / P after the barrier that follows (A))
  }
  // the factor -> global (column-major lower triangle, nc x nc)
#pragma unroll
  for (int q = 0; q < MAXQ; q++)
    if (bI[q] >= 0) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int R = 16 * bI[q] + g + 4 * r, C = 16 * bJ[q] + c;
        if (R < nc && C <= R) Lc[R + (size_t)nc * C] = blk[q][r];
      }
    }
  for (int e = tid; e < npan * 272; e += NTH) Lc[(size_t)nc * nc + e] = Di[e];  // (the inverse diagonal blocks where k_cons_solve_lds keeps them: its vector-only solves may follow)
  // L' x = z, right-looking over the block rows from the bottom
  for (int p = npan - 1; p >= 0; p--) {
    const int kb = 16 * p, bs = min(16, nc - kb);
    double v = 0.0;
    if (tid < bs) {
#pragma unroll
      for (int k = 0; k < 16; k++)
        if (k >= tid && k < bs) v = fma(Di[p * 272 + k * 17 + tid], zs[kb + k], v);  // (L_kk^-T)[tid][k] = Linv[k][tid]
    }
    __syncthreads();
    if (tid < bs) zs[kb + tid] = v;
    __syncthreads();
    if (kb > 0) {
#pragma unroll
      for (int q = 0; q < MAXQ; q++)
        if (bI[q] == p && bJ[q] < p) {
          double s_ = 0.0;
#pragma unroll
          for (int r = 0; r < 4; r++) s_ = fma(blk[q][r], zs[kb + g + 4 * r], s_);  // (rows >= nc: zs = 0)
          part[g * NCP + 16 * launder_s(bJ[q]) + c] = s_;
        }
      __syncthreads();
      for (int e = tid; e < kb; e += NTH) zs[e] -= (part[e] + part[NCP + e]) + (part[2 * NCP + e] + part[3 * NCP + e]);
      __syncthreads();
    }
  }
  for (int e = tid; e < nc; e += NTH) duc[e] = zs[e];
}

__global__ void __launch_bounds__(1024) k_cons_small(const double *Hc_part, const double *gc_part, int M, int nc, int with_H,
                                                     double *outH, double *outg, int solve_now, double *Lc, double *duc,
                                                     int *fail, AsCtlCall pend) {
  __shared__ double red[32][33];
  __shared__ double y[8];
  // one block more than partials when a round-control call rides along: it takes the decision about the PREVIOUS round
  // (counters of its forward sweep -> done / status / tolerance of the next forward sweep), see as_ctl_dev.h
  const int nblk = pend.ctl ? (int)gridDim.x - 1 : (int)gridDim.x;
  if (pend.ctl && (int)blockIdx.x == nblk) {
    as_ctl_block(pend.ctl, pend.cnt_part, pend.M, pend.fail, 1, 1, 0, pend.mirror, pend.mirror_seq, pend.seq, nullptr, pend.viol, pend.open_part);
    if (pend.perm) {  // (uniform) the launch order of this round's forward sweep, from the flags the round before left
      __syncthreads();
      as_perm_block(pend.settled, pend.M, pend.perm);
    }
    return;
  }
  const int tid = threadIdx.x, e = tid & 31, pl = tid >> 5;
  const int nH = nc * nc, E = with_H ? nH + nc : nc;
  const int per = (M + nblk - 1) / nblk, i0 = blockIdx.x * per, i1 = min(M, i0 + per);
  const bool isH = with_H && e < nH;
  double acc = 0.0;
  if (e < E) {
    const double *src = isH ? Hc_part + e : gc_part + (with_H ? e - nH : e);
    const int stride = isH ? nH : nc;
    // independent accumulators: the loads of a lane are all in flight together (a single dependent chain of 16 loads of
    // ~1 us each was most of this kernel's 15 us at 512 particles); summation order is fixed, so the result stays deterministic
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int i = i0 + pl;
    for (; i + 96 < i1; i += 128) {
      a0 += src[(size_t)i * stride];
      a1 += src[(size_t)(i + 32) * stride];
      a2 += src[(size_t)(i + 64) * stride];
      a3 += src[(size_t)(i + 96) * stride];
    }
    for (; i < i1; i += 32) a0 += src[(size_t)i * stride];
    acc = (a0 + a1) + (a2 + a3);
  }
  red[pl][e] = acc;
  __syncthreads();
  if (pl == 0 && e < E) {
    double t = 0.0;
    for (int k = 0; k < 32; k++) t += red[k][e];
    if (isH) outH[(size_t)blockIdx.x * nH + e] = t;
    else outg[(size_t)blockIdx.x * nc + (with_H ? e - nH : e)] = t;
  }
  __syncthreads();
  if (solve_now && tid == 0) {
    const double *H = outH, *g = outg;
    if (with_H) {
      for (int q = 0; q < nc; q++) {
        double d = H[q + nc * q];
        for (int k = 0; k < q; k++) d -= Lc[q + nc * k] * Lc[q + nc * k];
        if (!(d > 0.0)) { *fail = 2; d = 1.0; }
        d = sqrt(d);
        Lc[q + nc * q] = d;
        for (int p = q + 1; p < nc; p++) {
          double v = H[q + nc * p];  // upper triangle (the only one the fast path fills)
          for (int k = 0; k < q; k++) v -= Lc[p + nc * k] * Lc[q + nc * k];
          Lc[p + nc * q] = v / d;
        }
      }
    }
    for (int p = 0; p < nc; p++) {
      double v = -g[p];
      for (int k = 0; k < p; k++) v -= Lc[p + nc * k] * y[k];
      y[p] = v / Lc[p + nc * p];
    }
    for (int p = nc - 1; p >= 0; p--) {
      double v = y[p];
      for (int k = p + 1; k < nc; k++) v -= Lc[k + nc * p] * y[k];
      y[p] = v / Lc[p + nc * p];
    }
    for (int p = 0; p < nc; p++) duc[p] = y[p];
  }
}

}  // namespace

void launch_cons_small(const double *Hc_part, const double *gc_part, int M, int nc, bool with_H, double *Hg, double *tmp,
                       bool solve_now, double *Lc, double *duc, int *fail, hipStream_t s) {
  const int nH = nc * nc;
  int G = (M + 127) / 128;  // ~4 particles per lane and slice
  if (G > 64) G = 64;
  if (M <= 512) G = 1;  // small shards are launch-bound: one block sums <= 32 particles per lane and solves, one launch instead of two
  double *Hc = Hg, *gc = Hg + nH;
  if (G <= 1) {
    hipLaunchKernelGGL(k_cons_small, dim3(1), dim3(1024), 0, s, Hc_part, gc_part, M, nc, with_H ? 1 : 0, Hc, gc, solve_now ? 1 : 0,
                       Lc, duc, fail, AsCtlCall{});
    return;
  }
  double *tH = tmp, *tg = tmp + (size_t)64 * nH;
  hipLaunchKernelGGL(k_cons_small, dim3(G), dim3(1024), 0, s, Hc_part, gc_part, M, nc, with_H ? 1 : 0, tH, tg, 0, Lc, duc, fail, AsCtlCall{});
  hipLaunchKernelGGL(k_cons_small, dim3(1), dim3(1024), 0, s, (const double *)tH, (const double *)tg, G, nc, with_H ? 1 : 0, Hc, gc,
                     solve_now ? 1 : 0, Lc, duc, fail, AsCtlCall{});
}

// first stage alone: block partials [G][nc*nc] at tmp, [G][nc] at tmp + 64 nc^2 (the consumers sum them in block order)
int launch_cons_partials(const double *Hc_part, const double *gc_part, int M, int nc, double *tmp, const AsCtlCall &pend, hipStream_t s) {
  const int nH = nc * nc;
  static const int per = getenv("PMPC_CONS_PER_PARTIAL") ? atoi(getenv("PMPC_CONS_PER_PARTIAL")) : 256;  // particles per block partial
  int G = (M + per - 1) / per;
  if (G > 64) G = 64;
  hipLaunchKernelGGL(k_cons_small, dim3(G + (pend.ctl ? 1 : 0)), dim3(1024), 0, s, Hc_part, gc_part, M, nc, 1, tmp, tmp + (size_t)64 * nH, 0,
                     nullptr, nullptr, nullptr, pend);
  return G;
}

// Consensus weights (cone objective, solver.hip lcone_body): the sweeps run UNWEIGHTED — scaling a particle's whole cost leaves its
// gains, its active set and its conditional optimum given the shared controls unchanged — and the weights lambda_i enter only where
// the particles meet:  sum_i lambda_i (H_i, g_i).  One elementwise pass writes lambda_i H_i, lambda_i g_i for the reductions; the terms
// of the SHARED controls' box, which the owner's particle 0 carries on its diagonal / in its gradient (the 1e30 penalty of a held
// control in the active-set rounds, the barrier diagonal and shift Du / wu in the interior-point sweeps), are not part of its cost
// and keep weight one.
__global__ void __launch_bounds__(256) k_cons_scale(const double *Hc_part, const double *gc_part, const double *w, int M, int nc, int with_H,
                                                    double *outH, double *outg, const int *as_act, double as_big, const double *Du,
                                                    const double *wu, int u, int owner) {
  const int nH = nc * nc, E = with_H ? nH + nc : nc;
  const long long tot = (long long)M * E;
  for (long long k = blockIdx.x * 256ll + threadIdx.x; k < tot; k += (long long)gridDim.x * 256) {
    const int i = (int)(k / E), e = (int)(k % E);
    const bool isH = with_H && e < nH;
    const int r = isH ? e % nc : (with_H ? e - nH : e), cc = isH ? e / nc : -1;
    const double v = isH ? Hc_part[(size_t)i * nH + e] : gc_part[(size_t)i * nc + r];
    double keep = 0.0;  // the part of v that is not the particle's own cost
    if (i == 0 && owner) {
      // (stage j = r / u, control r % u of particle 0: index (0 * N + j) * u + r % u = r)
      if (isH && cc == r) keep = as_act ? (as_act[r] ? as_big : 0.0) : (Du ? Du[r] : 0.0);
      if (!isH && !as_act && wu) keep = wu[r];
    }
    const double o = fma(w[i], v - keep, keep);
    if (isH) outH[(size_t)i * nH + e] = o;
    else outg[(size_t)i * nc + r] = o;
  }
}
void launch_cons_scale(const double *Hc_part, const double *gc_part, const double *w, int M, int nc, bool with_H, double *outH, double *outg,
                       const int *as_act, double as_big, const double *Du, const double *wu, int u, int owner, hipStream_t s) {
  const long long tot = (long long)M * (with_H ? nc * nc + nc : nc);
  const long long blocks = (tot + 255) / 256;
  hipLaunchKernelGGL(k_cons_scale, dim3((unsigned)(blocks < 2048 ? (blocks > 0 ? blocks : 1) : 2048)), dim3(256), 0, s, Hc_part, gc_part, w, M, nc,
                     with_H ? 1 : 0, outH, outg, as_act, as_big, Du, wu, u, owner);
}

// KKT check of the epigraph rows of the cone objective (solver.hip lcone_body) on the device: multipliers lam_i in [0, cap], particle
// costs J_i (scaled by the caller's weights if any).  Threshold cost t = mean cost of the rows strictly inside (0, cap) (none: the midpoint
// between the cheapest full row and the costliest empty one); violation = max over rows of  |J_i - t| (inside), t - J_i (full),
// J_i - t (empty).  out = {violation, t, rows inside, 0}.  One block.
__global__ void __launch_bounds__(1024) k_epi_check(const double *lam, const double *J, const double *user, int M, double cap, double *out,
                                                   double *mirror, unsigned long long *mirror_seq, unsigned long long seq) {
  __shared__ double rs[1024], rmin[1024], rmax[1024];
  __shared__ int rn[1024];
  __shared__ double tsh;
  const int t = threadIdx.x;
  double s = 0.0, jmin = 1e300, jmax = -1e300;
  int n = 0;
  for (int i = t; i < M; i += 1024) {
    const double l = lam[i], j = J[i] * (user ? user[i] : 1.0);
    if (l > 1e-12 && l < cap - 1e-12) { s += j; n++; }
    else if (l >= cap - 1e-12) jmin = fmin(jmin, j);
    else jmax = fmax(jmax, j);
  }
  rs[t] = s; rn[t] = n; rmin[t] = jmin; rmax[t] = jmax;
  __syncthreads();
  for (int w = 512; w > 0; w >>= 1) {
    if (t < w) { rs[t] += rs[t + w]; rn[t] += rn[t + w]; rmin[t] = fmin(rmin[t], rmin[t + w]); rmax[t] = fmax(rmax[t], rmax[t + w]); }
    __syncthreads();
  }
  if (t == 0) {
    const double a = rmin[0], b = rmax[0];
    tsh = rn[0] ? rs[0] / (double)rn[0] : ((a < 1e300 && b > -1e300) ? 0.5 * (a + b) : (a < 1e300 ? a : b));
  }
  __syncthreads();
  const double th = tsh;
  double v = 0.0;
  bool bad = false;
  for (int i = t; i < M; i += 1024) {
    const double l = lam[i], j = J[i] * (user ? user[i] : 1.0);
    bad |= !(j == j);
    if (l > 1e-12 && l < cap - 1e-12) v = fmax(v, fabs(j - th));
    else if (l >= cap - 1e-12) v = fmax(v, th - j);
    else v = fmax(v, j - th);
  }
  const int nin = rn[0];
  __syncthreads();
  rs[t] = bad ? 1e300 : v;
  __syncthreads();
  for (int w = 512; w > 0; w >>= 1) {
    if (t < w) rs[t] = fmax(rs[t], rs[t + w]);
    __syncthreads();
  }
  if (t == 0) {
    out[0] = rs[0]; out[1] = th; out[2] = (double)nin; out[3] = 0.0;
    if (mirror) {  // host-coherent copy + sequence number: the host polls it without draining the stream (work enqueued behind keeps running)
      mirror[0] = rs[0]; mirror[1] = th; mirror[2] = (double)nin; mirror[3] = 0.0;
      __threadfence_system();
      *(volatile unsigned long long *)mirror_seq = seq;
    }
  }
}
void launch_epi_check(const double *lam, const double *J, const double *user, int M, double cap, double *out, hipStream_t s, double *mirror,
                      unsigned long long *mirror_seq, unsigned long long seq) {
  hipLaunchKernelGGL(k_epi_check, dim3(1), dim3(1024), 0, s, lam, J, user, M, cap, out, mirror, mirror_seq, seq);
}

size_t lq_generic_lds_bytes(const LQArgs &a) { return lds_doubles(a.x, a.u, a.n) * sizeof(double); }

void launch_rollout(const LQArgs &a, const double *U, double *X, hipStream_t s) {
  hipLaunchKernelGGL(k_rollout, dim3(a.M), dim3(WV), (a.x + a.u) * sizeof(double), s, a, U, X);
}

void launch_bwd_generic(const LQArgs &a, bool factor, hipStream_t s) {
  size_t lds = lq_generic_lds_bytes(a);
  if (factor) hipLaunchKernelGGL(k_bwd_generic<true>, dim3(a.M), dim3(WV), lds, s, a);
  else hipLaunchKernelGGL(k_bwd_generic<false>, dim3(a.M), dim3(WV), lds, s, a);
}

void launch_fwd_generic(const LQArgs &a, hipStream_t s) {
  hipLaunchKernelGGL(k_fwd_generic, dim3(a.M), dim3(WV), (2 * a.n + a.u) * sizeof(double), s, a);
}

// [Hc | gc] in one pass: entries e < EH come from srcH (stride EH), the rest from srcG (stride EG); dst = [G][EH + EG]
__global__ void __launch_bounds__(256) k_reduce_particles_hg(const double *srcH, const double *srcG, double *dst, int MH, int M, int EH, int EG) {
  __shared__ double red[4][64];
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int e = blockIdx.x * 64 + tx, E = EH + EG;
  double acc = 0.0;
  // (H entries strictly below the diagonal are never read by the consensus solve — it takes the UPPER triangle, the only one the
  //  condensing kernel fills —: skipping them halves this kernel's traffic, M (Nc u)^2 doubles at full consensus)
  const bool lower = e < EH && EG > 0 && (e % EG) > (e / EG);
  if (e < E && !lower) {
    const double *src = e < EH ? srcH + e : srcG + (e - EH);
    const int stride = e < EH ? EH : EG;
    const int Mi = e < EH ? MH : M;  // (the H slabs may be group sums: fewer of them than particles)
    for (int i = blockIdx.y * 4 + ty; i < Mi; i += 4 * gridDim.y) acc += src[(size_t)i * stride];
  }
  red[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && e < E) dst[(size_t)blockIdx.y * E + e] = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
}
// Hg = [Hc | gc] (contiguous, as the all-reduce wants it) from the particle partials: two launches instead of four
void launch_reduce_particles_hg(const double *Hc_part, const double *gc_part, double *tmp, double *Hg, int M, int nc, hipStream_t s, int MH) {
  if (MH < 0) MH = M;
  const int EH = nc * nc, E = EH + nc;
  int gy = (M + 3) / 4;
  if (gy > 64) gy = 64;
  dim3 blk(64, 4), grd((E + 63) / 64, gy);
  if (gy == 1) {
    hipLaunchKernelGGL(k_reduce_particles_hg, grd, blk, 0, s, Hc_part, gc_part, Hg, MH, M, EH, nc);
  } else {
    hipLaunchKernelGGL(k_reduce_particles_hg, grd, blk, 0, s, Hc_part, gc_part, tmp, MH, M, EH, nc);
    hipLaunchKernelGGL(k_reduce_particles, dim3((E + 63) / 64, 1), blk, 0, s, (const double *)tmp, Hg, gy, E);
  }
}
void launch_reduce_particles(const double *src, double *tmp, double *dst, int M, int E, hipStream_t s) {
  int gy = (M + 3) / 4;
  if (gy > 64) gy = 64;
  dim3 blk(64, 4), grd((E + 63) / 64, gy);
  if (gy == 1) {
    hipLaunchKernelGGL(k_reduce_particles, grd, blk, 0, s, src, dst, M, E);
  } else {
    hipLaunchKernelGGL(k_reduce_particles, grd, blk, 0, s, src, tmp, M, E);
    hipLaunchKernelGGL(k_reduce_particles, dim3((E + 63) / 64, 1), blk, 0, s, (const double *)tmp, dst, gy, E);
  }
}

void launch_cons_solve(double *Hc, double *Lc, const double *gc, double *duc, int nc, bool factor, int *fail,
                       hipStream_t s) {
  static const bool reg_on = !(getenv("PMPC_CONS_REG") && atoi(getenv("PMPC_CONS_REG")) == 0);
  // (117 KB of dynamic LDS: above the 64 KB a kernel gets without asking — where the runtime refuses, the older variants below serve)
  static const bool reg_lds_ok =
      hipFuncSetAttribute(reinterpret_cast<const void *>(k_cons_solve_reg<12>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CONS_REG_LDS) == hipSuccess &&
      hipFuncSetAttribute(reinterpret_cast<const void *>(k_cons_solve_reg<17>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CONS_REG_LDS) == hipSuccess;
  if (reg_on && reg_lds_ok && factor && cons_reg_fits(nc)) {
    const int nb = (nc + 1 + 15) / 16, slots = (nb * (nb + 1) / 2 + CONS_REG_NW - 1) / CONS_REG_NW;
    if (slots <= 12) hipLaunchKernelGGL(k_cons_solve_reg<12>, dim3(1), dim3(CONS_REG_NTH), CONS_REG_LDS, s, (const double *)Hc, Lc, gc, duc, nc, fail);
    else hipLaunchKernelGGL(k_cons_solve_reg<17>, dim3(1), dim3(CONS_REG_NTH), CONS_REG_LDS, s, (const double *)Hc, Lc, gc, duc, nc, fail);
    return;
  }
  // LDS-resident factor (+ inverse diagonal blocks + a 16-column scratch panel): up to nc = 80 inside 64 KB
  const int npan = (nc + 15) / 16;
  const size_t lds_full = ((size_t)nc * nc + nc + 16 * 17 + (size_t)npan * 272 + (size_t)16 * nc) * sizeof(double);
  static const bool lds_on = !(getenv("PMPC_CONS_LDS") && atoi(getenv("PMPC_CONS_LDS")) == 0);
  if (lds_on && nc > 16 && lds_full <= 64 * 1024) {
    hipLaunchKernelGGL(k_cons_solve_lds, dim3(1), dim3(1024), lds_full, s, (const double *)Hc, Lc, gc, duc, nc, factor ? 1 : 0, fail);
    return;
  }
  const size_t lds = ((size_t)17 * nc + 16 * 17) * sizeof(double);
  if (nc > 16 && lds <= 64 * 1024)
    hipLaunchKernelGGL(k_cons_solve_blocked, dim3(1), dim3(PMPC_CONS_BLOCKED_THREADS), lds, s, (const double *)Hc, Lc, gc, duc, nc, factor ? 1 : 0, fail);
  else
    hipLaunchKernelGGL(k_cons_solve, dim3(1), dim3(256), nc * sizeof(double), s, (const double *)Hc, Lc, gc, duc, nc,
                       factor ? 1 : 0, fail);
}
