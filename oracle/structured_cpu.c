/* TEST / MEASUREMENT INFRASTRUCTURE — never linked into libpmpc_hip.so, never called by pmpc_amd.
 *
 * A multi-core CPU implementation of the STRUCTURED algorithm the HIP path uses (per-particle Riccati recursion, condensing of
 * the Nc consensus stages, Mehrotra predictor-corrector on the boxes; the numpy model is tests/support/structured_np.py):
 * OpenMP over particles.  It is the "second, stronger CPU line" SURVEY.md section 8(d) asks for next to the
 * reference-shaped baseline (sparse assembly + ADMM, oracle/lqp_oracle.py): the same QP of PMPC.jl/src/lqp_utils.jl:2-393
 * solved by the best CPU algorithm we know, so that the GPU/CPU ratio is not flattered by the reference's generic solver.
 * tests/test_oracle_golden.py checks it against the oracle's exact solve.
 *
 * Scope: every ABI feature except the slew terms (slew_reg / slew_reg0 => return -2).
 * Layout: the C ABI's (include/pmpc_abi.h): vectors (M, N, d) C-order, matrices (M, N) stacks of column-major blocks.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MX 16 /* max xdim */
#define MU 8  /* max udim */

typedef struct {
  int M, N, x, u, Nc, nc;
  const double *f, *fx, *fu, *X_prev, *U_prev, *Q, *R, *X_ref, *U_ref;
  double reg_x, reg_u;
  /* factorisation */
  double *K, *Hinv, *Phi, *Hc, *kff;
} LQ;

/* effective symmetric cost block: OSQP keeps triu(P) (lqp_utils.jl:130-141 hands the full block over) */
static inline double triu_sym(const double *B, int d, int r, int c) { return r <= c ? B[r + d * c] : B[c + d * r]; }

static int chol_inplace(double *A, int n) { /* lower Cholesky, column-major n x n; returns 0 on success */
  for (int q = 0; q < n; q++) {
    double v = A[q + n * q];
    for (int k = 0; k < q; k++) v -= A[q + n * k] * A[q + n * k];
    if (!(v > 0.0)) return 1;
    v = sqrt(v);
    A[q + n * q] = v;
    for (int p = q + 1; p < n; p++) {
      double w = A[p + n * q];
      for (int k = 0; k < q; k++) w -= A[p + n * k] * A[q + n * k];
      A[p + n * q] = w / v;
    }
  }
  return 0;
}

static void chol_solve_vec(const double *L, int n, double *y) {
  for (int p = 0; p < n; p++) {
    double v = y[p];
    for (int k = 0; k < p; k++) v -= L[p + n * k] * y[k];
    y[p] = v / L[p + n * p];
  }
  for (int p = n - 1; p >= 0; p--) {
    double v = y[p];
    for (int k = p + 1; k < n; k++) v -= L[k + n * p] * y[k];
    y[p] = v / L[p + n * p];
  }
}

/* Riccati factorisation of one particle; Dx (N, x) / Du (N, u) extra diagonals of this particle or NULL.
 * Hc_acc (nc x nc) accumulates this particle's condensed Hessian. */
static int factor_particle(LQ *q, int i, const double *Dx, const double *Du, double *Hc_acc) {
  const int N = q->N, x = q->x, u = q->u, Nc = q->Nc, nc = q->nc;
  double S[MX * MX], SA[MX * MX], SB[MX * MU], Hxx[MX * MX], Hux[MU * MX], Huu[MU * MU], Hi[MU * MU], Kj[MU * MX];
  const size_t pb = (size_t)i * N;
  const double *Qn = q->Q + (pb + N - 1) * x * x;
  for (int c = 0; c < x; c++)
    for (int r = 0; r < x; r++) S[r + x * c] = triu_sym(Qn, x, r, c) + (r == c ? q->reg_x + (Dx ? Dx[(size_t)(N - 1) * x + r] : 0.0) : 0.0);
  for (int j = N - 1; j >= Nc; j--) {
    const double *A = q->fx + (pb + j) * x * x, *B = q->fu + (pb + j) * x * u, *Rj = q->R + (pb + j) * u * u;
    const int hasA = j > 0;
    if (hasA)
      for (int c = 0; c < x; c++)
        for (int r = 0; r < x; r++) {
          double v = 0.0;
          for (int k = 0; k < x; k++) v += S[r + x * k] * A[k + x * c];
          SA[r + x * c] = v;
        }
    for (int c = 0; c < u; c++)
      for (int r = 0; r < x; r++) {
        double v = 0.0;
        for (int k = 0; k < x; k++) v += S[r + x * k] * B[k + x * c];
        SB[r + x * c] = v;
      }
    for (int c = 0; c < x; c++) {
      for (int r = 0; r < x; r++) {
        double v = 0.0;
        if (hasA)
          for (int k = 0; k < x; k++) v += A[k + x * r] * SA[k + x * c];
        Hxx[r + x * c] = v;
      }
      for (int r = 0; r < u; r++) {
        double v = 0.0;
        if (hasA)
          for (int k = 0; k < x; k++) v += B[k + x * r] * SA[k + x * c];
        Hux[r + u * c] = v;
      }
    }
    for (int c = 0; c < u; c++)
      for (int r = 0; r < u; r++) {
        double v = triu_sym(Rj, u, r, c) + (r == c ? q->reg_u + (Du ? Du[(size_t)j * u + r] : 0.0) : 0.0);
        for (int k = 0; k < x; k++) v += B[k + x * r] * SB[k + x * c];
        Huu[r + u * c] = v;
      }
    if (chol_inplace(Huu, u)) return 1;
    for (int c = 0; c < u; c++) { /* Huu^-1 */
      double e[MU];
      for (int r = 0; r < u; r++) e[r] = r == c ? 1.0 : 0.0;
      chol_solve_vec(Huu, u, e);
      for (int r = 0; r < u; r++) Hi[r + u * c] = e[r];
    }
    for (int c = 0; c < x; c++) { /* K = Huu^-1 Hux */
      double e[MU];
      for (int r = 0; r < u; r++) e[r] = Hux[r + u * c];
      chol_solve_vec(Huu, u, e);
      for (int r = 0; r < u; r++) Kj[r + u * c] = e[r];
    }
    memcpy(q->K + (pb + j) * u * x, Kj, sizeof(double) * u * x);
    memcpy(q->Hinv + (pb + j) * u * u, Hi, sizeof(double) * u * u);
    for (int c = 0; c < x; c++)
      for (int r = 0; r <= c; r++) {
        double v = Hxx[r + x * c];
        for (int k = 0; k < u; k++) v -= Hux[k + u * r] * Kj[k + u * c];
        S[r + x * c] = v;
      }
    const double *Qp = j > 0 ? q->Q + (pb + j - 1) * x * x : NULL;
    for (int c = 0; c < x; c++)
      for (int r = 0; r <= c; r++) {
        double v = S[r + x * c];
        if (Qp) v += triu_sym(Qp, x, r, c) + (r == c ? q->reg_x + (Dx ? Dx[(size_t)(j - 1) * x + r] : 0.0) : 0.0);
        S[r + x * c] = S[c + x * r] = v;
      }
  }
  if (Nc > 0) { /* forward sensitivities Phi_j = A_j Phi_{j-1} + B_j E_j and Hc += Phi_j' M_j Phi_j + R~_j */
    double *Phi = q->Phi + (size_t)i * Nc * x * nc;
    double Mj[MX * MX];
    double *T = (double *)malloc(sizeof(double) * x * nc);
    for (int j = 0; j < Nc; j++) {
      const double *A = q->fx + (pb + j) * x * x, *B = q->fu + (pb + j) * x * u, *Rj = q->R + (pb + j) * u * u;
      double *P = Phi + (size_t)j * x * nc;
      const double *Pp = j > 0 ? P - (size_t)x * nc : NULL;
      for (int c = 0; c < nc; c++)
        for (int r = 0; r < x; r++) {
          double v = 0.0;
          if (Pp)
            for (int k = 0; k < x; k++) v += A[r + x * k] * Pp[k + x * c];
          if (c / u == j) v += B[r + x * (c - j * u)];
          P[r + x * c] = v;
        }
      if (j == Nc - 1)
        memcpy(Mj, S, sizeof(double) * x * x);
      else {
        const double *Qj = q->Q + (pb + j) * x * x;
        for (int c = 0; c < x; c++)
          for (int r = 0; r < x; r++)
            Mj[r + x * c] = triu_sym(Qj, x, r, c) + (r == c ? q->reg_x + (Dx ? Dx[(size_t)j * x + r] : 0.0) : 0.0);
      }
      const int cols = (j + 1) * u; /* Phi_j is zero beyond the controls applied so far */
      for (int c = 0; c < cols; c++)
        for (int r = 0; r < x; r++) {
          double v = 0.0;
          for (int k = 0; k < x; k++) v += Mj[r + x * k] * P[k + x * c];
          T[r + x * c] = v;
        }
      for (int c = 0; c < cols; c++)
        for (int r = 0; r < cols; r++) {
          double v = 0.0;
          for (int k = 0; k < x; k++) v += P[k + x * r] * T[k + x * c];
          Hc_acc[r + nc * c] += v;
        }
      for (int c = 0; c < u; c++)
        for (int r = 0; r < u; r++) Hc_acc[(j * u + r) + nc * (j * u + c)] += triu_sym(Rj, u, r, c) + (r == c ? q->reg_u : 0.0);
    }
    /* the state cost of stage Nc-1 is inside S (added after the last free stage) */
    free(T);
  }
  return 0;
}

static int lq_factor(LQ *q, const double *Dx, const double *Du, const double *Dc) {
  const int M = q->M, N = q->N, x = q->x, u = q->u, nc = q->nc;
  int bad = 0;
  if (nc) memset(q->Hc, 0, sizeof(double) * nc * nc);
#pragma omp parallel
  {
    double *acc = nc ? (double *)calloc((size_t)nc * nc, sizeof(double)) : NULL;
#pragma omp for schedule(static) reduction(| : bad)
    for (int i = 0; i < M; i++)
      bad |= factor_particle(q, i, Dx ? Dx + (size_t)i * N * x : NULL, Du ? Du + (size_t)i * N * u : NULL, acc);
    if (nc) {
#pragma omp critical
      for (int k = 0; k < nc * nc; k++) q->Hc[k] += acc[k];
      free(acc);
    }
  }
  if (bad) return 1;
  if (nc) {
    if (Dc)
      for (int k = 0; k < nc; k++) q->Hc[k + nc * k] += Dc[k];
    if (chol_inplace(q->Hc, nc)) return 1;
  }
  return 0;
}

/* Newton step of the equality-constrained QP: gx (M,N,x), gu (M,N,u) per-particle gradients; gce (nc) added once */
static void lq_solve(LQ *q, const double *gx, const double *gu, const double *gce, double *dX, double *dU, double *s_end) {
  const int M = q->M, N = q->N, x = q->x, u = q->u, Nc = q->Nc, nc = q->nc;
  double *gsum = nc ? (double *)calloc(nc, sizeof(double)) : NULL;
#pragma omp parallel
  {
    double *acc = nc ? (double *)calloc(nc, sizeof(double)) : NULL;
#pragma omp for schedule(static)
    for (int i = 0; i < M; i++) {
      const size_t pb = (size_t)i * N;
      double s[MX], hx[MX], hu[MU];
      for (int r = 0; r < x; r++) s[r] = gx[(pb + N - 1) * x + r];
      for (int j = N - 1; j >= Nc; j--) {
        const double *A = q->fx + (pb + j) * x * x, *B = q->fu + (pb + j) * x * u;
        const double *Kj = q->K + (pb + j) * u * x, *Hi = q->Hinv + (pb + j) * u * u;
        for (int c = 0; c < x; c++) {
          double v = 0.0;
          if (j > 0)
            for (int k = 0; k < x; k++) v += A[k + x * c] * s[k];
          hx[c] = v;
        }
        for (int c = 0; c < u; c++) {
          double v = gu[(pb + j) * u + c];
          for (int k = 0; k < x; k++) v += B[k + x * c] * s[k];
          hu[c] = v;
        }
        for (int r = 0; r < u; r++) {
          double v = 0.0;
          for (int c = 0; c < u; c++) v += Hi[r + u * c] * hu[c];
          q->kff[(pb + j) * u + r] = v;
        }
        for (int c = 0; c < x; c++) {
          double v = hx[c];
          for (int k = 0; k < u; k++) v -= Kj[k + u * c] * hu[k];
          s[c] = v + (j > 0 ? gx[(pb + j - 1) * x + c] : 0.0);
        }
      }
      if (Nc > 0) {
        const double *Phi = q->Phi + (size_t)i * Nc * x * nc;
        for (int j = 0; j < Nc; j++) {
          const double *P = Phi + (size_t)j * x * nc;
          const double *m = (j == Nc - 1) ? s : gx + (pb + j) * x;
          for (int c = 0; c < (j + 1) * u; c++) {
            double v = 0.0;
            for (int k = 0; k < x; k++) v += P[k + x * c] * m[k];
            acc[c] += v;
          }
          for (int c = 0; c < u; c++) acc[j * u + c] += gu[(pb + j) * u + c];
        }
      }
    }
    if (nc) {
#pragma omp critical
      for (int k = 0; k < nc; k++) gsum[k] += acc[k];
      free(acc);
    }
  }
  if (nc) {
    if (gce)
      for (int k = 0; k < nc; k++) gsum[k] += gce[k];
    chol_solve_vec(q->Hc, nc, gsum);
  }
#pragma omp parallel for schedule(static)
  for (int i = 0; i < M; i++) {
    const size_t pb = (size_t)i * N;
    double xi[MX], xn[MX], du[MU];
    for (int r = 0; r < x; r++) xi[r] = 0.0;
    for (int j = 0; j < N; j++) {
      const double *A = q->fx + (pb + j) * x * x, *B = q->fu + (pb + j) * x * u;
      if (j < Nc) {
        for (int r = 0; r < u; r++) du[r] = -gsum[j * u + r];
      } else {
        const double *Kj = q->K + (pb + j) * u * x;
        for (int r = 0; r < u; r++) {
          double v = -q->kff[(pb + j) * u + r];
          for (int c = 0; c < x; c++) v -= Kj[r + u * c] * xi[c];
          du[r] = v;
        }
      }
      for (int r = 0; r < x; r++) {
        double v = 0.0;
        if (j > 0)
          for (int c = 0; c < x; c++) v += A[r + x * c] * xi[c];
        for (int c = 0; c < u; c++) v += B[r + x * c] * du[c];
        xn[r] = v;
      }
      for (int r = 0; r < x; r++) dX[(pb + j) * x + r] = xi[r] = xn[r];
      for (int r = 0; r < u; r++) dU[(pb + j) * u + r] = du[r];
    }
  }
  (void)s_end;
  free(gsum);
}

static void rollout(const LQ *q, const double *U, double *X) { /* types.jl:161-173 */
  const int M = q->M, N = q->N, x = q->x, u = q->u;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < M; i++) {
    const size_t pb = (size_t)i * N;
    for (int j = 0; j < N; j++) {
      const double *A = q->fx + (pb + j) * x * x, *B = q->fu + (pb + j) * x * u;
      for (int r = 0; r < x; r++) {
        double v = q->f[(pb + j) * x + r];
        for (int c = 0; c < u; c++) v += B[r + x * c] * (U[(pb + j) * u + c] - q->U_prev[(pb + j) * u + c]);
        if (j > 0)
          for (int c = 0; c < x; c++) v += A[r + x * c] * (X[(pb + j - 1) * x + c] - q->X_prev[(pb + j - 1) * x + c]);
        X[(pb + j) * x + r] = v;
      }
    }
  }
}

static void gradient(const LQ *q, const double *X, const double *U, double *gx, double *gu) {
  const int N = q->N, x = q->x, u = q->u;
  const long long MN = (long long)q->M * N;
#pragma omp parallel for schedule(static)
  for (long long b = 0; b < MN; b++) {
    const double *Qb = q->Q + b * x * x, *Rb = q->R + b * u * u;
    for (int r = 0; r < x; r++) {
      double v = q->reg_x * (X[b * x + r] - q->X_prev[b * x + r]);
      for (int c = 0; c < x; c++) v += triu_sym(Qb, x, r, c) * X[b * x + c] - Qb[r + x * c] * q->X_ref[b * x + c];
      gx[b * x + r] = v;
    }
    for (int r = 0; r < u; r++) {
      double v = q->reg_u * (U[b * u + r] - q->U_prev[b * u + r]);
      for (int c = 0; c < u; c++) v += triu_sym(Rb, u, r, c) * U[b * u + c] - Rb[r + u * c] * q->U_ref[b * u + c];
      gu[b * u + r] = v;
    }
  }
}

/* one group of boxed variables (states or controls) with its slacks and multipliers */
typedef struct {
  long long n;
  double *z, *dz;
  double *lo, *hi, *w; /* bounds (+-inf = absent), weight of the pair in sums (0 on non-owner copies of a consensus bound) */
  double *tl, *tu, *ll, *lu, *rl, *ru, *wl, *wu, *dtl, *dtu, *dll, *dlu, *cl, *cu;
} Grp;

static void grp_alloc(Grp *g, long long n) {
  memset(g, 0, sizeof(*g));
  g->n = n;
  double **f[] = {&g->lo, &g->hi, &g->w, &g->tl, &g->tu, &g->ll, &g->lu, &g->rl, &g->ru, &g->wl, &g->wu, &g->dtl, &g->dtu, &g->dll,
                  &g->dlu, &g->cl, &g->cu};
  for (unsigned k = 0; k < sizeof(f) / sizeof(f[0]); k++) *f[k] = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
}
static void grp_free(Grp *g) {
  double *f[] = {g->lo, g->hi, g->w, g->tl, g->tu, g->ll, g->lu, g->rl, g->ru, g->wl, g->wu, g->dtl, g->dtu, g->dll, g->dlu, g->cl, g->cu};
  for (unsigned k = 0; k < sizeof(f) / sizeof(f[0]); k++) free(f[k]);
}

/* returns 0 ok, 1 numerical failure, 2 iteration limit, -2 unsupported */
int structured_cpu_solve(int xdim, int udim, int N, int M, long long Nc_in, const double *x0, const double *f, const double *fx,
                         const double *fu, const double *X_prev, const double *U_prev, const double *Q, const double *R,
                         const double *X_ref, const double *U_ref, const double *lx, const double *ux, const double *lu,
                         const double *uu, double reg_x, double reg_u, int threads, double *X_out, double *U_out, int *iters_out) {
  (void)x0; /* lqp_utils.jl:288-296: x0 itself never enters the QP */
  if (xdim > MX || udim > MU) return -2;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#endif
  LQ q;
  memset(&q, 0, sizeof(q));
  const int x = xdim, u = udim, Nc = (Nc_in < 0 || Nc_in > N) ? N : (int)Nc_in, nc = Nc * u;
  q.M = M; q.N = N; q.x = x; q.u = u; q.Nc = Nc; q.nc = nc;
  q.f = f; q.fx = fx; q.fu = fu; q.X_prev = X_prev; q.U_prev = U_prev; q.Q = Q; q.R = R; q.X_ref = X_ref; q.U_ref = U_ref;
  q.reg_x = reg_x; q.reg_u = reg_u;
  const long long nX = (long long)M * N * x, nU = (long long)M * N * u;
  q.K = (double *)malloc(sizeof(double) * (size_t)M * N * u * x);
  q.Hinv = (double *)malloc(sizeof(double) * (size_t)M * N * u * u);
  q.kff = (double *)malloc(sizeof(double) * (size_t)nU);
  q.Phi = nc ? (double *)malloc(sizeof(double) * (size_t)M * Nc * x * nc) : NULL;
  q.Hc = nc ? (double *)malloc(sizeof(double) * (size_t)nc * nc) : NULL;
  double *X = X_out, *U = U_out;
  double *gx = (double *)malloc(sizeof(double) * nX), *gu = (double *)malloc(sizeof(double) * nU);
  double *gxx = (double *)malloc(sizeof(double) * nX), *guu = (double *)malloc(sizeof(double) * nU);
  double *dX = (double *)malloc(sizeof(double) * nX), *dU = (double *)malloc(sizeof(double) * nU);
  double *Dx = (double *)calloc(nX, sizeof(double)), *Du = (double *)calloc(nU, sizeof(double));
  double *Dc = nc ? (double *)calloc(nc, sizeof(double)) : NULL, *gce = nc ? (double *)calloc(nc, sizeof(double)) : NULL;
  int status = 0, it = 0;
  const int has_xb = lx && ux, has_ub = lu && uu;
  Grp G[2];
  grp_alloc(&G[0], has_xb ? nX : 0);
  grp_alloc(&G[1], has_ub ? nU : 0);
  G[0].z = X; G[0].dz = dX; G[1].z = U; G[1].dz = dU;

  /* ---- unconstrained Newton step from a dynamics-consistent base point (consensus controls at 0) ---- */
  for (long long k = 0; k < nU; k++) U[k] = ((k / u) % N) < Nc ? 0.0 : U_prev[k];
  rollout(&q, U, X);
  gradient(&q, X, U, gx, gu);
  if (lq_factor(&q, NULL, NULL, NULL)) { status = 1; goto done; }
  lq_solve(&q, gx, gu, NULL, dX, dU, NULL);
  for (long long k = 0; k < nX; k++) X[k] += dX[k];
  for (long long k = 0; k < nU; k++) U[k] += dU[k];
  if (!has_xb && !has_ub) goto done;
  /* bounds; consensus-control bounds from particle 0 (lqp_utils.jl:329-330) */
  if (has_xb)
    for (long long k = 0; k < nX; k++) { G[0].lo[k] = lx[k]; G[0].hi[k] = ux[k]; G[0].w[k] = 1.0; }
  if (has_ub)
    for (long long k = 0; k < nU; k++) {
      const long long j = (k / u) % N, i = k / ((long long)N * u), k0 = k - i * (long long)N * u;
      const int consj = j < Nc;
      G[1].lo[k] = consj ? lu[k0] : lu[k];
      G[1].hi[k] = consj ? uu[k0] : uu[k];
      G[1].w[k] = (consj && i > 0) ? 0.0 : 1.0;
    }
  {
    double viol = -INFINITY;
    for (int g = 0; g < 2; g++)
      for (long long k = 0; k < G[g].n; k++) {
        viol = fmax(viol, G[g].lo[k] - G[g].z[k]);
        viol = fmax(viol, G[g].z[k] - G[g].hi[k]);
      }
    if (viol <= 0.0) goto done;
  }
  double m_cnt = 0.0;
  for (int g = 0; g < 2; g++)
    for (long long k = 0; k < G[g].n; k++) m_cnt += G[g].w[k] * ((isfinite(G[g].lo[k]) ? 1.0 : 0.0) + (isfinite(G[g].hi[k]) ? 1.0 : 0.0));
  if (has_ub) { /* pull the controls strictly inside their box, re-roll the states */
    for (long long k = 0; k < nU; k++) {
      const double lo = G[1].lo[k], hi = G[1].hi[k];
      const int ml = isfinite(lo), mh = isfinite(hi);
      const double wid = (ml && mh) ? hi - lo : fmax(1.0, ml ? fabs(lo) : (mh ? fabs(hi) : 1.0));
      if (ml && U[k] < lo + 0.1 * wid) U[k] = lo + 0.1 * wid;
      if (mh && U[k] > hi - 0.1 * wid) U[k] = hi - 0.1 * wid;
    }
    rollout(&q, U, X);
  }
  for (int g = 0; g < 2; g++)
    for (long long k = 0; k < G[g].n; k++) {
      const double lo = G[g].lo[k], hi = G[g].hi[k], z = G[g].z[k];
      const int ml = isfinite(lo), mh = isfinite(hi);
      const double thr = fmax(1e-2 * ((ml && mh) ? hi - lo : 1.0), 1e-4);
      G[g].tl[k] = ml ? fmax(z - lo, thr) : 1.0;
      G[g].tu[k] = mh ? fmax(hi - z, thr) : 1.0;
      G[g].ll[k] = ml ? 1.0 / G[g].tl[k] : 0.0;
      G[g].lu[k] = mh ? 1.0 / G[g].tu[k] : 0.0;
    }
  double nu = 1.0, mu_peak = 1.0;
  const double tol = 1e-12;
  status = 2;
  for (it = 1; it <= 80; it++) {
    double mu = 0.0, res = 0.0;
    for (int g = 0; g < 2; g++) {
      Grp *p = &G[g];
      double s = 0.0, rm = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : s) reduction(max : rm)
      for (long long k = 0; k < p->n; k++) {
        const int ml = isfinite(p->lo[k]), mh = isfinite(p->hi[k]);
        s += p->w[k] * ((ml ? p->tl[k] * p->ll[k] : 0.0) + (mh ? p->tu[k] * p->lu[k] : 0.0));
        p->rl[k] = ml ? p->z[k] - p->lo[k] - p->tl[k] : 0.0;
        p->ru[k] = mh ? p->hi[k] - p->z[k] - p->tu[k] : 0.0;
        rm = fmax(rm, fmax(fabs(p->rl[k]), fabs(p->ru[k])));
      }
      mu += s;
      res = fmax(res, rm);
    }
    mu /= m_cnt;
    mu_peak = fmax(mu_peak, mu);
    if (mu <= tol * mu_peak && res <= 1e-10 && nu <= 1e-8) { status = 0; break; }
    if (has_xb)
      for (long long k = 0; k < nX; k++)
        Dx[k] = (isfinite(G[0].lo[k]) ? G[0].ll[k] / G[0].tl[k] : 0.0) + (isfinite(G[0].hi[k]) ? G[0].lu[k] / G[0].tu[k] : 0.0);
    if (has_ub) {
      for (long long k = 0; k < nU; k++) {
        const double d = (isfinite(G[1].lo[k]) ? G[1].ll[k] / G[1].tl[k] : 0.0) + (isfinite(G[1].hi[k]) ? G[1].lu[k] / G[1].tu[k] : 0.0);
        Du[k] = ((k / u) % N) < Nc ? 0.0 : d;
        if (((k / u) % N) < Nc && k < (long long)N * u) Dc[k] = d; /* particle 0: k = j u + r */
      }
    }
    if (lq_factor(&q, has_xb ? Dx : NULL, has_ub ? Du : NULL, (nc && has_ub) ? Dc : NULL)) { status = 1; break; }
    gradient(&q, X, U, gx, gu);
    double alpha = 1.0, sig_mu = 0.0;
    for (int pass = 0; pass < 2; pass++) {
      for (int g = 0; g < 2; g++) {
        Grp *p = &G[g];
#pragma omp parallel for schedule(static)
        for (long long k = 0; k < p->n; k++) {
          const int ml = isfinite(p->lo[k]), mh = isfinite(p->hi[k]);
          const double cl = pass ? p->cl[k] : 0.0, cu = pass ? p->cu[k] : 0.0;
          p->wl[k] = ml ? (sig_mu - cl - p->ll[k] * p->rl[k]) / p->tl[k] : 0.0;
          p->wu[k] = mh ? (sig_mu - cu - p->lu[k] * p->ru[k]) / p->tu[k] : 0.0;
        }
      }
      for (long long k = 0; k < nX; k++) gxx[k] = gx[k] + (has_xb ? -G[0].wl[k] + G[0].wu[k] : 0.0);
      if (nc) memset(gce, 0, sizeof(double) * nc);
      for (long long k = 0; k < nU; k++) {
        const int consj = ((k / u) % N) < Nc;
        const double sh = has_ub ? -G[1].wl[k] + G[1].wu[k] : 0.0;
        guu[k] = gu[k] + (consj ? 0.0 : sh);
        if (consj && k < (long long)N * u) gce[k] = sh;
      }
      lq_solve(&q, gxx, guu, nc ? gce : NULL, dX, dU, NULL);
      double amax = 1.0;
      for (int g = 0; g < 2; g++) {
        Grp *p = &G[g];
        double am = 1.0;
#pragma omp parallel for schedule(static) reduction(min : am)
        for (long long k = 0; k < p->n; k++) {
          const int ml = isfinite(p->lo[k]), mh = isfinite(p->hi[k]);
          const double dz = p->dz[k];
          p->dtl[k] = dz + p->rl[k];
          p->dtu[k] = -dz + p->ru[k];
          p->dll[k] = ml ? p->wl[k] - p->ll[k] - (p->ll[k] / p->tl[k]) * dz : 0.0;
          p->dlu[k] = mh ? p->wu[k] - p->lu[k] + (p->lu[k] / p->tu[k]) * dz : 0.0;
          if (ml && p->dtl[k] < 0.0) am = fmin(am, -p->tl[k] / p->dtl[k]);
          if (mh && p->dtu[k] < 0.0) am = fmin(am, -p->tu[k] / p->dtu[k]);
          if (ml && p->dll[k] < 0.0) am = fmin(am, -p->ll[k] / p->dll[k]);
          if (mh && p->dlu[k] < 0.0) am = fmin(am, -p->lu[k] / p->dlu[k]);
        }
        amax = fmin(amax, am);
      }
      if (pass == 0) {
        double s = 0.0;
        for (int g = 0; g < 2; g++) {
          Grp *p = &G[g];
          double sg = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : sg)
          for (long long k = 0; k < p->n; k++) {
            const int ml = isfinite(p->lo[k]), mh = isfinite(p->hi[k]);
            sg += p->w[k] * ((ml ? (p->tl[k] + amax * p->dtl[k]) * (p->ll[k] + amax * p->dll[k]) : 0.0) +
                             (mh ? (p->tu[k] + amax * p->dtu[k]) * (p->lu[k] + amax * p->dlu[k]) : 0.0));
            p->cl[k] = p->dtl[k] * p->dll[k];
            p->cu[k] = p->dtu[k] * p->dlu[k];
          }
          s += sg;
        }
        const double ratio = (s / m_cnt) / mu;
        sig_mu = ratio * ratio * ratio * mu;
      } else {
        alpha = amax < 1.0 ? fmin(1.0, fmax(0.99, 1.0 - mu) * amax) : 1.0;
      }
    }
    for (long long k = 0; k < nX; k++) X[k] += alpha * dX[k];
    for (long long k = 0; k < nU; k++) U[k] += alpha * dU[k];
    for (int g = 0; g < 2; g++) {
      Grp *p = &G[g];
#pragma omp parallel for schedule(static)
      for (long long k = 0; k < p->n; k++) {
        p->tl[k] += alpha * p->dtl[k];
        p->tu[k] += alpha * p->dtu[k];
        p->ll[k] += alpha * p->dll[k];
        p->lu[k] += alpha * p->dlu[k];
      }
    }
    nu *= (1.0 - alpha);
  }
  /* ---- polish on the control boxes: the interior-point iterate sits ~sqrt(mu) from the vertex at weakly active boxes
   * (1e-6 relative on the quadrotor at mu = 1e-12), too coarse for a 1e-7 comparison.  With the active set it names (multiplier
   * above slack) ONE structured Newton step from a dynamics-consistent point whose held controls sit exactly on their bounds
   * (penalty 1e30 on them) is the exact optimum on that set; it is kept only if it is primal feasible and the held controls'
   * multipliers have the right sign — otherwise the interior-point answer stands. ---- */
  if (status == 0 && has_ub && !has_xb) {
    const double big = 1e30;
    double *Xs = (double *)malloc(sizeof(double) * nX), *Us = (double *)malloc(sizeof(double) * nU);
    int *act = (int *)malloc(sizeof(int) * nU);
    memcpy(Xs, X, sizeof(double) * nX);
    memcpy(Us, U, sizeof(double) * nU);
    for (long long k = 0; k < nU; k++) { /* first guess: multiplier above slack */
      const int consj = ((k / u) % N) < Nc;
      const long long ks = consj ? (k % ((long long)N * u)) : k; /* consensus controls: particle 0 decides for everyone */
      const int lo_a = isfinite(G[1].lo[ks]) && G[1].ll[ks] > G[1].tl[ks], hi_a = isfinite(G[1].hi[ks]) && G[1].lu[ks] > G[1].tu[ks];
      act[k] = lo_a ? 1 : (hi_a ? 2 : 0);
      U[k] = U[consj ? ks : k];
    }
    int ok = 0;
    for (int pass = 0; pass < 12 && !ok; pass++) { /* primal-dual active-set iteration: exact solve on the set, KKT sign check */
      for (long long k = 0; k < nU; k++) {
        const int consj = ((k / u) % N) < Nc;
        const long long ks = consj ? (k % ((long long)N * u)) : k;
        U[k] = act[k] ? (act[k] == 1 ? G[1].lo[ks] : G[1].hi[ks]) : fmin(fmax(U[k], G[1].lo[ks]), G[1].hi[ks]);
        Du[k] = consj ? 0.0 : (act[k] ? big : 0.0);
        if (consj && k < (long long)N * u) Dc[k] = act[k] ? big : 0.0;
      }
      rollout(&q, U, X);
      if (lq_factor(&q, NULL, Du, nc ? Dc : NULL)) break;
      gradient(&q, X, U, gx, gu);
      if (nc) memset(gce, 0, sizeof(double) * nc);
      lq_solve(&q, gx, gu, nc ? gce : NULL, dX, dU, NULL);
      long long changes = 0;
      int nan_seen = 0;
      for (long long k = 0; k < nU; k++) {
        const int consj = ((k / u) % N) < Nc;
        const long long ks = consj ? (k % ((long long)N * u)) : k;
        if (!(dU[k] == dU[k])) { nan_seen = 1; break; }
        if (act[k]) {
          const double lam = act[k] == 1 ? -big * dU[k] : big * dU[k];
          if (lam < -1e-11) { act[k] = 0; changes++; } /* a held control wants to leave its bound: release */
        } else {
          const double z = U[k] + dU[k];
          if (z < G[1].lo[ks] - 1e-13 * fmax(1.0, fabs(G[1].lo[ks]))) { act[k] = 1; changes++; }
          else if (z > G[1].hi[ks] + 1e-13 * fmax(1.0, fabs(G[1].hi[ks]))) { act[k] = 2; changes++; }
        }
      }
      if (nan_seen) break;
      if (changes == 0) ok = 1;
      else
        for (long long k = 0; k < nU; k++) /* next base point: this pass's step, clipped (held / newly held ones are snapped above) */
          if (!act[k]) U[k] += dU[k];
    }
    if (ok) {
      for (long long k = 0; k < nX; k++) X[k] += dX[k];
      for (long long k = 0; k < nU; k++)
        if (!act[k]) U[k] += dU[k];
    } else { /* did not settle: keep the interior-point solution */
      memcpy(X, Xs, sizeof(double) * nX);
      memcpy(U, Us, sizeof(double) * nU);
    }
    if (iters_out) it = ok ? it + 1000 : it; /* (1000 + iterations: polished) */
    free(Xs); free(Us); free(act);
  }
done:
  if (iters_out) *iters_out = it;
  grp_free(&G[0]);
  grp_free(&G[1]);
  free(q.K); free(q.Hinv); free(q.kff); free(q.Phi); free(q.Hc);
  free(gx); free(gu); free(gxx); free(guu); free(dX); free(dU); free(Dx); free(Du); free(Dc); free(gce);
  return status;
}
