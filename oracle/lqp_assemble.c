/*
 * ORACLE (test infrastructure, not product code).
 *
 * CPU restatement of the joint (all-particle, consensus) QP that the reference
 * assembles in PMPC.jl/src/lqp_utils.jl and hands to OSQP:
 *
 *     min_z  1/2 z' P z + q' z    s.t.   A z = b,   l <= G z <= u
 *     z = [ U_cons (Nc*udim) ; U_free particle-major (M*Nf*udim) ; X particle-major (M*N*xdim) ]
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this file.  PARITY PINNED by reference-held OUTPUT (the reference's own tests hold no numeric
 * golden vector for this path — PMPC.jl/test/runtests.jl:33-41 asserts !isnan only — and
 * Julia/OSQP cannot run in the build container): seven (obj, resid) tables that the
 * reference's Julia + ECOS stack printed into its committed notebooks, 4 significant digits,
 * transcribed by tests/golden/make_golden.py into tests/golden/ref_*.npz and reproduced
 * through this assembly by tests/test_host_logic.py (CPU) and tests/test_golden_gpu.py (GPU):
 * hard control boxes (examples/gpu_solver.ipynb), slew + log-barrier smoothing
 * (tests/root_testing.ipynb cells 3-4), consensus with M = 20 / Nc = 5 and the
 * eps-anchored particle weights (tests/root_testing.ipynb cells 10-11), and four more
 * (DESIGN.md section 7).  Still UNPINNED by any reference output — nothing in the
 * reference exercises them with recorded numbers: state boxes, slew_reg0 / slew_um1,
 * M > 500, k < M; those are held by the KKT certificate (lqp_oracle.py), two independent
 * derivations agreeing to 1e-15 and the reference's Python SCP loop run over this oracle.
 *
 * Every function cites the reference lines it follows.  Indices are 0-based here
 * (the reference is 1-based Julia); array layouts are the C-ABI layouts of
 * PMPC.jl/src/c_interface.jl:28-46, i.e. column-major
 *     x0 (xdim,M)  f,X_prev,X_ref,lx,ux (xdim,N,M)  U_prev,U_ref,lu,uu (udim,N,M)
 *     fx,Q (xdim,xdim,N,M)  fu (xdim,udim,N,M)  R (udim,udim,N,M)
 *     slew_reg (M)  slew_reg0 (M)  slew_um1 (udim,M)
 * reg_x / reg_u are the scalars of the ABI (broadcast per particle by
 * PMPC.jl/src/main.jl:19-21).
 */
#include <stddef.h>
#include <string.h>

typedef long long i64;

/* column-major accessors matching unsafe_wrap dims (c_interface.jl:28-46) */
#define V3(p, d0, r, j, i) ((p)[(size_t)(r) + (size_t)(d0) * ((size_t)(j) + (size_t)N * (size_t)(i))])
#define M4(p, d0, d1, r, t, j, i) \
  ((p)[(size_t)(r) + (size_t)(d0) * ((size_t)(t) + (size_t)(d1) * ((size_t)(j) + (size_t)N * (size_t)(i)))])

/* sizes of the joint QP: lqp_utils.jl:4-15 (P), :223-226 (A), :311-319 (G) */
void lqp_sizes(size_t xdim, size_t udim, size_t N, size_t M, i64 Nc_in, int has_ub, int has_xb, i64 *n,
               i64 *m_eq, i64 *m_in, i64 *nnzP_max, i64 *nnzA) {
  i64 Nc = Nc_in >= 0 ? Nc_in : (i64)N; /* lqp_utils.jl:4 */
  i64 Nf = (i64)N - Nc;
  *n = Nc * (i64)udim + (i64)M * (Nf * (i64)udim + (i64)N * (i64)xdim);
  *m_eq = (i64)M * (i64)N * (i64)xdim;
  *m_in = 0;
  if (has_ub) *m_in += (i64)udim * Nc + (i64)M * Nf * (i64)udim;
  if (has_xb) *m_in += (i64)M * (i64)N * (i64)xdim;
  /* lqp_utils.jl:6 counts 3*udim^2 per control column; the consensus->free slew
     coupling (:47-57) can emit M entries in one column, so bound it generously */
  *nnzP_max = (i64)3 * Nc * (i64)udim * (i64)udim + (i64)M * ((i64)3 * Nf * (i64)udim * (i64)udim +
              (i64)N * (i64)xdim * (i64)xdim) + (i64)M * (i64)udim + (i64)2 * (*n);
  *nnzA = (i64)xdim * (i64)udim * (i64)M * (i64)N + (i64)M * (i64)N * (i64)xdim +
          (i64)M * ((i64)N - 1) * (i64)xdim * (i64)xdim; /* lqp_utils.jl:225 */
}

/* slew diagonal contribution of control step j (0-based): lqp_utils.jl:31-39, :81-88 */
static double slew_diag(double s0, double s, size_t j, size_t N) {
  if (j == 0) return s0 + s;
  if (j == N - 1) return s;
  return 2.0 * s;
}

/* lqp_repr_Pq: PMPC.jl/src/lqp_utils.jl:2-216 (Hf/hf branches are unreachable from
 * lqp_solve, main.jl:134, and are not restated).  Returns nnz(P).  P is emitted
 * exactly as the reference does: full (both triangles), column by column.
 * wts (NULL = none): per-particle cost weights applied as scale_probs_cost! does to each
 * OCProb before assembly (main.jl:96-112: Q, R, reg_x, reg_u, slew_reg0, slew_reg times w_i;
 * neither the normalisation by sum(w) nor the extra scaling of slew_um1 is applied here —
 * the caller passes the multipliers it means). */
#define WT(i) (wts ? wts[i] : 1.0)
i64 lqp_repr_Pq(size_t xdim, size_t udim, size_t N, size_t M, i64 Nc_in, const double *X_prev,
                const double *U_prev, const double *Q, const double *R, const double *X_ref,
                const double *U_ref, double reg_x, double reg_u, const double *slew_reg,
                const double *slew_reg0, const double *slew_um1, i64 *Pp, i64 *Pi, double *Px,
                double *q, const double *wts) {
  size_t Nc = Nc_in >= 0 ? (size_t)Nc_in : N;
  size_t Nf = N - Nc;
  size_t n = Nc * udim + M * (Nf * udim + N * xdim);
  i64 k = 0;
  size_t c = 0;
  Pp[0] = 0;
  double slew_sum = 0.0; /* mapreduce(i -> probs[i].slew_reg, +, 1:M), :20 */
  for (size_t i = 0; i < M; i++) slew_sum += WT(i) * slew_reg[i];

  for (size_t j = 0; j < Nc; j++) { /* U consensus, :17-61 */
    for (size_t t = 0; t < udim; t++) {
      i64 k_old = k;
      if (j > 0 && slew_sum != 0.0) { /* slew top half, :21-25 */
        Pi[k] = (i64)(udim * (j - 1) + t);
        Px[k] = -slew_sum;
        k++;
      }
      for (size_t r = 0; r < udim; r++) { /* core cost, :26-46 */
        double val = 0.0;
        for (size_t i = 0; i < M; i++) {
          val += WT(i) * M4(R, udim, udim, r, t, j, i);
          if (r == t) val += WT(i) * (reg_u + slew_diag(slew_reg0[i], slew_reg[i], j, N));
        }
        if (val != 0.0) {
          Pi[k] = (i64)(udim * j + r);
          Px[k] = val;
          k++;
        }
      }
      if (j + 1 < Nc && slew_sum != 0.0) { /* slew bottom half, :47-50 */
        Pi[k] = (i64)(udim * (j + 1) + t);
        Px[k] = -slew_sum;
        k++;
      } else if (j + 1 < N && slew_sum != 0.0) { /* last consensus -> first free of each particle, :51-57 */
        for (size_t i = 0; i < M; i++) {
          Pi[k] = (i64)(Nc * udim + udim * Nf * i + t);
          Px[k] = -WT(i) * slew_reg[i];
          k++;
        }
      }
      c++;
      Pp[c] = Pp[c - 1] + (k - k_old);
    }
  }
  for (size_t i = 0; i < M; i++) { /* U free, :63-102 */
    for (size_t j = Nc; j < N; j++) {
      for (size_t t = 0; t < udim; t++) {
        i64 k_old = k;
        double s = WT(i) * slew_reg[i];
        if (j > 0 && s != 0.0) { /* slew top half, :67-75 */
          /* previous control is free iff (1-based) j-1 > Nc, i.e. 0-based j > Nc */
          size_t idx = (j > Nc) ? Nc * udim + Nf * udim * i + udim * (j - Nc - 1) + t : udim * (j - 1) + t;
          Pi[k] = (i64)idx;
          Px[k] = -s;
          k++;
        }
        for (size_t r = 0; r < udim; r++) { /* core cost, :76-93 */
          double val = WT(i) * M4(R, udim, udim, r, t, j, i);
          if (r == t) val += WT(i) * reg_u + slew_diag(WT(i) * slew_reg0[i], s, j, N);
          if (val != 0.0) {
            Pi[k] = (i64)(Nc * udim + Nf * udim * i + udim * (j - Nc) + r);
            Px[k] = val;
            k++;
          }
        }
        if (j + 1 < N && s != 0.0) { /* slew bottom half, :94-98 */
          Pi[k] = (i64)(Nc * udim + Nf * udim * i + udim * (j - Nc + 1) + t);
          Px[k] = -s;
          k++;
        }
        c++;
        Pp[c] = Pp[c - 1] + (k - k_old);
      }
    }
  }
  size_t offset = Nc * udim + M * Nf * udim; /* :109 */
  for (size_t i = 0; i < M; i++) {           /* X, :110-160 */
    for (size_t j = 0; j < N; j++) {
      for (size_t t = 0; t < xdim; t++) {
        i64 k_old = k;
        for (size_t r = 0; r < xdim; r++) {
          double val = WT(i) * (M4(Q, xdim, xdim, r, t, j, i) + (r == t ? reg_x : 0.0)); /* :130-131 */
          if (val != 0.0) {
            Pi[k] = (i64)(offset + N * xdim * i + xdim * j + r);
            Px[k] = val;
            k++;
          }
        }
        c++;
        Pp[c] = Pp[c - 1] + (k - k_old);
      }
    }
  }

  memset(q, 0, n * sizeof(double)); /* :161 */
  /* q[1:udim] .+= sum_i -slew_reg0_i * slew_um1_i, :165.  Applied to the first udim
     entries of z whatever Nc is; with Nc == 0 those entries are then overwritten
     by the '=' assignment of the free-control loop (:190), as in the reference. */
  if (n >= udim) {
    for (size_t r = 0; r < udim; r++) {
      double acc = 0.0;
      for (size_t i = 0; i < M; i++) acc += -WT(i) * slew_reg0[i] * slew_um1[r + udim * i];
      q[r] += acc;
    }
  }
  for (size_t j = 0; j < Nc; j++) { /* U cons, :166-178 */
    size_t sidx = udim * j;
    for (size_t r = 0; r < udim; r++) {
      double val = 0.0;
      for (size_t i = 0; i < M; i++) {
        val -= WT(i) * reg_u * V3(U_prev, udim, r, j, i);
        for (size_t t = 0; t < udim; t++) val -= WT(i) * M4(R, udim, udim, r, t, j, i) * V3(U_ref, udim, t, j, i);
      }
      q[sidx + r] += val;
    }
  }
  for (size_t i = 0; i < M; i++) { /* U free, :179-191 */
    for (size_t j = Nc; j < N; j++) {
      size_t sidx = Nc * udim + Nf * udim * i + udim * (j - Nc);
      for (size_t r = 0; r < udim; r++) {
        double val = -reg_u * V3(U_prev, udim, r, j, i);
        for (size_t t = 0; t < udim; t++) val -= M4(R, udim, udim, r, t, j, i) * V3(U_ref, udim, t, j, i);
        q[sidx + r] = WT(i) * val;
      }
    }
  }
  for (size_t i = 0; i < M; i++) { /* X, :197-213 */
    for (size_t j = 0; j < N; j++) {
      size_t sidx = udim * (M * Nf + Nc) + N * xdim * i + xdim * j;
      for (size_t r = 0; r < xdim; r++) {
        double val = -reg_x * V3(X_prev, xdim, r, j, i);
        for (size_t t = 0; t < xdim; t++) val -= M4(Q, xdim, xdim, r, t, j, i) * V3(X_ref, xdim, t, j, i);
        q[sidx + r] = WT(i) * val;
      }
    }
  }
  return k;
}

/* lqp_repr_Ab: PMPC.jl/src/lqp_utils.jl:219-303.  Row block (i,j):
 *   fu_j u_j - x_j + fx_j x_{j-1} = -f_j + fu_j U_prev_j + fx_j X_prev_{j-1}   (fx term absent for j = 0) */
i64 lqp_repr_Ab(size_t xdim, size_t udim, size_t N, size_t M, i64 Nc_in, const double *f, const double *fx,
                const double *fu, const double *X_prev, const double *U_prev, i64 *Ap, i64 *Ai, double *Ax,
                double *b) {
  size_t Nc = Nc_in >= 0 ? (size_t)Nc_in : N;
  i64 k = 0;
  size_t c = 0;
  Ap[0] = 0;
  for (size_t j = 0; j < Nc; j++) { /* U consensus columns, :233-246 */
    for (size_t t = 0; t < udim; t++) {
      i64 k_old = k;
      for (size_t i = 0; i < M; i++)
        for (size_t r = 0; r < xdim; r++) {
          Ai[k] = (i64)(N * xdim * i + xdim * j + r);
          Ax[k] = M4(fu, xdim, udim, r, t, j, i);
          k++;
        }
      c++;
      Ap[c] = Ap[c - 1] + (k - k_old);
    }
  }
  for (size_t i = 0; i < M; i++) /* U free columns, :247-260 */
    for (size_t j = Nc; j < N; j++)
      for (size_t t = 0; t < udim; t++) {
        i64 k_old = k;
        for (size_t r = 0; r < xdim; r++) {
          Ai[k] = (i64)(N * xdim * i + xdim * j + r);
          Ax[k] = M4(fu, xdim, udim, r, t, j, i);
          k++;
        }
        c++;
        Ap[c] = Ap[c - 1] + (k - k_old);
      }
  for (size_t i = 0; i < M; i++) /* X columns, :261-279 */
    for (size_t j = 0; j < N; j++)
      for (size_t t = 0; t < xdim; t++) {
        i64 k_old = k;
        Ai[k] = (i64)(N * xdim * i + xdim * j + t);
        Ax[k] = -1.0;
        k++;
        if (j + 1 != N) {
          for (size_t r = 0; r < xdim; r++) {
            Ai[k] = (i64)(N * xdim * i + xdim * j + xdim + r);
            Ax[k] = M4(fx, xdim, xdim, r, t, j + 1, i);
            k++;
          }
        }
        c++;
        Ap[c] = Ap[c - 1] + (k - k_old);
      }
  for (size_t i = 0; i < M; i++) /* b, :280-300; x0 never enters (:293-296 commented out upstream) */
    for (size_t j = 0; j < N; j++) {
      size_t sidx = N * xdim * i + xdim * j;
      for (size_t r = 0; r < xdim; r++) {
        double val = -V3(f, xdim, r, j, i);
        for (size_t t = 0; t < udim; t++) val += M4(fu, xdim, udim, r, t, j, i) * V3(U_prev, udim, t, j, i);
        if (j != 0)
          for (size_t t = 0; t < xdim; t++) val += M4(fx, xdim, xdim, r, t, j, i) * V3(X_prev, xdim, t, j - 1, i);
        b[sidx + r] = val;
      }
    }
  return k;
}

/* lqp_repr_Gla: PMPC.jl/src/lqp_utils.jl:306-393.  G has one 1.0 per bounded variable;
 * consensus-control bounds are read from particle 0 (:329-330). */
i64 lqp_repr_Gla(size_t xdim, size_t udim, size_t N, size_t M, i64 Nc_in, int has_ub, int has_xb,
                 const double *lx, const double *ux, const double *lu, const double *uu, i64 *Gp, i64 *Gi,
                 double *Gx, double *l, double *u) {
  size_t Nc = Nc_in >= 0 ? (size_t)Nc_in : N;
  size_t Nf = N - Nc;
  i64 k = 0;
  size_t c = 0;
  Gp[0] = 0;
  if (has_ub) {
    for (size_t j = 0; j < Nc; j++)
      for (size_t r = 0; r < udim; r++) {
        Gi[k] = (i64)(udim * j + r);
        Gx[k] = 1.0;
        l[k] = lu[udim * j + r]; /* probs[1].lu linear index, :329 */
        u[k] = uu[udim * j + r];
        k++;
        c++;
        Gp[c] = Gp[c - 1] + 1;
      }
    for (size_t i = 0; i < M; i++)
      for (size_t j = Nc; j < N; j++)
        for (size_t r = 0; r < udim; r++) {
          Gi[k] = (i64)(Nc * udim + Nf * udim * i + udim * (j - Nc) + r);
          Gx[k] = 1.0;
          l[k] = V3(lu, udim, r, j, i);
          u[k] = V3(uu, udim, r, j, i);
          k++;
          c++;
          Gp[c] = Gp[c - 1] + 1;
        }
  } else {
    for (size_t cc = 0; cc < Nc * udim + M * Nf * udim; cc++) {
      c++;
      Gp[c] = Gp[c - 1];
    }
  }
  if (has_xb) {
    size_t roff = has_ub ? udim * (Nc + M * Nf) : 0; /* :368 */
    for (size_t i = 0; i < M; i++)
      for (size_t j = 0; j < N; j++)
        for (size_t r = 0; r < xdim; r++) {
          Gi[k] = (i64)(N * xdim * i + xdim * j + r + roff);
          Gx[k] = 1.0;
          l[k] = V3(lx, xdim, r, j, i);
          u[k] = V3(ux, xdim, r, j, i);
          k++;
          c++;
          Gp[c] = Gp[c - 1] + 1;
        }
  } else {
    for (size_t cc = 0; cc < M * N * xdim; cc++) {
      c++;
      Gp[c] = Gp[c - 1];
    }
  }
  return k;
}

/* split_lqp_vars: PMPC.jl/src/lqp_utils.jl:395-423.  Outputs X (xdim,N,M), U (udim,N,M). */
void split_lqp_vars(size_t xdim, size_t udim, size_t N, size_t M, i64 Nc_in, const double *z, double *X,
                    double *U) {
  size_t Nc = Nc_in >= 0 ? (size_t)Nc_in : N;
  size_t Nf = N - Nc;
  for (size_t i = 0; i < M; i++)
    for (size_t j = 0; j < Nc; j++)
      for (size_t r = 0; r < udim; r++) V3(U, udim, r, j, i) = z[udim * j + r];
  for (size_t i = 0; i < M; i++)
    for (size_t j = Nc; j < N; j++)
      for (size_t r = 0; r < udim; r++) V3(U, udim, r, j, i) = z[Nc * udim + Nf * udim * i + udim * (j - Nc) + r];
  size_t offset = Nc * udim + M * Nf * udim;
  for (size_t i = 0; i < M; i++)
    for (size_t j = 0; j < N; j++)
      for (size_t r = 0; r < xdim; r++) V3(X, xdim, r, j, i) = z[offset + N * xdim * i + xdim * j + r];
}

/* rollout!: PMPC.jl/src/types.jl:161-173 (linearised dynamics propagation used by `coerce`
 * and by the tests' dynamics-violation checks). */
void lin_rollout(size_t xdim, size_t udim, size_t N, size_t M, const double *f, const double *fx,
                 const double *fu, const double *X_prev, const double *U_prev, const double *U, double *X) {
  for (size_t i = 0; i < M; i++)
    for (size_t j = 0; j < N; j++)
      for (size_t r = 0; r < xdim; r++) {
        double val = V3(f, xdim, r, j, i);
        for (size_t t = 0; t < udim; t++)
          val += M4(fu, xdim, udim, r, t, j, i) * (V3(U, udim, t, j, i) - V3(U_prev, udim, t, j, i));
        if (j > 0)
          for (size_t t = 0; t < xdim; t++)
            val += M4(fx, xdim, xdim, r, t, j, i) * (V3(X, xdim, t, j - 1, i) - V3(X_prev, xdim, t, j - 1, i));
        V3(X, xdim, r, j, i) = val;
      }
}
