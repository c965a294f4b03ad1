"""ORACLE (test infrastructure, not product code) for the reference's DEFAULT solver path, `c_lcone_solve`.

A DIRECT restatement — not a derivation — of how the reference builds and solves its cone program
(PMPC.jl/src/main.jl:194-354 `lcone_solve`): the same matrices, row for row, handed to a generic conic solver.

  qp_repr_Pq_py            PMPC.jl/src/qp_utils.jl:60-162   per-particle (P, q, resid) over [U; X]
  Pqr2Gh_py                PMPC.jl/src/cone_utils.jl:25-61  1/2 z'Pz + q'z + r <= tau  as a second-order cone
  lcone_repr_Pq_py         PMPC.jl/src/cone_utils.jl:64-95  the M epigraph cones over [U_cons; U_free; X; y; t]
  augment_cone_problem_py  PMPC.jl/src/cone_utils.jl:99-151 appending (l, q, e) rows and new variables
  make_logbarrier_constraint_py / smoothen_linear_inequalities_py
                           PMPC.jl/src/cone_utils.jl:173-232
  lcone_problem_py         PMPC.jl/src/main.jl:204-316      objective with COST_ANCHOR_EPS, y >= 0, boxes hard / logbarrier /
                                                            squareplus, `extra_cstrs`
  conic_solve_py           the numeric solver: ECOS 2.0.8 through JuMP (cone_solver.jl:121-191) is a third-party dependency that
                           is not in /root/reference; what is restated is the PROBLEM it is given —
                           min c'xi  s.t.  A xi = b,  h - G xi >= 0 (first l rows),  G xi - h in SOC(q_k),  G xi - h in K_exp
                           (cone_solver.jl:163-188) — solved here by primal log-barrier path following on the sparse KKT system.

Exponential cone convention.  make_logbarrier_constraint (cone_utils.jl:173-203) orders its three rows for the solver named in
`solver`; for "ecos" its own comment states the cone as  exp(x/z) <= y/z  with (x, y, z) = (-alpha t, -alpha (g'u - h), 1), i.e.
t >= -(1/alpha) log(alpha (h - g'u)): the log barrier.  JuMP_solve then imposes MOI.ExponentialCone on those rows, whose textbook
definition  y exp(x/y) <= z  would read the same rows as t >= s log(alpha s).  Which of the two the authors' runs computed is
decided by their printed output: tests/golden/ref_root_testing_consensus.npz (23 rows the reference's Julia + ECOS stack printed)
fits the LOG BARRIER to its 4 digits and not the other reading (tests/test_host_logic.py).  `exp_convention="ecos"` (default) is
that reading; "moi" is kept for the record.

PINS: the (obj, resid) tables of tests/golden/ref_*.npz, through `lcone_direct_py` at M = 1 and M = 20 (tests/test_oracle_golden.py).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from . import lqp_oracle as orc

COST_ANCHOR_EPS = 1e-3  # main.jl:223


# -------------------------------------------------------------------------------------------------
# per-particle quadratic (qp_utils.jl:60-162), variables [U (N u); X (N x)]
# -------------------------------------------------------------------------------------------------
def qp_repr_Pq_py(Q, R, X_prev, U_prev, X_ref, U_ref, reg_x, reg_u, slew_reg=0.0, slew_reg0=0.0, slew_um1=None):
    """One particle: Q (N,x,x), R (N,u,u) with [j, r, t] = entry (r, t) of stage j; vectors (N,d).  Returns dense P, q, resid."""
    N, xdim = X_prev.shape
    udim = U_prev.shape[1]
    n = N * (xdim + udim)
    P = np.zeros((n, n))
    for j in range(N):  # qp_utils.jl:67-103: control blocks, column t of stage j
        for t in range(udim):
            col = udim * j + t
            if j > 0 and slew_reg != 0.0:  # :71-76 slew top half
                P[udim * (j - 1) + t, col] = -slew_reg
            for r in range(udim):  # :77-95
                val = R[j, r, t]
                if r == t:
                    val += reg_u
                    if j == 0:
                        val += slew_reg0 + slew_reg
                    elif j == N - 1:
                        val += slew_reg
                    else:
                        val += 2 * slew_reg
                P[udim * j + r, col] = val
            if j < N - 1 and slew_reg != 0.0:  # :96-100 slew bottom half
                P[udim * (j + 1) + t, col] = -slew_reg
    for j in range(N):  # :104-117 state blocks
        for t in range(xdim):
            for r in range(xdim):
                P[N * udim + xdim * j + r, N * udim + xdim * j + t] = Q[j, r, t] + (reg_x if r == t else 0.0)
    q = np.zeros(n)  # :118-139
    if slew_um1 is not None:
        q[:udim] += -slew_reg0 * np.asarray(slew_um1, dtype=np.float64)
    for j in range(N):
        q[udim * j:udim * (j + 1)] += -reg_u * U_prev[j] - R[j] @ U_ref[j]
    for j in range(N):
        q[N * udim + xdim * j:N * udim + xdim * (j + 1)] = -reg_x * X_prev[j] - Q[j] @ X_ref[j]
    resid = 0.0  # :140-160 (no slew part upstream)
    for j in range(N):
        resid += 0.5 * reg_u * float(U_prev[j] @ U_prev[j]) + 0.5 * float((R[j] @ U_ref[j]) @ U_ref[j])
    for j in range(N):
        resid += 0.5 * reg_x * float(X_prev[j] @ X_prev[j]) + 0.5 * float((Q[j] @ X_ref[j]) @ X_ref[j])
    return P, q, resid


def Pqr2Gh_py(P, q, r=0.0):
    """cone_utils.jl:25-61:  || (tau - bet ; L z - b) ||_2 <= tau - alf  <=>  1/2 z'Pz + q'z + r <= tau, with L'L = P/2.
    Returns G_left ((2 + n) x n), G_right ((2 + n) x 1) and h in the convention  G [z; tau] - h  in SOC."""
    n = P.shape[0]
    C = np.linalg.cholesky(P)            # :36 F = cholesky(P): P = C C'
    L = C.T / np.sqrt(2.0)               # :37-42 L = F.L' / sqrt(2)
    err = np.linalg.norm(L.T @ L - 0.5 * P)
    assert err <= 1e-9 * max(1.0, np.linalg.norm(P)), err  # :45-48 (a warning upstream)
    b = np.linalg.solve(L.T, -q / 2.0)   # :50
    bTb = float(b @ b)
    alf, bet = -(bTb - r) - 0.25, 0.25 - (bTb - r)  # :54
    G_left = np.vstack([np.zeros((2, n)), L])       # :56
    h = np.concatenate([[alf, bet], b])             # :57
    G_right = np.concatenate([[1.0, 1.0], np.zeros(n)])[:, None]  # :58
    return G_left, G_right, h


@dataclass
class ConeProblem:  # cone_utils.jl:4-13
    l: int
    q: List[int]
    e: int
    G: sp.csr_matrix
    A: sp.csr_matrix
    c: np.ndarray
    h: np.ndarray
    b: np.ndarray
    dims: tuple = ()        # (xdim, udim, N, M, Nc) of the trajectory block
    nz: int = 0             # variables of [U_cons; U_free; X]
    notes: list = field(default_factory=list)


def lcone_repr_Pq_py(parts, Nc, xdim, udim, N):
    """cone_utils.jl:64-95.  `parts` = per-particle (P, q, resid).  Columns of G_left: [U_cons | U_free (M blocks) | X (M blocks)];
    G_right: [y (M) | t]."""
    M = len(parts)
    Nf = N - Nc
    Gl, Gr, hs = zip(*(Pqr2Gh_py(*p) for p in parts))  # :72-74
    m = Gl[0].shape[0]
    G_right = sp.hstack([sp.block_diag([sp.csr_matrix(g) for g in Gr]), sp.vstack([sp.csr_matrix(g) for g in Gr])]).tocsr()  # :75-76
    h = np.concatenate(hs)
    rows = []
    for i in range(M):  # :80-90
        G_ucons = Gl[i][:, :Nc * udim]
        G_rest_u = Gl[i][:, Nc * udim:N * udim]
        G_x = Gl[i][:, N * udim:]
        rows.append(sp.hstack([sp.csr_matrix(G_ucons), sp.csr_matrix((m, Nf * udim * i)), sp.csr_matrix(G_rest_u),
                               sp.csr_matrix((m, Nf * udim * (M - 1 - i))), sp.csr_matrix((m, N * xdim * i)), sp.csr_matrix(G_x),
                               sp.csr_matrix((m, N * xdim * (M - 1 - i)))]))
    return sp.vstack(rows).tocsr(), G_right, h


def augment_cone_problem_py(prob: ConeProblem, l, q, e, G_left, G_right, h, c_left, c_right):
    """cone_utils.jl:99-151 (the `extra_cstr` branch): rows are appended inside their class (linear | SOC | exp), columns of G_left
    are the LEADING variables, G_right adds new variables at the end."""
    q = [int(v) for v in q]
    G_left, G_right = sp.csr_matrix(G_left), sp.csr_matrix(G_right)
    h = np.asarray(h, dtype=np.float64).reshape(-1)
    assert l + sum(q) + 3 * e == G_left.shape[0] == G_right.shape[0] == h.size  # :104-107
    assert len(c_left) == G_left.shape[1]
    n_fill, n_new = prob.G.shape[1] - G_left.shape[1], G_right.shape[1]
    pad = lambda Gp: sp.hstack([Gp, sp.csr_matrix((Gp.shape[0], n_new))])
    ext = lambda a, b_: sp.hstack([G_left[a:b_], sp.csr_matrix((b_ - a, n_fill)), G_right[a:b_]])
    pl, pq = prob.l, sum(prob.q)
    G_lin = sp.vstack([pad(prob.G[:pl]), ext(0, l)])                                 # :115-119
    h_lin = np.concatenate([prob.h[:pl], h[:l]])
    G_soc = sp.vstack([pad(prob.G[pl:pl + pq]), ext(l, l + sum(q))])                # :126-130
    h_soc = np.concatenate([prob.h[pl:pl + pq], h[l:l + sum(q)]])
    G_exp = sp.vstack([pad(prob.G[pl + pq:pl + pq + 3 * prob.e]), ext(l + sum(q), l + sum(q) + 3 * e)])  # :132-138
    h_exp = np.concatenate([prob.h[pl + pq:pl + pq + 3 * prob.e], h[l + sum(q):]])
    prob.G = sp.vstack([G_lin, G_soc, G_exp]).tocsr()
    prob.h = np.concatenate([h_lin, h_soc, h_exp])
    prob.l, prob.q, prob.e = prob.l + l, list(prob.q) + q, prob.e + e
    assert prob.l + sum(prob.q) + 3 * prob.e == prob.G.shape[0]
    c = prob.c.copy()                                                                # :147-149
    c[:len(c_left)] += np.asarray(c_left, dtype=np.float64)
    prob.c = np.concatenate([c, np.asarray(c_right, dtype=np.float64).reshape(-1)])
    prob.A = sp.hstack([prob.A, sp.csr_matrix((prob.A.shape[0], n_new))]).tocsr()    # :151


def make_logbarrier_constraint_py(g, hi, alpha, solver="ecos"):
    """cone_utils.jl:173-203: three rows + one new variable per inequality g'u <= hi."""
    n = g.shape[1]
    zero = sp.csr_matrix((1, n))
    if solver.lower() == "ecos":  # :181-188
        G_left = sp.vstack([zero, -alpha * g, zero])
        h = np.array([0.0, -alpha * hi, -1.0])
    else:                         # :190-197
        G_left = sp.vstack([zero, zero, -alpha * g])
        h = np.array([0.0, -1.0, -alpha * hi])
    G_right = sp.csr_matrix(np.array([[-alpha], [0.0], [0.0]]))
    return G_left, G_right, h


def smoothen_linear_inequalities_py(A, b, alpha, beta=1.0, method="logbarrier", solver="ecos"):
    """cone_utils.jl:205-232."""
    A = sp.csr_matrix(A)
    m = A.shape[0]
    Gl, Gr, hs = [], [], []
    for i in range(m):
        a, bi = A[i], float(b[i])
        if method == "logbarrier":
            g_l, g_r, h_ = make_logbarrier_constraint_py(a, bi, alpha, solver=solver)
        else:  # squareplus :223-228
            g_l = sp.vstack([-a, a, sp.csr_matrix((1, A.shape[1]))])
            g_r = sp.csr_matrix(np.array([[2.0 / beta], [0.0], [0.0]]))
            h_ = np.array([-bi, bi, 1.0 / alpha])
        Gl.append(g_l), Gr.append(g_r), hs.append(h_)
    if m == 0:
        return sp.csr_matrix((0, A.shape[1])), sp.csr_matrix((0, 0)), np.zeros(0)
    return sp.vstack(Gl).tocsr(), sp.block_diag(Gr).tocsr(), np.concatenate(hs)


def lcone_problem_py(x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref, *, reg_x, reg_u, Nc=-1, x_l=None, x_u=None, u_l=None, u_u=None,
                     slew_reg=None, slew_reg0=None, slew_um1=None, k=None, smooth_cstr="", smooth_alpha=float("nan"), smooth_beta=1.0,
                     extra_cstrs=(), solver="ecos") -> ConeProblem:
    """main.jl:194-316.  py layout in (see lqp_oracle).  The equality / box rows are lqp_repr_Ab / lqp_repr_Gla (cone_utils.jl:234-239)."""
    f = orc._f64(f)
    M, N, xdim = f.shape
    udim = np.shape(fu)[-1]
    Ncc = N if Nc < 0 else int(Nc)   # main.jl:200-201
    Nf = N - Ncc
    kk = M if (k is None or k < 0) else int(k)  # :204-205
    nanx, nanu = np.full((M, N, xdim), np.nan), np.full((M, N, udim), np.nan)
    bx = lambda z, d: d if z is None or np.size(z) == 0 else np.broadcast_to(orc._f64(z), d.shape).copy()
    sr = np.full(M, np.nan) if slew_reg is None else np.broadcast_to(orc._f64(slew_reg), (M,)).copy()
    sr0 = np.full(M, np.nan) if slew_reg0 is None else np.broadcast_to(orc._f64(slew_reg0), (M,)).copy()
    um1 = np.full((M, udim), np.nan) if slew_um1 is None else np.broadcast_to(orc._f64(slew_um1), (M, udim)).copy()
    qp = orc.assemble_abi(xdim, udim, N, M, Ncc, f, orc.to_abi_mat(fx), orc.to_abi_mat(fu), orc._f64(X_prev), orc._f64(U_prev), orc.to_abi_mat(Q),
                          orc.to_abi_mat(R), orc._f64(X_ref), orc._f64(U_ref), bx(x_l, nanx), bx(x_u, nanx), bx(u_l, nanu), bx(u_u, nanu),
                          float(reg_x), float(reg_u), sr, sr0, um1)
    # sentinels as c_interface.jl:56-70 resolves them
    s_reg = np.where(np.isnan(sr), 0.0, sr) if not np.any(np.isnan(sr)) else np.zeros(M)
    use0 = not (np.any(np.isnan(sr0)) or np.any(np.isnan(um1)))
    s_reg0 = sr0 if use0 else np.zeros(M)
    s_um1 = um1 if use0 else np.zeros((M, udim))
    Q_, R_ = orc._f64(Q), orc._f64(R)
    parts = [qp_repr_Pq_py(Q_[i], R_[i], orc._f64(X_prev)[i], orc._f64(U_prev)[i], orc._f64(X_ref)[i], orc._f64(U_ref)[i], float(reg_x), float(reg_u),
                           float(s_reg[i]), float(s_reg0[i]), s_um1[i]) for i in range(M)]
    Pq_G_left, Pq_G_right, Pq_h = lcone_repr_Pq_py(parts, Ncc, xdim, udim, N)        # main.jl:216
    nz = Ncc * udim + M * (N * xdim + Nf * udim)
    assert Pq_G_left.shape[1] == nz == qp.A.shape[1]
    Pq_G = sp.hstack([Pq_G_left, Pq_G_right]).tocsr()                                # :217
    nr = Pq_G_right.shape[1]
    F = sp.hstack([qp.G, sp.csr_matrix((qp.G.shape[0], nr))]).tocsr()               # :220
    A = sp.hstack([qp.A, sp.csr_matrix((qp.A.shape[0], nr))]).tocsr()               # :221
    c = np.concatenate([np.zeros(nz), (1 + COST_ANCHOR_EPS) * np.ones(M), [(1 - COST_ANCHOR_EPS) * kk]])  # :224-228
    y_nonneg = sp.hstack([sp.csr_matrix((M, nz)), -sp.identity(M), sp.csr_matrix((M, 1))])  # :231
    G = sp.vstack([y_nonneg, Pq_G]).tocsr()                                          # :232
    h = np.concatenate([np.zeros(M), Pq_h])                                         # :233
    prob = ConeProblem(M, [2 + N * (xdim + udim)] * M, 0, G, A, c, h, qp.b.copy(), dims=(xdim, udim, N, M, Ncc), nz=nz)  # :239
    # ---- boxes (main.jl:242-291) -------------------------------------------------------------------------------
    if smooth_alpha != smooth_alpha:     # :243-245
        smooth_cstr = ""
    elif smooth_cstr == "":
        smooth_cstr = "logbarrier"       # :246
    l_, u_ = qp.l, qp.u
    if F.shape[0] > 0:
        fin_l, fin_u = np.isfinite(l_), np.isfinite(u_)
        # (rows with an infinite side are no constraint; upstream passes the +-Inf through to the solver)
        Arows = sp.vstack([(-F)[fin_l], F[fin_u]]).tocsr()
        brows = np.concatenate([-l_[fin_l], u_[fin_u]])
        if smooth_cstr == "logbarrier":  # :251-264
            G_left, G_right, hh = smoothen_linear_inequalities_py(Arows, brows, smooth_alpha, method="logbarrier", solver=solver)
            augment_cone_problem_py(prob, 0, [], Arows.shape[0], G_left, G_right, hh, np.zeros(G_left.shape[1]), np.ones(G_right.shape[1]))
        elif smooth_cstr == "squareplus":  # :265-279
            G_left, G_right, hh = smoothen_linear_inequalities_py(Arows, brows, smooth_alpha, smooth_beta, method="squareplus", solver=solver)
            augment_cone_problem_py(prob, 0, [3] * Arows.shape[0], 0, G_left, G_right, hh, np.zeros(G_left.shape[1]), np.ones(G_right.shape[1]))
        elif smooth_cstr == "":          # :280-287
            augment_cone_problem_py(prob, Arows.shape[0], [], 0, Arows, sp.csr_matrix((Arows.shape[0], 0)), brows, np.zeros(prob.c.size), np.zeros(0))
        else:
            raise ValueError(f"Unknown smoothing method: [{smooth_cstr}]")
        # ---- ad hoc constraints (main.jl:293-316; only inside `if size(F, 1) > 0`, as upstream) --------------------------
        for (l, q, e, G_left, G_right, hh, c_left, c_right) in extra_cstrs:
            G_left = sp.csr_matrix(G_left)
            G_right = sp.csr_matrix(G_right) if sp.issparse(G_right) else sp.csr_matrix(np.asarray(G_right, dtype=np.float64).reshape(G_left.shape[0], -1))
            hh = np.asarray(hh, dtype=np.float64).reshape(-1)
            q = [int(v) for v in q]
            if smooth_cstr == "logbarrier":  # :299-312
                assert G_right.shape[1] == 0, "We only support left matrix reformulation"
                Gn_l, Gn_r, hn = smoothen_linear_inequalities_py(G_left[:l], hh[:l], smooth_alpha, method="logbarrier", solver=solver)
                G_left, G_right = sp.vstack([G_left[l:], Gn_l]).tocsr(), Gn_r  # :308 (G_right has the NEW rows only: with cone rows next
                # to the linear ones the row counts differ and augment_cone_problem!'s assertion :105 fails upstream as it does here)
                hh = np.concatenate([hh[l:], hn])
                l, e = 0, e + Gn_l.shape[0] // 3
                c_right = np.ones(G_right.shape[1])
            augment_cone_problem_py(prob, l, q, e, G_left, G_right, hh, c_left, c_right)
    return prob


# -------------------------------------------------------------------------------------------------
# the numeric solve: primal log-barrier path following on  min c'xi  s.t.  A xi = b,  rows in their cones
# -------------------------------------------------------------------------------------------------
def _exp_barrier(S, convention):
    """Barrier value, gradients (k,3) and Hessians (k,3,3) of k exponential-cone row triples S (k,3) = G xi - h.
    "ecos": K = cl{(x, y, z): z > 0, exp(x/z) <= y/z} (the reading of cone_utils.jl:177-183);  "moi": y exp(x/y) <= z, which is the
    same cone with the last two coordinates exchanged.  Barrier: -log(z log(y/z) - x) - log y - log z."""
    S = np.asarray(S, dtype=np.float64).reshape(-1, 3)
    x = S[:, 0]
    y, z = (S[:, 1], S[:, 2]) if convention == "ecos" else (S[:, 2], S[:, 1])
    if np.any(y <= 0) or np.any(z <= 0):
        return None
    w = z * np.log(y / z) - x
    if np.any(w <= 0):
        return None
    k = S.shape[0]
    val = float(-np.sum(np.log(w)) - np.sum(np.log(y)) - np.sum(np.log(z)))
    gw = np.stack([-np.ones(k), z / y, np.log(y / z) - 1.0], axis=1)           # dw / d(x, y, z)
    Hw = np.zeros((k, 3, 3))
    Hw[:, 1, 1], Hw[:, 1, 2], Hw[:, 2, 1], Hw[:, 2, 2] = -z / y ** 2, 1.0 / y, 1.0 / y, -1.0 / z
    g = -gw / w[:, None]
    g[:, 1] -= 1.0 / y
    g[:, 2] -= 1.0 / z
    H = gw[:, :, None] * gw[:, None, :] / (w ** 2)[:, None, None] - Hw / w[:, None, None]
    H[:, 1, 1] += 1.0 / y ** 2
    H[:, 2, 2] += 1.0 / z ** 2
    if convention != "ecos":  # back to the row order (x, z, y)
        idx = [0, 2, 1]
        g, H = g[:, idx], H[:, idx][:, :, idx]
    return val, g, H


def conic_solve_py(prob: ConeProblem, xi0, mu0=1.0, mu_final=1e-12, exp_convention="ecos", verbose=False):
    """min c'xi s.t. A xi = b, h - G xi >= 0 (l rows), G xi - h in SOC (each q_k rows), in K_exp (each 3 rows): the problem JuMP_solve
    states (cone_solver.jl:163-188).  Primal barrier: Newton on c'xi + mu Phi(xi) to a decrement below 1e-12 at every mu, mu -> mu_final.
    `xi0` must be strictly feasible for the cones and satisfy A xi = b."""
    G, A, c, h, b = prob.G.tocsr(), prob.A.tocsr(), prob.c, prob.h, prob.b
    n = c.size
    xi = np.asarray(xi0, dtype=np.float64).copy()
    assert np.max(np.abs(A @ xi - b), initial=0.0) <= 1e-8 * max(1.0, np.max(np.abs(b), initial=0.0)), "xi0 violates the equalities"
    l, qs, e = prob.l, prob.q, prob.e
    offs = np.concatenate([[l], l + np.cumsum(qs)]).astype(int)
    e0 = int(offs[-1])

    def barrier(xi):
        s = G @ xi - h
        sl = -s[:l]
        if np.any(sl <= 0):
            return None
        val = -np.sum(np.log(sl))
        gs = np.zeros_like(s)
        gs[:l] = 1.0 / sl  # d/ds of -log(-s)
        blocks = [sp.diags(1.0 / sl ** 2)] if l else []
        for k, qk in enumerate(qs):
            sk = s[offs[k]:offs[k] + qk]
            d = sk[0] * sk[0] - float(sk[1:] @ sk[1:])
            if sk[0] <= 0 or d <= 0:
                return None
            Js = np.concatenate([[sk[0]], -sk[1:]])
            val -= np.log(d)
            gs[offs[k]:offs[k] + qk] = -2.0 * Js / d
            Jm = np.diag(np.concatenate([[1.0], -np.ones(qk - 1)]))
            blocks.append(4.0 * np.outer(Js, Js) / (d * d) - 2.0 * Jm / d)
        if e:
            r = _exp_barrier(s[e0:e0 + 3 * e], exp_convention)
            if r is None:
                return None
            val += r[0]
            gs[e0:e0 + 3 * e] = r[1].reshape(-1)
            ii = (np.arange(3 * e).reshape(e, 3, 1) + np.zeros((1, 1, 3), dtype=int)).ravel()
            jj = (np.arange(3 * e).reshape(e, 1, 3) + np.zeros((1, 3, 1), dtype=int)).ravel()
            blocks.append(sp.csr_matrix((r[2].ravel(), (ii, jj)), shape=(3 * e, 3 * e)))
        W = sp.block_diag(blocks, format="csr") if blocks else sp.csr_matrix((0, 0))
        return val, G.T @ gs, (G.T @ W @ G).tocsc()

    if barrier(xi) is None:
        raise ValueError("conic_solve_py: xi0 is not strictly feasible")
    mu, newton = mu0, 0
    stalled = []
    zero_eq = np.zeros(A.shape[0])
    while True:
        for _ in range(1000):
            val, g, H = barrier(xi)
            grad = c + mu * g
            dxi, _, _ = orc._kkt_solve((mu * H).tocsc(), A.tocsc(), sp.csc_matrix((0, n)), -grad, zero_eq, np.zeros(0))
            dec = float(-grad @ dxi)
            if dec < 0.0 and abs(dec) <= 1e-9:  # round-off of a converged iterate
                break
            if dec < 0.0:
                raise RuntimeError(f"conic oracle: negative Newton decrement {dec:.3e} at mu {mu:.1e} (KKT solve lost its accuracy)")
            if dec <= 1e-12 * max(1.0, mu):
                break
            # damped Newton of a self-concordant function (c'xi / mu + Phi): with the decrement lam = sqrt(dec / mu) in the barrier's
            # local norm the step 1 / (1 + lam) stays inside the cones and decreases the merit function; full steps once lam is small
            lam = np.sqrt(dec / mu)
            t = 1.0 if lam <= 0.25 else 1.0 / (1.0 + lam)
            m0 = float(c @ xi) + mu * val
            while True:
                xt = xi + t * dxi
                bt = barrier(xt)
                if bt is not None and float(c @ xt) + mu * bt[0] <= m0 - 1e-4 * t * dec:
                    break
                t *= 0.5
                if t < 1e-12:
                    xt = None
                    break
            if xt is None:
                # the merit function c'xi + mu Phi no longer resolves the decrease (round-off of c'xi against a decrement of `dec`):
                # at the precision floor of this mu.  A genuine failure (large decrement) is an error.
                if dec <= 1e-7 * max(1.0, abs(m0)) * 1e-3 or dec <= 1e-9:
                    stalled.append((mu, dec))
                    break
                raise RuntimeError(f"conic oracle: line search failed (mu {mu:.1e}, decrement {dec:.3e})")
            xi = xt
            newton += 1
        if verbose:
            print(f"conic oracle: mu {mu:.1e}  objective {float(c @ xi):.12g}  newton {newton}")
        if mu <= mu_final:
            break
        mu = max(0.2 * mu, mu_final)
    return xi, dict(newton=newton, mu=mu, objective=float(c @ xi), stalled=stalled)


def strictly_feasible_start(prob: ConeProblem, U0, exp_convention="ecos", margin=1.0):
    """A strictly feasible xi0 from controls U0 (M,N,u) strictly inside their boxes (shared stages: particle 0's): the states follow
    from the dynamics rows, every added variable (y, t, the smoothing variables, each in exactly one row triple or cone) is pushed
    `margin` inside its row.  Rows the trajectory itself violates (state boxes, user cones) raise."""
    xdim, udim, N, M, Nc = prob.dims
    nz, n = prob.nz, prob.c.size
    Nf = N - Nc
    ncu = Nc * udim + M * Nf * udim
    U0 = np.asarray(U0, dtype=np.float64)
    z = np.zeros(nz)
    z[:Nc * udim] = U0[0, :Nc].reshape(-1)
    z[Nc * udim:ncu] = U0[:, Nc:].reshape(-1)
    Az = prob.A.tocsc()[:, :nz]
    z[ncu:] = spla.spsolve(Az[:, ncu:].tocsc(), prob.b - Az[:, :ncu] @ z[:ncu])
    xi = np.zeros(n)
    xi[:nz] = z
    G, h = prob.G.tocsr(), prob.h
    # (y, t) near the central path of mu = margin: the cone rows read  d_i = t + y_i - J_i(z) > 0  (the two head rows of Pqr2Gh differ by
    # 1/2, so s0^2 - |s_tail|^2 = tau - J_i), the multipliers on the path are lam_i = mu / d_i, nu_i = mu / y_i with
    # lam_i + nu_i = 1 + eps and sum lam_i = (1 - eps) k: lam ~ (1 - eps) k / M leaves nu ~ 2 eps for k = M, i.e. y ~ mu / (2 eps).
    # A start with y of order one is hundreds of damped Newton steps away from that.
    l, qs = prob.l, prob.q
    offs = np.concatenate([[l], l + np.cumsum(qs)]).astype(int)
    s = G @ xi - h
    J = np.zeros(M)
    for i in range(M):
        sk = s[offs[i]:offs[i] + qs[i]]
        alf, bet = h[offs[i]], h[offs[i] + 1]
        J[i] = float(sk[2:] @ sk[2:]) + 0.5 * (alf + bet)  # (tau - alf)^2 = (tau - bet)^2 + |w|^2  <=>  tau = |w|^2 + (alf + bet) / 2
    lam_bar = float(prob.c[nz + M]) / M
    nu_bar = max(float(prob.c[nz]) - lam_bar, 1e-6)
    y0 = J - J.min() + margin / nu_bar
    xi[nz:nz + M] = y0
    xi[nz + M] = J.min() + margin / lam_bar - margin / nu_bar
    # every further new variable sits in exactly one SOC (squareplus) or exp triple (logbarrier): push it inside
    s = G @ xi - h
    Gc = G.tocsc()
    for v in range(nz + M + 1, n):
        rows = Gc.indices[Gc.indptr[v]:Gc.indptr[v + 1]]
        vals = Gc.data[Gc.indptr[v]:Gc.indptr[v + 1]]
        assert rows.size == 1, "new variable in more than one row: pass xi0 yourself"
        r0, g0 = int(rows[0]), float(vals[0])
        if r0 >= offs[-1]:  # exp triple, first row: x = g0 * v (g0 = -alpha)
            k3 = (r0 - offs[-1]) // 3
            trip = s[offs[-1] + 3 * k3:offs[-1] + 3 * k3 + 3]
            if exp_convention == "ecos":
                x_max = trip[2] * np.log(trip[1] / trip[2])  # x < z log(y / z)
            else:
                x_max = trip[1] * np.log(trip[2] / trip[1])
            assert np.isfinite(x_max), "a smoothed row is violated by the start"
            xi[v] = (x_max - margin * abs(g0) - (trip[0])) / g0 if g0 < 0 else (x_max - margin) / g0
            # (trip[0] = 0 here: the variable is zero so far)
        else:  # SOC head row: s0 = g0 * v + rest >= |tail|
            kq = int(np.searchsorted(offs, r0, side="right") - 1)
            sk = s[offs[kq]:offs[kq] + qs[kq]]
            assert r0 == offs[kq] and g0 > 0
            xi[v] = (np.linalg.norm(sk[1:]) + margin - sk[0]) / g0
    return xi


def lcone_direct_py(x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref, *, reg_x, reg_u, Nc=-1, U0=None, return_info=False, mu_final=1e-12,
                    exp_convention="ecos", verbose=False, **kw):
    """The reference's cone program, built as main.jl:204-316 builds it and solved as stated.  `U0` (M,N,u): strictly feasible controls
    (default: the centre of finite control boxes, else U_prev).  Returns X (M,N,x), U (M,N,u)."""
    prob = lcone_problem_py(x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref, reg_x=reg_x, reg_u=reg_u, Nc=Nc, **kw)
    xdim, udim, N, M, Ncc = prob.dims
    if U0 is None:
        # START only (the answer is whatever the conic solve converges to): the plain-sum QP with its boxes behind a log barrier of
        # weight 1 is close to the central path of mu = 1 in the trajectory variables; a start at the box centre is hundreds of damped
        # Newton steps away from it
        qkw = {k_: kw.get(k_) for k_ in ("x_l", "x_u", "u_l", "u_u", "slew_reg", "slew_reg0", "slew_um1")}
        has_box = any(qkw[k_] is not None for k_ in ("x_l", "x_u", "u_l", "u_u"))
        # (linear rows of extra_cstrs tuples join the boxes behind the barrier, so that the start is strictly inside them too)
        rws = [(sp.csr_matrix(t[3])[:int(t[0])], np.asarray(t[5], dtype=np.float64).reshape(-1)[:int(t[0])]) for t in kw.get("extra_cstrs", ()) if int(t[0]) > 0]
        rows = None
        if rws:
            Gs = [sp.hstack([g, sp.csr_matrix((g.shape[0], prob.nz - g.shape[1]))]) if g.shape[1] < prob.nz else g[:, :prob.nz] for g, _ in rws]
            rows = (sp.vstack(Gs).tocsr(), np.concatenate([h_ for _, h_ in rws]))
        _, U0 = orc.lqp_solve_py(x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref, reg_x=reg_x, reg_u=reg_u, Nc=Nc,
                                 barrier_mu=1.0 if (has_box or rows is not None) else 0.0, rows=rows, **qkw)
    xi0 = strictly_feasible_start(prob, U0, exp_convention=exp_convention)
    xi, info = conic_solve_py(prob, xi0, mu_final=mu_final, exp_convention=exp_convention, verbose=verbose)
    qp = orc.JointQP()
    qp.dims, qp.Nc = (xdim, udim, N, M), Ncc
    X, U = orc.split_vars(qp, xi[:prob.nz])
    info.update(y=xi[prob.nz:prob.nz + M], t=float(xi[prob.nz + M]), n_vars=xi.size, rows=prob.G.shape[0])
    return (X, U, info) if return_info else (X, U)
