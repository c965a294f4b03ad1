"""`lqp_generate_problem_matrices` — the canonical joint-QP representation `(P, q, A, b, G, l, u)` that the
reference hands to OSQP, for users who inspect or re-solve it themselves (pmpc/scp_mpc.py:66-75 ->
PMPC.jl/src/main.jl:374-409 -> lqp_repr_Pq / lqp_repr_Ab / lqp_repr_Gla, PMPC.jl/src/lqp_utils.jl:2-393).

    min_z 1/2 z'Pz + q'z   s.t.  A z = b,  l <= G z <= u,
    z = [U_cons (Nc*udim) ; U_free particle-major (M*Nf*udim) ; X particle-major (M*N*xdim)]

Host-side numpy/scipy.sparse (vectorised COO assembly); the HIP solver never forms these matrices.  Arrays
are in the Python layout of `pmpc.solve`: vectors `(M, N, d)`, matrices `(M, N, row, col)`.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from .utils import atleast_nd, to_numpy_f64


def _per_particle(v, M, default=0.0):
    v = default if v is None else v
    v = np.asarray(v, dtype=np.float64)
    return np.full(M, float(v)) if v.ndim == 0 else v.reshape(M)


def lqp_generate_problem_matrices(x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref, **settings):
    """Returns `(P, q, A, b, G, l, u)`; `P`, `A`, `G` are `scipy.sparse.csc_matrix` (P with both triangles, as
    the reference builds it).  Settings as in the reference: `Nc` (default -1 = N, main.jl:377-378), `reg_x`,
    `reg_u` (default 0, main.jl:19), `slew_reg`, `slew_reg0`, `slew_um1`, `lx`, `ux`, `lu`, `uu`."""
    x0 = atleast_nd(to_numpy_f64(x0), 2)
    f, X_prev, U_prev, X_ref, U_ref = [atleast_nd(to_numpy_f64(z), 3) for z in (f, X_prev, U_prev, X_ref, U_ref)]
    fx, fu, Q, R = [atleast_nd(to_numpy_f64(z), 4) for z in (fx, fu, Q, R)]
    M, N, x, u = fu.shape
    Nc = int(settings.get("Nc", -1))
    Nc = N if Nc < 0 else min(Nc, N)
    Nf = N - Nc
    reg_x, reg_u = _per_particle(settings.get("reg_x"), M), _per_particle(settings.get("reg_u"), M)
    s, s0 = _per_particle(settings.get("slew_reg"), M), _per_particle(settings.get("slew_reg0"), M)
    um1 = settings.get("slew_um1")
    um1 = np.zeros((M, u)) if um1 is None else np.broadcast_to(np.asarray(um1, dtype=np.float64), (M, u))
    n = Nc * u + M * (Nf * u + N * x)

    # variable indices (lqp_utils.jl:12-15, :109): cu[i, j, r] consensus for j < Nc, free otherwise
    I, J = np.arange(M)[:, None, None], np.arange(N)[None, :, None]
    ru, rx = np.arange(u)[None, None, :], np.arange(x)[None, None, :]
    cu = np.where(J < Nc, u * J + ru, Nc * u + Nf * u * I + u * (J - Nc) + ru) + 0 * I
    cx = Nc * u + M * Nf * u + N * x * I + x * J + rx

    # ---- P (lqp_utils.jl:17-160) ---------------------------------------------------------------------------
    rows, cols, vals = [], [], []

    def add(r, c, v):
        r, c, v = np.broadcast_arrays(r, c, v)
        rows.append(r.ravel()), cols.append(c.ravel()), vals.append(v.ravel())

    add(cu[:, :, :, None], cu[:, :, None, :], R)  # core control cost, summed over particles on consensus stages
    sdiag = np.where(J[..., 0] == 0, (s0 + s)[:, None], np.where(J[..., 0] == N - 1, s[:, None], 2.0 * s[:, None]))  # :31-39
    add(cu, cu, (reg_u[:, None] + sdiag)[:, :, None])
    if N > 1:  # slew tridiagonal (:21-25, :47-57, :67-75, :94-98), both triangles
        off = np.broadcast_to(-s[:, None, None], (M, N - 1, u))
        add(cu[:, 1:], cu[:, :-1], off)
        add(cu[:, :-1], cu[:, 1:], off)
    add(cx[:, :, :, None], cx[:, :, None, :], Q)  # :130-141
    add(cx, cx, np.broadcast_to(reg_x[:, None, None], (M, N, x)))
    P = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n)).tocsc()
    P.eliminate_zeros()  # the reference skips zero values (:41, :89, :132)

    # ---- q (lqp_utils.jl:161-213) --------------------------------------------------------------------------
    q = np.zeros(n)
    qu = -reg_u[:, None, None] * U_prev - np.einsum("mnrt,mnt->mnr", R, U_ref)
    np.add.at(q, cu.ravel(), qu.ravel())
    q[cx.ravel()] = (-reg_x[:, None, None] * X_prev - np.einsum("mnrt,mnt->mnr", Q, X_ref)).ravel()
    if Nc >= 1:  # :165 — with Nc == 0 the free-control assignment (:190) overwrites it
        q[:u] += -(s0[:, None] * um1).sum(0)

    # ---- A, b (lqp_utils.jl:219-303): fu_j u_j - x_j + fx_j x_{j-1} = -f_j + fu_j U_prev_j + fx_j X_prev_{j-1} ---
    eq = N * x * I + x * J + rx
    rows, cols, vals = [], [], []
    add(eq[:, :, :, None], cu[:, :, None, :], fu)
    add(eq, cx, -np.ones((M, N, x)))
    if N > 1:
        add(eq[:, 1:, :, None], cx[:, :-1, None, :], fx[:, 1:])
    A = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(M * N * x, n)).tocsc()
    b = -f + np.einsum("mnrt,mnt->mnr", fu, U_prev)
    b[:, 1:] += np.einsum("mnrt,mnt->mnr", fx[:, 1:], X_prev[:, :-1])
    b = b.ravel()

    # ---- G, l, u (lqp_utils.jl:306-393): one unit row per bounded variable; consensus bounds from particle 0 ---
    lx, ux, lu, uu = (settings.get(k) for k in ("lx", "ux", "lu", "uu"))
    has_ub = lu is not None and uu is not None and np.size(lu) > 0 and np.size(uu) > 0
    has_xb = lx is not None and ux is not None and np.size(lx) > 0 and np.size(ux) > 0
    gcols, lo, hi = [], [], []
    if has_ub:
        lu, uu = (np.broadcast_to(atleast_nd(to_numpy_f64(z), 3), (M, N, u)) for z in (lu, uu))
        gcols += [np.arange(Nc * u), cu[:, Nc:].ravel()]
        lo += [lu[0, :Nc].ravel(), lu[:, Nc:].ravel()]
        hi += [uu[0, :Nc].ravel(), uu[:, Nc:].ravel()]
    if has_xb:
        lx, ux = (np.broadcast_to(atleast_nd(to_numpy_f64(z), 3), (M, N, x)) for z in (lx, ux))
        gcols.append(cx.ravel()), lo.append(lx.ravel()), hi.append(ux.ravel())
    gcols = np.concatenate(gcols) if gcols else np.zeros(0, dtype=np.int64)
    m_in = gcols.size
    G = sp.coo_matrix((np.ones(m_in), (np.arange(m_in), gcols)), shape=(m_in, n)).tocsc()
    lo = np.concatenate(lo) if lo else np.zeros(0)
    hi = np.concatenate(hi) if hi else np.zeros(0)
    return P, q, A, b, G, lo, hi
