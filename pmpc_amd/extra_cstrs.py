"""The reference's conic-constraint tuples `(l, q, e, G_left, G_right, h, c_left, c_right)` — `G z - h in K` over the joint
variable vector `z = [U_cons (Nc*udim); U_free (M*(N-Nc)*udim); X (M*N*xdim)]` (README.md:219-239 and "Variable Layout";
PMPC.jl/src/main.jl:293-316, cone_solver.jl:166-177) — recognised in the structured case the device solver implements:
second-order cones only, every cone on the controls of ONE stage (of one particle, or of the shared consensus block), the
same `(W, w0, v, v0)` for all of them:

    row 0 of a cone:  v'u - (-v0)   = t        rows 1..q:  W u - (-w0) = x        ||x||_2 <= t.

`stage_soc_from_extra_cstrs` returns `dict(W, w0, v, v0)` for `DeviceSolver.lsoc_solve` / `solve(..., device="cuda", soc=...)`,
and raises `ValueError` saying which part of the structure is outside that case (linear / exponential cones, new variables,
cost terms, state columns, cones spanning stages, stage-dependent data, stages without a cone)."""
from __future__ import annotations

from typing import Any, Dict, Sequence

import numpy as np
import scipy.sparse as sp


def stage_soc_from_extra_cstrs(cstr: Sequence[Any], M: int, N: int, xdim: int, udim: int, Nc: int, atol: float = 0.0) -> Dict[str, Any]:
    l, q, e, G_left, G_right, h, c_left, c_right = cstr
    q = [int(v) for v in np.atleast_1d(np.asarray(q, dtype=np.int64))] if np.size(q) else []
    Nc = N if Nc < 0 else min(int(Nc), N)
    Nf = N - Nc
    ncu = Nc * udim + M * Nf * udim
    n = ncu + M * N * xdim
    if int(l) != 0 or int(e) != 0:
        raise ValueError("only second-order cones are supported (l = e = 0)")
    if G_right is not None and np.size(G_right) > 0 and sp.csr_matrix(G_right).shape[1] > 0:
        raise ValueError("cones that introduce new variables (G_right) are not supported")
    for c in (c_left, c_right):
        if c is not None and np.size(c) > 0 and np.any(np.asarray(c, dtype=np.float64) != 0.0):
            raise ValueError("cost augmentation (c_left / c_right) is not supported")
    G = sp.csr_matrix(G_left, dtype=np.float64)
    h = np.asarray(h, dtype=np.float64).reshape(-1)
    if G.shape != (sum(q), n) or h.size != sum(q):
        raise ValueError(f"G_left must be ({sum(q)}, {n}) over z = [U_cons; U_free; X], h ({sum(q)},)")
    if len(set(q)) != 1:
        raise ValueError("all cones must have the same size")
    if G[:, ncu:].count_nonzero() > 0:
        raise ValueError("cones on the states are not supported (only the controls of one stage)")
    nblocks = ncu // udim  # Nc shared control blocks, then M*Nf free ones
    if len(q) != nblocks:
        raise ValueError(f"one cone per control block is required: {nblocks} blocks (Nc shared + M*(N-Nc) free), {len(q)} cones")
    qq = q[0]
    seen = np.zeros(nblocks, dtype=bool)
    ref = None
    for k in range(len(q)):
        rows = G[k * qq:(k + 1) * qq]
        cols = np.unique(rows.indices)
        if cols.size == 0:
            raise ValueError(f"cone {k} does not touch any control")
        blk = int(cols[0]) // udim
        if np.any(cols // udim != blk):
            raise ValueError(f"cone {k} couples the controls of several stages")
        if seen[blk]:
            raise ValueError(f"control block {blk} carries more than one cone")
        seen[blk] = True
        A = rows[:, blk * udim:(blk + 1) * udim].toarray()
        hk = h[k * qq:(k + 1) * qq]
        cur = (A, hk)
        if ref is None:
            ref = cur
        elif not (np.allclose(A, ref[0], rtol=0.0, atol=atol) and np.allclose(hk, ref[1], rtol=0.0, atol=atol)):
            raise ValueError(f"cone {k} differs from cone 0: stage-dependent cone data are not supported")
    A, hk = ref
    return dict(W=A[1:].copy(), w0=-hk[1:].copy(), v=A[0].copy(), v0=float(-hk[0]))


def stage_soc_to_extra_cstrs(W, w0, v, v0, M: int, N: int, xdim: int, udim: int, Nc: int):
    """The inverse: the reference-format tuple of the stage-wise cone ||W u + w0|| <= v'u + v0 (what a pyjulia user of the
    reference would pass in `extra_cstrs` / return from `extra_cstrs_fns`)."""
    W, w0, v = np.atleast_2d(np.asarray(W, float)), np.asarray(w0, float).reshape(-1), np.asarray(v, float).reshape(-1)
    Nc = N if Nc < 0 else min(int(Nc), N)
    ncu = Nc * udim + M * (N - Nc) * udim
    n = ncu + M * N * xdim
    A = np.vstack([v[None, :], W])
    qq = A.shape[0]
    nblocks = ncu // udim
    G = sp.kron(sp.identity(nblocks, format="csr"), sp.csr_matrix(A), format="csr")
    G = sp.hstack([G, sp.csr_matrix((nblocks * qq, n - ncu))], format="csr")
    h = np.tile(np.concatenate([[-float(v0)], -w0]), nblocks)
    return (0, [qq] * nblocks, 0, G, sp.csr_matrix((nblocks * qq, 0)), h, np.zeros(n), np.zeros(0))


def linear_rows_to_boxes(cstr: Sequence[Any], M: int, N: int, xdim: int, udim: int, Nc: int):
    """Linear rows `G z <= h` (the `l` part of a tuple, PMPC.jl/src/cone_solver.jl:163-166) with ONE nonzero per row are box
    constraints on single variables.  Returns `(x_l, x_u, u_l, u_u)` as (M, N, d) arrays with -inf / +inf where nothing is
    imposed — to be intersected with the caller's boxes — or raises `ValueError` if the tuple is anything else (cones, new
    variables, cost terms, rows that couple variables).  A row on a consensus control (shared column) bounds that control for
    every particle; the joint QP takes particle 0's bound there (lqp_utils.jl:329-330)."""
    l, q, e, G_left, G_right, h, c_left, c_right = cstr
    Nc = N if Nc < 0 else min(int(Nc), N)
    Nf = N - Nc
    ncu = Nc * udim + M * Nf * udim
    n = ncu + M * N * xdim
    if (np.size(q) and sum(int(v) for v in np.atleast_1d(q)) > 0) or int(e) != 0:
        raise ValueError("not a purely linear tuple (q / e non-empty)")
    if G_right is not None and np.size(G_right) > 0 and sp.csr_matrix(G_right).shape[1] > 0:
        raise ValueError("rows that introduce new variables (G_right) are not supported")
    for c in (c_left, c_right):
        if c is not None and np.size(c) > 0 and np.any(np.asarray(c, dtype=np.float64) != 0.0):
            raise ValueError("cost augmentation (c_left / c_right) is not supported")
    G = sp.csr_matrix(G_left, dtype=np.float64)
    G.eliminate_zeros()
    h = np.asarray(h, dtype=np.float64).reshape(-1)
    if G.shape != (int(l), n) or h.size != int(l):
        raise ValueError(f"G_left must be ({int(l)}, {n}) over z = [U_cons; U_free; X], h ({int(l)},)")
    if np.any(np.diff(G.indptr) != 1):
        raise ValueError("rows that couple several variables are not supported (one nonzero per row: a box on one variable)")
    x_l, x_u = np.full((M, N, xdim), -np.inf), np.full((M, N, xdim), np.inf)
    u_l, u_u = np.full((M, N, udim), -np.inf), np.full((M, N, udim), np.inf)
    for col, coef, rhs in zip(G.indices, G.data, h):
        bound = rhs / coef
        if col < Nc * udim:  # shared control: stage j, component r, every particle
            j, r = divmod(int(col), udim)
            lo, hi, idx = u_l, u_u, (slice(None), j, r)
        elif col < ncu:
            i, rest = divmod(int(col) - Nc * udim, Nf * udim)
            j, r = divmod(rest, udim)
            lo, hi, idx = u_l, u_u, (i, Nc + j, r)
        else:
            i, rest = divmod(int(col) - ncu, N * xdim)
            j, r = divmod(rest, xdim)
            lo, hi, idx = x_l, x_u, (i, j, r)
        if coef > 0:
            hi[idx] = np.minimum(hi[idx], bound)
        else:
            lo[idx] = np.maximum(lo[idx], bound)
    return x_l, x_u, u_l, u_u


def stage_cones_from_extra_cstrs(cstrs: Sequence[Sequence[Any]], M: int, N: int, xdim: int, udim: int, Nc: int) -> Dict[str, Any]:
    """Several `extra_cstrs` tuples -> the device solver's general stage-cone form (`DeviceSolver.lsoc_solve(cones=...)`).
    Accepted: linear rows (`l`, rows  G z <= h  — cone_solver.jl:163-166) and second-order cones (`q`, G z - h in SOC,
    cone_solver.jl:167-177) whose nonzeros all lie on the controls of ONE stage block — one particle's free stage, or one shared
    consensus stage — with the SAME list of cone sizes on every block (the data may differ from stage to stage).  Refused with
    the reason: exponential cones, new variables, cost terms, rows on states, rows spanning stages, blocks with different
    structures, more than 4 cones / 8 rows per stage or cones larger than 4 rows.
    Returns dict(sizes=[q_k], A, c) with `s = A u + c`: per-stage arrays `A (M, N, rows, udim)`, `c (M, N, rows)` (a shared
    control's block is replicated over the particles)."""
    Ncc = N if Nc < 0 else min(int(Nc), N)
    Nf = N - Ncc
    ncu = Ncc * udim + M * Nf * udim
    n = ncu + M * N * xdim
    nblocks = ncu // udim
    per_block = [[] for _ in range(nblocks)]  # (q, A rows (q+1, udim), c (q+1))
    for cstr in cstrs:
        l, q, e, G_left, G_right, h, c_left, c_right = cstr
        q = [int(v) for v in np.atleast_1d(np.asarray(q, dtype=np.int64))] if np.size(q) else []
        if int(e) != 0:
            raise ValueError("exponential cones are not supported")
        if G_right is not None and np.size(G_right) > 0 and sp.csr_matrix(G_right).shape[1] > 0:
            raise ValueError("rows that introduce new variables (G_right) are not supported")
        for cv in (c_left, c_right):
            if cv is not None and np.size(cv) > 0 and np.any(np.asarray(cv, dtype=np.float64) != 0.0):
                raise ValueError("cost augmentation (c_left / c_right) is not supported")
        G = sp.csr_matrix(G_left, dtype=np.float64)
        hv = np.asarray(h, dtype=np.float64).reshape(-1)
        nr = int(l) + sum(q)
        if G.shape[0] != nr or hv.size != nr or G.shape[1] > n:
            raise ValueError(f"G_left must have {nr} rows over (a prefix of) z = [U_cons; U_free; X] ({n} columns), h ({nr},)")
        if G.shape[1] > ncu and G[:, ncu:].count_nonzero() > 0:
            raise ValueError("rows on the states are not supported (only the controls of one stage)")
        G = G[:, :min(G.shape[1], ncu)]
        pieces = [(0, r, r + 1) for r in range(int(l))]
        r0 = int(l)
        for qk in q:
            if qk > 4 or qk < 2:
                raise ValueError("second-order cones of 2 .. 4 rows are supported")
            pieces.append((qk - 1, r0, r0 + qk))
            r0 += qk
        for qs, ra, rb in pieces:
            rows = G[ra:rb]
            cols = np.unique(rows.indices)
            if cols.size == 0:
                raise ValueError(f"rows {ra}..{rb - 1} touch no control")
            blk = int(cols[0]) // udim
            if np.any(cols // udim != blk):
                raise ValueError(f"rows {ra}..{rb - 1} couple the controls of several stages")
            Ablk = np.zeros((rb - ra, udim))
            sub = rows[:, blk * udim:min((blk + 1) * udim, rows.shape[1])].toarray()
            Ablk[:, :sub.shape[1]] = sub
            if qs == 0:   # G z <= h  ->  s = h - G u >= 0
                per_block[blk].append((0, -Ablk, hv[ra:rb].copy()))
            else:         # G z - h in SOC  ->  s = G u - h
                per_block[blk].append((qs, Ablk, -hv[ra:rb]))
    sizes = [c_[0] for c_ in per_block[0]]
    if not sizes:
        raise ValueError("the first control block carries no constraint: every stage needs the same list of cones")
    if len(sizes) > 4 or sum(sizes) + len(sizes) > 8:
        raise ValueError("at most 4 cones and 8 rows per stage are supported")
    for b, lst in enumerate(per_block):
        if [c_[0] for c_ in lst] != sizes:
            raise ValueError(f"control block {b} carries cones of sizes {[c_[0] for c_ in lst]}, block 0 {sizes}: every stage needs the same list")
    rows_tot = sum(sizes) + len(sizes)
    A = np.zeros((M, N, rows_tot, udim))
    c = np.zeros((M, N, rows_tot))
    for b, lst in enumerate(per_block):
        Ab, cb = np.vstack([c_[1] for c_ in lst]), np.concatenate([c_[2] for c_ in lst])
        if b < Ncc:
            A[:, b], c[:, b] = Ab, cb
        else:
            i, j = divmod(b - Ncc, Nf)
            A[i, Ncc + j], c[i, Ncc + j] = Ab, cb
    return dict(sizes=sizes, A=A, c=c)



def _decode_column(col: int, M: int, N: int, xdim: int, udim: int, Ncc: int):
    """Column of z = [U_cons; U_free; X] (lqp_utils.jl:12-15, split_lqp_vars :395-423) -> (kind, particle or None, stage, component)."""
    Nf = N - Ncc
    ncu = Ncc * udim + M * Nf * udim
    if col < Ncc * udim:
        j, r = divmod(col, udim)
        return "u", None, j, r
    if col < ncu:
        i, rest = divmod(col - Ncc * udim, Nf * udim)
        j, r = divmod(rest, udim)
        return "u", i, Ncc + j, r
    i, rest = divmod(col - ncu, N * xdim)
    j, r = divmod(rest, xdim)
    return "x", i, j, r


def stage_rows_from_extra_cstrs(cstrs: Sequence[Sequence[Any]], M: int, N: int, xdim: int, udim: int, Nc: int):
    """Linear rows `G z <= h` (the `l` part of `extra_cstrs` tuples, cone_solver.jl:163-166) that couple the STATE and the CONTROL of one
    stage of one particle:  a_x'X[i, t] + a_u'U[i, t] <= h  (array indices: x_{t+1} with u_t, the pair one step of the dynamics links), or
    a_x'X[i, t-1] + a_u'U[i, t] <= h  (x_t with u_t, the pair of one stage of the optimal-control problem); either part may be absent.
    Returns a list of `(i, t, form, a_x, a_u, h)`, form 0 / 1 as above, `t` the index of the state array at which `aux_state_problem`
    bounds the row's value.  Raises `ValueError` with the reason for anything else: cones, new variables, cost terms, rows coupling
    particles or stages further apart."""
    Ncc = N if Nc < 0 else min(int(Nc), N)
    n = Ncc * udim + M * (N - Ncc) * udim + M * N * xdim
    out = []
    for cstr in cstrs:
        l, q, e, G_left, G_right, h, c_left, c_right = cstr
        if (np.size(q) and sum(int(v) for v in np.atleast_1d(q)) > 0) or int(e) != 0:
            raise ValueError("rows on the states are supported as linear rows only (no second-order or exponential cones on states)")
        if G_right is not None and np.size(G_right) > 0 and sp.csr_matrix(G_right).shape[1] > 0:
            raise ValueError("rows that introduce new variables (G_right) are not supported")
        for cv in (c_left, c_right):
            if cv is not None and np.size(cv) > 0 and np.any(np.asarray(cv, dtype=np.float64) != 0.0):
                raise ValueError("cost augmentation (c_left / c_right) is not supported")
        G = sp.csr_matrix(G_left, dtype=np.float64)
        G.eliminate_zeros()
        hv = np.asarray(h, dtype=np.float64).reshape(-1)
        if G.shape[0] != int(l) or hv.size != int(l) or G.shape[1] > n:
            raise ValueError(f"G_left must have {int(l)} rows over (a prefix of) z = [U_cons; U_free; X] ({n} columns), h ({int(l)},)")
        for r in range(int(l)):
            cols, vals = G.indices[G.indptr[r]:G.indptr[r + 1]], G.data[G.indptr[r]:G.indptr[r + 1]]
            if cols.size == 0:
                raise ValueError(f"row {r} touches no variable")
            a_x, a_u = np.zeros(xdim), np.zeros(udim)
            parts, sx, su = set(), set(), set()
            for col, v in zip(cols, vals):
                kind, i, j, comp = _decode_column(int(col), M, N, xdim, udim, Ncc)
                if i is not None:
                    parts.add(i)
                if kind == "x":
                    sx.add(j)
                    a_x[comp] = v
                else:
                    su.add(j)
                    a_u[comp] = v
            if len(parts) > 1:
                raise ValueError(f"row {r} couples several particles")
            if len(sx) > 1 or len(su) > 1:
                raise ValueError(f"row {r} couples several stages")
            i = parts.pop() if parts else 0  # (a row on shared controls alone: one row of the joint problem, carried by particle 0)
            if sx and su:
                jx, ju = next(iter(sx)), next(iter(su))
                if ju == jx:
                    out.append((i, jx, 0, a_x, a_u, float(hv[r])))
                elif ju == jx + 1:
                    out.append((i, ju, 1, a_x, a_u, float(hv[r])))
                else:
                    raise ValueError(f"row {r} couples state stage {jx} with control stage {ju}: only the same array index (x_(t+1), u_t) "
                                     "or the same stage (x_t, u_t) are supported")
            elif sx:
                out.append((i, next(iter(sx)), 0, a_x, a_u, float(hv[r])))
            else:
                out.append((i, next(iter(su)), 0, a_x, a_u, float(hv[r])))
    return out


def aux_state_problem(rows, x0, f, fx, fu, X_prev, U_prev, Q, X_ref, reg_x, x_l, x_u):
    """The problem with every row of `stage_rows_from_extra_cstrs` restated as an UPPER BOUND ON AN AUXILIARY STATE (py layout, batched
    arrays in and out): the state grows by `m` components (the largest number of rows any (particle, stage) carries), component `xdim + s`
    of x~_(t+1) is the row's left-hand side as the linearised dynamics produce it,

        form 0:  xi = a_x'x_(t+1) + a_u'u_t = a_x'f_t + a_u'up_t + (a_x'fx_t)(x_t - xp_t) + (a_x'fu_t + a_u')(u_t - up_t)
        form 1:  xi = a_x'x_t + a_u'u_t     = a_x'xp_t + a_u'up_t + a_x'(x_t - xp_t) + a_u'(u_t - up_t)

    (xp, up: the linearisation point, lqp_utils.jl:233-290), and the row is the box  xi <= h  on it — which the device solver holds with
    the semismooth state rows of `kernels_xbox.hip` (hard boxes) or the barrier (smoothing: the reference smooths `extra_cstrs` rows with
    the boxes too, main.jl:298-312).  The auxiliary components carry NO cost: their block of Q is -reg_x and their reference and previous
    value 0, so that the cost term -reg_x/2 xi^2 and the reference's proximal term +reg_x/2 xi^2 cancel exactly (value, gradient, Hessian).
    Returns dict(x0, f, fx, fu, X_prev, Q, X_ref, x_l, x_u, m)."""
    M, N, xdim = f.shape
    udim = fu.shape[-1]
    slot = {}
    for (i, t, form, a_x, a_u, h) in rows:
        if form == 1 and t < 1:
            raise ValueError("a row on (x_t, u_t) needs t >= 1 (x_0 is data)")
        slot.setdefault((i, t), []).append((form, a_x, a_u, h))
    m = max(len(v) for v in slot.values())
    xd = xdim + m
    F = np.zeros((M, N, xd)); F[..., :xdim] = f
    FX = np.zeros((M, N, xd, xd)); FX[..., :xdim, :xdim] = fx
    FU = np.zeros((M, N, xd, udim)); FU[..., :xdim, :] = fu
    XP = np.zeros((M, N, xd)); XP[..., :xdim] = X_prev
    XR = np.zeros((M, N, xd)); XR[..., :xdim] = X_ref
    QQ = np.zeros((M, N, xd, xd)); QQ[..., :xdim, :xdim] = Q
    QQ[..., np.arange(xdim, xd), np.arange(xdim, xd)] = -float(reg_x)
    X0 = np.zeros((M, xd)); X0[:, :xdim] = x0
    lo = np.full((M, N, xd), -np.inf); hi = np.full((M, N, xd), np.inf)
    if x_l is not None and np.size(x_l):
        lo[..., :xdim] = np.broadcast_to(x_l, (M, N, xdim))
    if x_u is not None and np.size(x_u):
        hi[..., :xdim] = np.broadcast_to(x_u, (M, N, xdim))
    lo[np.isnan(lo)], hi[np.isnan(hi)] = -np.inf, np.inf
    for (i, t), lst in slot.items():
        up = U_prev[i, t]
        for s, (form, a_x, a_u, h) in enumerate(lst):
            c = xdim + s
            if form == 0:
                F[i, t, c] = a_x @ f[i, t] + a_u @ up
                FX[i, t, c, :xdim] = a_x @ fx[i, t]
                FU[i, t, c] = a_x @ fu[i, t] + a_u
            else:
                F[i, t, c] = a_x @ X_prev[i, t - 1] + a_u @ up
                FX[i, t, c, :xdim] = a_x
                FU[i, t, c] = a_u
            hi[i, t, c] = h
    return dict(x0=X0, f=F, fx=FX, fu=FU, X_prev=XP, Q=QQ, X_ref=XR, x_l=lo, x_u=hi, m=m)


def split_linear_cost(cstr: Sequence[Any], M: int, N: int, xdim: int, udim: int, Nc: int):
    """`c_left` of a tuple (PMPC.jl/src/cone_utils.jl:152-154: added to the cost vector of the cone program, OUTSIDE the epigraph rows) as
    per-stage arrays `cu (M, N, udim)`, `cx (M, N, xdim)` over z = [U_cons; U_free; X] (a shared control's entry lands in particle 0's
    block), plus the tuple with `c_left` zeroed.  Returns `(None, None, cstr)` when there is no cost term.  Entries beyond the trajectory
    variables (the y / t columns of the cone program) are refused."""
    l, q, e, G_left, G_right, h, c_left, c_right = cstr
    cl = np.zeros(0) if c_left is None else np.asarray(c_left, dtype=np.float64).reshape(-1)
    if cl.size == 0 or not np.any(cl != 0.0):
        return None, None, cstr
    Ncc = N if Nc < 0 else min(int(Nc), N)
    Nf = N - Ncc
    ncu = Ncc * udim + M * Nf * udim
    n = ncu + M * N * xdim
    if cl.size > n and np.any(cl[n:] != 0.0):
        raise ValueError("c_left on the epigraph variables (y, t) of the cone program is not supported")
    full = np.zeros(n)
    full[:min(cl.size, n)] = cl[:n]
    cu = np.zeros((M, N, udim))
    cu[0, :Ncc] = full[:Ncc * udim].reshape(Ncc, udim)
    if Nf:
        cu[:, Ncc:] = full[Ncc * udim:ncu].reshape(M, Nf, udim)
    cx = full[ncu:].reshape(M, N, xdim)
    return cu, cx, (l, q, e, G_left, G_right, h, np.zeros(cl.size), c_right)

