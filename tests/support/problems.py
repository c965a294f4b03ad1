"""Seeded problem generators shared by the CPU and GPU tests (py layout, see oracle/lqp_oracle.py)."""
from __future__ import annotations

import numpy as np


def rand_problem(rng, M, N, x, u, bounds_u=None, bounds_x=None, slew=None, slew0=None):
    """Dense random SPD problem in the style of PMPC.jl/test/runtests.jl:6-27 (random fx, fu, Q'Q, R'R,
    box bounds), made contractive so that box-feasible trajectories exist."""
    fx = 0.7 * np.eye(x) + 0.15 * rng.standard_normal((M, N, x, x))
    fu = rng.standard_normal((M, N, x, u))
    f = 0.5 * rng.standard_normal((M, N, x))
    X_prev, U_prev = rng.standard_normal((M, N, x)), 0.1 * rng.standard_normal((M, N, u))
    X_ref, U_ref = rng.standard_normal((M, N, x)), rng.standard_normal((M, N, u))
    A = rng.standard_normal((M, N, x, x))
    Q = A @ np.swapaxes(A, -1, -2) / x + 0.1 * np.eye(x)
    B = rng.standard_normal((M, N, u, u))
    R = B @ np.swapaxes(B, -1, -2) / u + 0.1 * np.eye(u)
    x0 = rng.standard_normal((M, x))
    kw = dict(reg_x=1.0, reg_u=0.1)
    if bounds_u is not None:
        kw["u_l"], kw["u_u"] = -bounds_u * np.ones((M, N, u)), bounds_u * np.ones((M, N, u))
    if bounds_x is not None:
        kw["x_l"], kw["x_u"] = -bounds_x * np.ones((M, N, x)), bounds_x * np.ones((M, N, x))
    if slew is not None:
        kw["slew_reg"] = slew * (1 + rng.random(M))
    if slew0 is not None:
        kw["slew_reg0"] = slew0 * (1 + rng.random(M))
        kw["slew_um1"] = rng.standard_normal((M, u))
    return (x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref), kw


def xbox_problem(rng, oracle, M, N, x, u, Nc, bounds_u=None, slew=None, slew0=None, pull=0.5, margin=0.02):
    """A problem whose state boxes BIND and are feasible by construction: X_f is a dynamics-, box- and consensus-consistent
    trajectory (the oracle's optimum for another reference), X* the optimum without state boxes; the boxes are the entrywise
    interval hull of X_f and the point `pull` of the way from X_f to X*, widened by `margin` — so X_f is strictly inside and X* is
    outside wherever it differs from X_f by more than margin / (1 - pull)."""
    args, kw = rand_problem(rng, M, N, x, u, bounds_u, None, slew, slew0)
    Xs, _ = oracle.lqp_solve_py(*args, Nc=Nc, **kw)
    other = list(args)
    other[8] = rng.standard_normal((M, N, x))
    Xf, _ = oracle.lqp_solve_py(*other, Nc=Nc, **kw)
    P = Xf + pull * (Xs - Xf)
    kw["x_l"], kw["x_u"] = np.minimum(Xf, P) - margin, np.maximum(Xf, P) + margin
    return args, kw


# (M, N, x, u, Nc, u-bound, x-bound, slew, slew0) — covers every structural branch of lqp_utils.jl
CASES = [
    (1, 5, 2, 1, 0, None, None, None, None),
    (3, 6, 3, 2, 0, None, None, None, None),
    (3, 6, 3, 2, 1, None, None, None, None),
    (3, 6, 3, 2, 3, None, None, None, None),
    (3, 6, 3, 2, -1, None, None, None, None),
    (3, 6, 3, 2, 2, None, None, 0.5, None),
    (3, 6, 3, 2, 0, None, None, 0.5, 0.3),
    (3, 6, 3, 2, 2, None, None, 0.5, 0.3),
    (3, 6, 3, 2, -1, None, None, 0.5, 0.3),
    (1, 1, 3, 2, -1, None, None, 0.5, 0.3),
    (2, 1, 3, 2, 0, None, None, 0.5, 0.3),
    (3, 6, 3, 2, 1, 0.3, None, None, None),
    (3, 6, 3, 2, 0, 0.3, None, None, None),
    (3, 6, 3, 2, 2, 0.3, 5.0, 0.5, 0.3),
    (4, 8, 4, 2, -1, 0.2, 30.0, None, None),
    (8, 11, 4, 2, 1, 0.2, 100.0, None, None),
    (70, 7, 5, 3, 1, 0.5, None, None, None),
    (5, 9, 12, 4, 1, 0.4, None, None, None),
    (5, 9, 12, 4, 0, None, None, None, None),
    # edge cases: single stage / single particle, state bounds only, padded fast-path dims, odd particle counts
    (1, 1, 4, 2, 0, 0.3, None, None, None),
    (3, 1, 4, 2, 1, 0.3, None, None, None),
    (1, 2, 2, 1, 0, 0.5, None, None, None),
    (6, 10, 4, 2, 1, None, 2.5, None, None),
    (6, 10, 12, 4, 0, None, 4.0, None, None),
    (7, 12, 5, 3, 1, 0.4, 6.0, None, None),
    (301, 5, 2, 1, 1, 0.5, None, None, None),
    (9, 6, 6, 2, 0, 0.3, 8.0, None, None),
    (4, 7, 8, 4, 1, 0.5, None, None, None),
    (3, 5, 4, 2, 1, None, None, None, 0.4),
    # consensus horizons > 1 on the register-resident path (condensing kernel): one tile, tiles that straddle
    # stages (udim 3), several tiles / several waves per particle, full consensus with both boxes
    (5, 9, 12, 4, 4, 0.4, None, None, None),
    (5, 9, 12, 4, -1, 0.4, 6.0, None, None),
    (6, 10, 4, 2, 5, None, 2.5, None, None),
    (3, 20, 2, 1, -1, 0.5, None, None, None),
    (4, 12, 5, 3, 7, 0.4, None, None, None),
    (2, 40, 12, 4, -1, 0.4, None, None, None),
    (3, 24, 8, 4, 20, None, None, None, None),
]


def abi_args(args, kw, Nc):
    """py-layout problem -> positional argument tuple of backend.lqp_solve (Julia shapes + sentinels)."""
    from pmpc_amd.backend import py2jl

    x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
    M, N, x = f.shape
    u = fu.shape[-1]
    nanx, nanu = np.full((M, N, x), np.nan), np.full((M, N, u), np.nan)
    lx, ux = kw.get("x_l", nanx), kw.get("x_u", nanx)
    lu, uu = kw.get("u_l", nanu), kw.get("u_u", nanu)
    sr = np.broadcast_to(kw.get("slew_reg", np.nan), (M,)).astype(float)
    sr0 = np.broadcast_to(kw.get("slew_reg0", np.nan), (M,)).astype(float)
    um1 = np.broadcast_to(kw.get("slew_um1", np.nan), (M, u)).astype(float)
    j1 = lambda a: py2jl(np.asarray(a, float), 1)
    j2 = lambda a: py2jl(np.asarray(a, float), 2)
    return (Nc, j1(x0), j1(f), j2(fx), j2(fu), j1(X_prev), j1(U_prev), j2(Q), j2(R), j1(X_ref), j1(U_ref), j1(lx), j1(ux),
            j1(lu), j1(uu), kw["reg_x"], kw["reg_u"], sr, sr0, j1(um1))
