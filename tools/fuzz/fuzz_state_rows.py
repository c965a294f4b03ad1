"""Random `extra_cstrs` rows on the state and control of one stage (obstacle half-spaces, mixed rows; PMPC.jl/src/main.jl:293-316) through
`pmpc_amd.backend.aff_solve` against the oracle's joint QP with the rows as rows: random dims, consensus horizons, control and state
boxes next to the rows, both pairings of state and control, several rows per stage.   usage: fuzz_state_rows.py SEED CASES"""
import signal
import sys

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, ".")
from oracle import lqp_oracle as orc
from pmpc_amd import backend
from tests.support.problems import rand_problem

seed, cases = int(sys.argv[1]), int(sys.argv[2])


class OracleTimeout(Exception):
    pass


def _alarm(*_):
    raise OracleTimeout()


signal.signal(signal.SIGALRM, _alarm)
rng = np.random.default_rng(seed)
worst, fails, skipped = 0.0, 0, 0
for case in range(cases):
    M, N = int(rng.integers(1, 6)), int(rng.integers(3, 10))
    x, u = [(4, 2), (3, 2), (6, 3), (5, 2), (4, 3), (6, 2)][int(rng.integers(0, 6))]
    Nc = int(rng.choice([0, 1, 2, -1]))
    Ncc = N if Nc < 0 else min(Nc, N)
    bu = float(rng.choice([0.6, 1.0, 3.0])) if rng.random() < 0.8 else None
    bx = 4.0 if rng.random() < 0.3 else None
    args, kw = rand_problem(rng, M, N, x, u, bu, bx)
    x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
    try:
        X0, U0 = orc.lqp_solve_py(*args, Nc=Nc, **kw)
    except Exception:
        skipped += 1
        continue
    ncu = Ncc * u + M * (N - Ncc) * u
    n = ncu + M * N * x
    xcol = lambda i, j, r: ncu + (i * N + j) * x + r
    ucol = lambda i, j, r: j * u + r if j < Ncc else Ncc * u + (i * (N - Ncc) + (j - Ncc)) * u + r
    # (at most one row per (particle, stage) pair apart from one deliberate double, few enough rows to stay feasible next to the boxes)
    pairs = [(i, t) for i in range(M) for t in range(N)]
    rng.shuffle(pairs)
    pairs = pairs[:int(rng.integers(1, M + 2))]
    if rng.random() < 0.3:
        pairs.append(pairs[0])
    nrows = len(pairs)
    G, h = np.zeros((nrows, n)), np.zeros(nrows)
    for k in range(nrows):
        i, t = pairs[k]
        form = int(rng.integers(0, 3))  # 0: (X[t], U[t]); 1: (X[t-1], U[t]); 2: state alone
        if form == 1 and t == 0:
            form = 0
        a, b = rng.standard_normal(x), rng.standard_normal(u)
        if form == 2:
            b[:] = 0.0
        jx = t - 1 if form == 1 else t
        for r in range(x):
            G[k, xcol(i, jx, r)] = a[r]
        for r in range(u):
            if b[r] != 0.0:
                G[k, ucol(i, t, r)] = b[r]
        h[k] = a @ X0[i, jx] + b @ U0[i, t] - 0.15 * rng.random()
    Gs = sp.csr_matrix(G)
    tup = (nrows, [], 0, Gs, sp.csr_matrix((nrows, 0)), h, np.zeros(n), np.zeros(0))
    try:  # (the rows can make the problem infeasible or degenerate next to tight boxes — the oracle then asserts or stalls: not a case)
        signal.alarm(20)
        Xo, Uo = orc.lqp_solve_py(*args, Nc=Nc, rows=(Gs, h), **kw)
        signal.alarm(0)
    except BaseException:
        signal.alarm(0)
        skipped += 1
        continue
    X, U, _ = backend.aff_solve(f, fx, fu, x0, X_prev, U_prev, Q, R, X_ref, U_ref, kw["reg_x"], kw["reg_u"], None, None, kw.get("x_l"), kw.get("x_u"), kw.get("u_l"),
                                kw.get("u_u"), solver_settings=dict(solver="osqp", Nc=Nc, extra_cstrs=[tup]))
    if np.isnan(U).any():
        fails += 1
        print(f"case {case}: M{M} N{N} x{x} u{u} Nc{Nc} bu{bu} bx{bx} rows{nrows}: solver failed", flush=True)
        continue
    rel = lambda a_, b_: np.linalg.norm(a_ - b_) / max(np.linalg.norm(b_), 1.0)
    e = max(rel(X[:, 1:], Xo), rel(U, Uo))
    worst = max(worst, e)
    if e > 1e-6:
        fails += 1
        print(f"case {case}: M{M} N{N} x{x} u{u} Nc{Nc} bu{bu} bx{bx} rows{nrows}: rel err {e:.3e}", flush=True)
print(f"{cases} cases ({skipped} skipped), {fails} failures, worst rel err {worst:.3e}")
