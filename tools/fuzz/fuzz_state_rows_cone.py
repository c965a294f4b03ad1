"""`extra_cstrs` rows on states inside the reference's DEFAULT solver path: eps-anchored cone objective, hard boxes or log-barrier
smoothing (the rows are smoothed with the boxes, main.jl:298-312), through `backend.aff_solve(solver="ecos")` against the restated
reference cone program with the rows handed to `augment_cone_problem!` (`cone_oracle.lcone_direct_py(extra_cstrs=...)`, slow: small
problems).   usage: fuzz_state_rows_cone.py SEED CASES"""
import signal
import sys

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, ".")
from oracle import cone_oracle as co
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
    M, N = int(rng.integers(2, 7)), int(rng.integers(3, 8))
    x, u = [(4, 2), (3, 2), (6, 3), (5, 2)][int(rng.integers(0, 4))]
    Nc = int(rng.choice([0, 1, 1, 2, -1]))
    bu = float(rng.choice([0.6, 1.0, 3.0]))
    alpha = None if rng.random() < 0.5 else float(rng.choice([1.0, 10.0, 50.0]))
    args, kw = rand_problem(rng, M, N, x, u, bu)
    x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
    Ncc = N if Nc < 0 else min(Nc, N)
    ncu = Ncc * u + M * (N - Ncc) * u
    n = ncu + M * N * x
    X0, U0 = orc.lqp_solve_py(*args, Nc=Nc, **kw)
    pairs = [(i, t) for i in range(M) for t in range(N)]
    rng.shuffle(pairs)
    pairs = pairs[:int(rng.integers(1, M + 1))]
    G, h = np.zeros((len(pairs), n)), np.zeros(len(pairs))
    for k_, (i, t) in enumerate(pairs):
        a = rng.standard_normal(x)
        G[k_, ncu + (i * N + t) * x: ncu + (i * N + t + 1) * x] = a
        h[k_] = a @ X0[i, t] - 0.1 * rng.random()
    tup = (len(pairs), [], 0, sp.csr_matrix(G), sp.csr_matrix((len(pairs), 0)), h, np.zeros(n), np.zeros(0))
    skw = {} if alpha is None else dict(smooth_alpha=alpha)
    try:
        signal.alarm(90)
        Xo, Uo = co.lcone_direct_py(*args, Nc=Nc, extra_cstrs=[tup], **skw, **kw)
        signal.alarm(0)
    except BaseException:
        signal.alarm(0)
        skipped += 1
        continue
    X, U, _ = backend.aff_solve(f, fx, fu, x0, X_prev, U_prev, Q, R, X_ref, U_ref, kw["reg_x"], kw["reg_u"], None, None, None, None, kw["u_l"], kw["u_u"],
                                solver_settings=dict(solver="ecos", Nc=Nc, extra_cstrs=[tup], **skw))
    tag = f"case {case}: M{M} N{N} x{x} u{u} Nc{Nc} bu{bu} alpha {alpha} rows {len(pairs)}"
    if np.isnan(U).any():
        fails += 1
        print(tag + ": solver failed", flush=True)
        continue
    rel = lambda a_, b_: np.linalg.norm(a_ - b_) / max(np.linalg.norm(b_), 1.0)
    e = max(rel(X[:, 1:], Xo), rel(U, Uo))
    worst = max(worst, e)
    if e > 1e-6:
        fails += 1
        print(tag + f": rel err {e:.3e}", flush=True)
print(f"{cases} cases ({skipped} skipped), {fails} failures, worst rel err {worst:.3e}")
