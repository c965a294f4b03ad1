"""State boxes inside the active-set rounds: the state-boxed cases of tests/support/problems.py (+ tighter ones) against the oracle,
cold and warm, with the solver's own account of what it did (rounds / interior-point iterations)."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from oracle import lqp_oracle as orc
from pmpc_amd import backend, _lib
from tests.support.problems import abi_args, rand_problem, xbox_problem

CASES = [(6, 10, 4, 2, 1, None, 2.5), (6, 10, 4, 2, 5, None, 2.5)]
verbose = int(os.environ.get("V", "0"))
worst = 0.0
rng0 = np.random.default_rng(99)
DIMS = [(12, 4), (6, 3), (4, 2), (3, 1), (6, 2), (5, 3), (2, 1), (9, 3), (8, 4)]
for _ in range(int(os.environ.get("NRAND", "24"))):
    x, u = DIMS[rng0.integers(len(DIMS))]
    CASES.append((int(rng0.integers(1, 9)), int(rng0.integers(4, 14)), x, u, int(rng0.choice([0, 1, 1, 2, -1])),
                  [None, 0.3, 0.5][rng0.integers(3)], -float(rng0.uniform(0.3, 0.8))))
only = os.environ.get("ONLY")
for k, (M, N, x, u, Nc, bu, bx) in enumerate(CASES):
    if only is not None and k != int(only):
        continue
    if bx < 0:  # feasible by construction, binding
        args, kw = xbox_problem(np.random.default_rng(4200 + k), orc, M, N, x, u, Nc, bu, pull=-bx)
    else:
        args, kw = rand_problem(np.random.default_rng(4200 + k), M, N, x, u, bu, bx)
    try:
        Xo, Uo = orc.lqp_solve_py(*args, Nc=Nc, **kw)
    except AssertionError:
        print((M, N, x, u, Nc, bu, bx), "oracle: infeasible / no certificate, skipped", flush=True)
        continue
    nact = int(np.sum((Xo <= kw["x_l"] + 1e-7) | (Xo >= kw["x_u"] - 1e-7)))
    for rep in range(2):
        X, U = backend.lqp_solve(*abi_args(args, kw, Nc), verbose=bool(verbose))
        info = ""
        e = max(np.linalg.norm(X - Xo) / np.linalg.norm(Xo), np.linalg.norm(U - Uo) / max(np.linalg.norm(Uo), 1.0))
        worst = max(worst, e)
        print((M, N, x, u, Nc, bu, bx), "active state rows", nact, "rep", rep, "err %.2e" % e, info, flush=True)
print("worst", worst)
