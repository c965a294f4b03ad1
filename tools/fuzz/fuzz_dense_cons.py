"""Random problems whose dense consensus system has 17 .. 256 unknowns (several shared stages, Nc up to N = 64): whole and ragged panels of the
register-resident factorisation (k_cons_solve_reg), both instantiations, held shared controls (1e30 on the diagonal), cold and warm — against
the oracle.   usage: fuzz_dense_cons.py SEED CASES"""
import sys

import numpy as np

sys.path.insert(0, ".")
from oracle import lqp_oracle as orc
from pmpc_amd import backend
from tests.support.problems import abi_args, rand_problem

seed, cases = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
worst, fails, held = 0.0, 0, 0
for case in range(cases):
    x, u = [(4, 2), (3, 2), (5, 3), (2, 1), (12, 4), (6, 3)][int(rng.integers(0, 6))]
    lo = -(-17 // u)
    Nc = int(rng.integers(lo, min(64, 256 // u) + 1))
    N = Nc if rng.random() < 0.5 else int(rng.integers(Nc, min(70, Nc + 8) + 1))
    M = int(rng.integers(2, 5))
    bu = float(rng.choice([0.2, 0.4, 1.0]))
    args, kw = rand_problem(rng, M, N, x, u, bu)
    Xo, Uo = orc.lqp_solve_py(*args, Nc=Nc, **kw)
    held += bool(np.any(np.abs(np.abs(Uo[0, :Nc]) - bu) <= 1e-9))
    rel = lambda a_, b_: np.linalg.norm(a_ - b_) / max(np.linalg.norm(b_), 1.0)
    e = 0.0
    for rep in range(2):
        X, U = backend.lqp_solve(*abi_args(args, kw, Nc))
        e = max(e, rel(X, Xo), rel(U, Uo)) if np.all(np.isfinite(U)) else float("inf")
    worst = max(worst, e)
    if not e <= 1e-7 or not np.all(U[:, :Nc] == U[0:1, :Nc]):
        fails += 1
        print(f"case {case}: M{M} N{N} x{x} u{u} Nc{Nc} (nc {Nc * u}) bu{bu}: rel err {e:.3e}", flush=True)
print(f"{cases} cases ({held} with a shared control on its bound), {fails} failures, worst rel err {worst:.3e}")
