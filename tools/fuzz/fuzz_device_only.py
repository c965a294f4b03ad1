"""Heap / crash stress of the library alone: random shapes through c_lqp_solve / c_lcone_solve, no oracle."""
import faulthandler, sys, numpy as np
faulthandler.enable()
sys.path.insert(0, ".")
from pmpc_amd import backend
from tests.support.problems import abi_args, rand_problem
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
dims = [(12, 4), (4, 2), (2, 1), (3, 2), (5, 3), (6, 2), (8, 4), (7, 3), (3, 1), (9, 5), (1, 1), (13, 2)]
ok = nan = 0
for k in range(int(sys.argv[2]) if len(sys.argv) > 2 else 50):
    x, u = dims[rng.integers(len(dims))]
    M, N = int(rng.integers(1, 40)), int(rng.integers(1, 14))
    Nc = int(rng.choice([0, min(1, N), min(2, N), -1, N, N + 3]))
    bu = None if rng.random() < 0.2 else float(rng.choice([0.05, 0.3, 1.0]))
    bx = None if rng.random() < 0.6 else float(rng.choice([0.5, 3.0, 8.0]))
    sl = None if rng.random() < 0.7 else 0.5
    sl0 = None if (sl is None or rng.random() < 0.5) else 0.3
    args, kw = rand_problem(rng, M, N, x, u, bu, bx, sl, sl0)
    a = abi_args(args, kw, Nc)
    if rng.random() < 0.3:
        X, U = backend.lcone_solve(*a, smooth_alpha=float(rng.choice([np.nan, 10.0])), solver="ecos")
    else:
        X, U = backend.lqp_solve(*a)
    if np.all(np.isfinite(X)) and np.all(np.isfinite(U)): ok += 1
    else:
        assert np.all(np.isnan(X)) and np.all(np.isnan(U))  # failure convention: everything NaN
        nan += 1
print(f"{ok} solved, {nan} reported infeasible / failed (NaN outputs)")
