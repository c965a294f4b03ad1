"""Fuzz of the state rows: random dims / consensus horizons / control boxes / slew penalties, state boxes that bind (feasible by
construction), c_lqp_solve-equivalent host entry against the oracle, cold and warm.  usage: fuzz_xbox.py [n] [seed] [cone]
(`cone`: the same problems through the cone path, c_lcone_solve semantics — particle weights meet the state rows)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from oracle import lqp_oracle as orc
from pmpc_amd import backend
from tests.support.problems import abi_args, xbox_problem

n, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0
cone = len(sys.argv) > 3 and sys.argv[3] == "cone"
rng = np.random.default_rng(seed)
DIMS = [(12, 4), (12, 3), (10, 2), (9, 3), (8, 4), (8, 2), (7, 3), (6, 4), (6, 3), (6, 2), (5, 3), (5, 2), (4, 4), (4, 2), (4, 1), (3, 3), (3, 2), (3, 1), (2, 2), (2, 1), (1, 1)]
worst, bad = 0.0, 0
for k in range(n):
    x, u = DIMS[rng.integers(len(DIMS))]
    M, N = int(rng.integers(1, 12)), int(rng.integers(2, 16))
    Nc = int(rng.choice([0, 1, 1, 2, -1]))
    if abs(Nc) > N:
        Nc = 1
    bu = [None, 0.3, 0.6][rng.integers(3)]
    sl = [None, None, 0.5][rng.integers(3)]
    sl0 = 0.3 if (sl is not None and rng.random() < 0.5) else None
    pull = float(rng.choice([0.97, 0.95, 0.9, 0.7]))
    t0 = time.time()
    try:
        args, kw = xbox_problem(np.random.default_rng(seed * 1000 + k), orc, M, N, x, u, Nc, bu, sl, sl0, pull=pull, margin=0.05)
        Xo, Uo = orc.lcone_solve_py(*args, Nc=Nc, **kw) if cone else orc.lqp_solve_py(*args, Nc=Nc, **kw)
    except (AssertionError, RuntimeError):
        print(k, (M, N, x, u, Nc, bu, sl, sl0, pull), "oracle: no certificate, skipped", flush=True)
        continue
    nb = int(np.sum((Xo <= kw["x_l"] + 1e-9) | (Xo >= kw["x_u"] - 1e-9)))
    errs = []
    for rep in range(2):
        X, U = backend.lcone_solve(*abi_args(args, kw, Nc), smooth_alpha=float("nan"), solver="ecos") if cone else backend.lqp_solve(*abi_args(args, kw, Nc))
        errs.append(max(np.linalg.norm(X - Xo) / max(np.linalg.norm(Xo), 1e-300), np.linalg.norm(U - Uo) / max(np.linalg.norm(Uo), 1.0)))
    e = max(errs) if all(np.isfinite(errs)) else np.inf
    worst = max(worst, e)
    flag = "" if e < (1e-6 if cone else 1e-7) else "   <<<<<<"
    bad += e >= (1e-6 if cone else 1e-7)
    print(k, (M, N, x, u, Nc, bu, sl, sl0, pull), "binding", nb, "err %.1e %.1e" % tuple(errs), "%.1fs" % (time.time() - t0), flag, flush=True)
print("worst", worst, "above 1e-7:", bad)
