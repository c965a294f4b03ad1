"""One-off stress: random problems through c_lqp_solve / c_lcone_solve vs the oracle (run on a GPU box)."""
import faulthandler, sys, time, numpy as np
faulthandler.enable()
sys.path.insert(0, ".")
from oracle import lqp_oracle as orc
from pmpc_amd import backend
from tests.support.problems import abi_args, rand_problem
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dims = [(12, 4), (4, 2), (2, 1), (3, 2), (5, 3), (6, 2), (8, 4), (7, 3), (3, 1), (9, 5)]
worst, fails = 0.0, []
t0 = time.time()
for k in range(n):
    x, u = dims[rng.integers(len(dims))]
    M, N = int(rng.integers(1, 9)), int(rng.integers(1, 12))
    Nc = int(rng.choice([0, min(1, N), min(2, N), -1, N]))
    bu = None if rng.random() < 0.2 else float(rng.choice([0.1, 0.3, 1.0]))
    bx = None if rng.random() < 0.6 else float(rng.choice([3.0, 8.0]))
    sl = None if rng.random() < 0.7 else 0.5
    sl0 = None if (sl is None or rng.random() < 0.5) else 0.3
    cone = rng.random() < 0.3 and sl is None
    args, kw = rand_problem(rng, M, N, x, u, bu, bx, sl, sl0)
    desc = (M, N, x, u, Nc, bu, bx, sl, sl0, cone)
    if '-v' in sys.argv: print(k, desc, flush=True)
    try:
        if cone:
            Xo, Uo = orc.lcone_solve_py(*args, Nc=Nc, **kw)
            X, U = backend.lcone_solve(*abi_args(args, kw, Nc), smooth_alpha=float("nan"), solver="ecos")
        else:
            Xo, Uo = orc.lqp_solve_py(*args, Nc=Nc, **kw)
            X, U = backend.lqp_solve(*abi_args(args, kw, Nc))
    except Exception as e:  # oracle could not solve (infeasible random boxes): skip
        print("skip", desc, type(e).__name__)
        continue
    err = max(np.linalg.norm(X - Xo) / max(np.linalg.norm(Xo), 1e-300), np.linalg.norm(U - Uo) / max(np.linalg.norm(Uo), 1.0))
    if not np.isfinite(err) or err > 1e-7:
        fails.append((desc, err))
        print("FAIL", desc, err)
    worst = max(worst, err if np.isfinite(err) else np.inf)
print(f"{n} cases, {len(fails)} failures, worst rel err {worst:.2e}, {time.time() - t0:.1f}s")
