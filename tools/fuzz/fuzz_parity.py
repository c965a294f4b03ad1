"""One-off stress: random problems through c_lqp_solve / c_lcone_solve vs the oracle (run on a GPU box)."""
import faulthandler, sys, time, numpy as np
faulthandler.enable()
sys.path.insert(0, ".")
from oracle import lqp_oracle as orc
from pmpc_amd import backend
from tests.support.problems import abi_args, rand_problem
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dims = [(12, 4), (12, 3), (12, 2), (10, 4), (10, 2), (9, 4), (9, 3), (8, 4), (8, 2), (7, 3), (6, 4), (6, 3), (6, 2), (5, 3), (5, 2), (4, 4),
        (4, 3), (4, 2), (4, 1), (3, 3), (3, 2), (3, 1), (2, 2), (2, 1), (1, 1), (9, 5), (13, 2), (11, 4)]
worst, fails = 0.0, []
t0 = time.time()
for k in range(n):
    x, u = dims[rng.integers(len(dims))]
    M, N = int(rng.integers(1, 9)), int(rng.integers(1, 12))
    Nc = int(rng.choice([0, min(1, N), min(2, N), -1, N]))
    bu = None if rng.random() < 0.2 else float(rng.choice([0.1, 0.3, 1.0]))
    bx = None if rng.random() < 0.6 else float(rng.choice([8.0, 20.0]))
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
        import os
        os.makedirs("gpurun_out", exist_ok=True)
        np.savez(f"gpurun_out/fuzz_fail_{len(fails)}.npz", desc=np.array(str(desc)), Nc=Nc, cone=cone, X=X, U=U, Xo=Xo, Uo=Uo,
                 **{f"a{i}": a for i, a in enumerate(args)}, **{f"k_{k2}": np.asarray(v2) for k2, v2 in kw.items()})
    worst = max(worst, err if np.isfinite(err) else np.inf)
print(f"{n} cases, {len(fails)} failures, worst rel err {worst:.2e}, {time.time() - t0:.1f}s")
