"""One-off stress of the warm-started active-set iteration: SEQUENCES of related problems of one shape through c_lqp_solve
(each solve starts from the previous solve's accepted active set and solution) vs the oracle (run on a GPU box).
usage: fuzz_warm_as.py [seed] [sequences] [solves per sequence]"""
import faulthandler, sys, time, numpy as np
faulthandler.enable()
sys.path.insert(0, ".")
from oracle import lqp_oracle as orc
from pmpc_amd import backend
from tests.support.problems import abi_args, rand_problem
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
L = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dims = [(12, 4), (12, 2), (10, 4), (8, 4), (6, 3), (5, 2), (4, 2), (4, 1), (3, 3), (2, 1), (9, 5), (13, 2)]
if "--all-dims" in sys.argv:  # every (xdim, udim) pair the MFMA kernels are compiled for (fast_common.h, PMPC_FAST_DIMS)
    dims = [(12, 4), (12, 3), (12, 2), (10, 4), (10, 2), (9, 4), (9, 3), (8, 4), (8, 2), (7, 3), (6, 4), (6, 3), (6, 2), (5, 3), (5, 2),
            (4, 4), (4, 3), (4, 2), (4, 1), (3, 3), (3, 2), (3, 1), (2, 2), (2, 1), (1, 1)]
worst, fails, solves = 0.0, [], 0
t0 = time.time()
for k in range(n):
    x, u = dims[rng.integers(len(dims))]
    M, N = int(rng.integers(1, 12)), int(rng.integers(2, 14))
    Nc = int(rng.choice([0, 1, min(2, N), -1]))
    bu = float(rng.choice([0.05, 0.1, 0.3, 1.0]))
    sl = None if rng.random() < 0.8 else 0.5
    args, kw = rand_problem(rng, M, N, x, u, bu, None, sl, None)
    amp = float(rng.choice([1e-3, 1e-2, 1e-1, 0.5]))
    for t in range(L):
        if t:  # the next sub-problem of an SCP-like sequence: perturbed linearisation, sometimes different boxes
            x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
            f = f + amp * rng.standard_normal(f.shape)
            fx = fx * (1 + amp * rng.standard_normal(fx.shape))
            fu = fu * (1 + amp * rng.standard_normal(fu.shape))
            X_prev = X_prev + amp * rng.standard_normal(X_prev.shape)
            U_prev = U_prev + amp * rng.standard_normal(U_prev.shape)
            args = (x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref)
            if rng.random() < 0.3:
                sc = 1 + 0.3 * rng.standard_normal()
                kw = dict(kw, u_l=kw["u_l"] * abs(sc), u_u=kw["u_u"] * abs(sc))
        desc = (k, t, M, N, x, u, Nc, bu, sl, amp)
        try:
            Xo, Uo = orc.lqp_solve_py(*args, Nc=Nc, **kw)
        except Exception as e:
            print("skip", desc, type(e).__name__)
            break
        X, U = backend.lqp_solve(*abi_args(args, kw, Nc))
        solves += 1
        err = max(np.linalg.norm(X - Xo) / max(np.linalg.norm(Xo), 1e-300), np.linalg.norm(U - Uo) / max(np.linalg.norm(Uo), 1.0))
        if not np.isfinite(err) or err > 1e-7:
            fails.append((desc, err))
            print("FAIL", desc, err, flush=True)
        worst = max(worst, err if np.isfinite(err) else np.inf)
print(f"{n} sequences, {solves} solves, {len(fails)} failures, worst rel err {worst:.2e}, {time.time() - t0:.1f}s")
