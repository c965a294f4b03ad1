import sys, faulthandler; faulthandler.enable(); sys.path.insert(0,'.')
import numpy as np
from pmpc_amd import backend
from tests.support.problems import abi_args, rand_problem
from tests.test_cone_gpu import CONE_CASES
for case in CONE_CASES[:3]:
    M, N, x, u, Nc, bu, bx, sl, sl0, kink = case
    rng = np.random.default_rng(1000 + M + 7 * N + x)
    args, kw = rand_problem(rng, M, N, x, u, bu, bx, sl, sl0)
    print("case", case, flush=True)
    X, U = backend.lcone_solve(*abi_args(args, kw, Nc), smooth_alpha=float("nan"), solver="ecos", verbose=2)
    print("ok", np.isfinite(X).all(), flush=True)
