import numpy as np, sys
sys.path.insert(0,'/root/repo')
from oracle import lqp_oracle
from pmpc_amd import backend
from tests.support.problems import abi_args, rand_problem
rng = np.random.default_rng(0)
for (M, N, x, u, Nc, bu) in [(8, 10, 12, 4, 1, 0.4), (4, 6, 4, 2, -1, None)]:
    args, kw = rand_problem(rng, M, N, x, u, bu)
args, kw = rand_problem(rng, 40, 6, 4, 2, 0.4)
for alpha in (float("nan"), 1e2):
    X, U = backend.lcone_solve(*abi_args(args, kw, 1), smooth_alpha=alpha, solver="ecos", verbose=True)
    print("alpha", alpha, "nan" if np.isnan(U).any() else "ok", flush=True)
