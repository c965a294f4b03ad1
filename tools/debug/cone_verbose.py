import sys, numpy as np
sys.path.insert(0, ".")
from tests.support.problems import abi_args, rand_problem
from tests.test_cone_gpu import CONE_CASES
from pmpc_amd import backend
idx, alpha = int(sys.argv[1]), float(sys.argv[2])
case = CONE_CASES[idx]
M, N, x, u, Nc = case[:5]
args, kw = rand_problem(np.random.default_rng(4000 + idx), M, N, x, u, *case[5:9])
X, U = backend.lcone_solve(*abi_args(args, kw, Nc), smooth_alpha=alpha, solver="ecos", verbose=2)
