import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from oracle import lqp_oracle as orc
from pmpc_amd import backend
from tests.support.problems import abi_args, xbox_problem
worst = 0
for k, (M, N, x, u, Nc, bu) in enumerate([(4, 1, 4, 2, 0, 0.4), (4, 1, 4, 2, 1, 0.4), (1, 1, 2, 1, 0, None), (3, 2, 12, 4, 1, 0.4), (5, 1, 3, 3, -1, None), (2, 2, 6, 2, 2, 0.5), (300, 3, 2, 1, 1, 0.5)]):
    args, kw = xbox_problem(np.random.default_rng(600 + k), orc, M, N, x, u, Nc, bu, pull=0.8, margin=0.02)
    Xo, Uo = orc.lqp_solve_py(*args, Nc=Nc, **kw)
    nb = int(np.sum((Xo <= kw["x_l"] + 1e-9) | (Xo >= kw["x_u"] - 1e-9)))
    for rep in range(2):
        X, U = backend.lqp_solve(*abi_args(args, kw, Nc))
        e = max(np.linalg.norm(X - Xo) / max(np.linalg.norm(Xo), 1e-300), np.linalg.norm(U - Uo) / max(np.linalg.norm(Uo), 1.0))
        worst = max(worst, e)
        print((M, N, x, u, Nc, bu), "binding", nb, "rep", rep, "err %.1e" % e, flush=True)
print("worst", worst)
