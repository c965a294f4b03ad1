"""One state-boxed problem of tests/test_xbox_gpu.py, verbose: usage xbox_one.py M N x u Nc bu pull margin seed"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from oracle import lqp_oracle as orc
from pmpc_amd import backend
from tests.support.problems import abi_args, xbox_problem
M, N, x, u, Nc = [int(v) for v in sys.argv[1:6]]
bu = None if sys.argv[6] == "None" else float(sys.argv[6])
pull, margin, seed = float(sys.argv[7]), float(sys.argv[8]), int(sys.argv[9])
args, kw = xbox_problem(np.random.default_rng(seed), orc, M, N, x, u, Nc, bu, pull=pull, margin=margin)
Xo, Uo = orc.lqp_solve_py(*args, Nc=Nc, **kw)
print("binding", int(np.sum((Xo <= kw["x_l"] + 1e-9) | (Xo >= kw["x_u"] - 1e-9))), "of", Xo.size, flush=True)
for rep in range(2):
    X, U = backend.lqp_solve(*abi_args(args, kw, Nc), verbose=True)
    print("err", np.linalg.norm(X - Xo) / np.linalg.norm(Xo), np.linalg.norm(U - Uo) / max(np.linalg.norm(Uo), 1.0), flush=True)
if os.environ.get("NEXT"):
    x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
    rng = np.random.default_rng(12)
    args2 = (x0, f + float(os.environ["NEXT"]) * rng.standard_normal(f.shape), fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref)
    Xo2, Uo2 = orc.lqp_solve_py(*args2, Nc=Nc, **kw)
    print("next problem: binding", int(np.sum((Xo2 <= kw["x_l"] + 1e-9) | (Xo2 >= kw["x_u"] - 1e-9))), flush=True)
    X, U = backend.lqp_solve(*abi_args(args2, kw, Nc), verbose=True)
    print("err", np.linalg.norm(X - Xo2) / np.linalg.norm(Xo2), np.linalg.norm(U - Uo2) / max(np.linalg.norm(Uo2), 1.0), flush=True)
