import sys
import numpy as np, torch
sys.path.insert(0, ".")
from oracle import lqp_oracle as orc
from pmpc_amd.device import DeviceSolver
from tests.support.problems import rand_problem
dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda")
T = lambda a: dev(np.swapaxes(a, -1, -2))
s = DeviceSolver(0)
for Nc in (0, 1, 2, -1):
    M, N, x, u = 4, 6, 5, 3
    args, kw = rand_problem(np.random.default_rng(5), M, N, x, u, 2.0)
    W = np.zeros((2, 3)); W[0, 1] = W[1, 2] = 1.0
    v, v0, w0, u_int = np.array([0.5, 0, 0]), 0.05, np.zeros(2), np.array([0.2, 0, 0])
    Xo, Uo = orc.lsoc_solve_py(*args, Nc=Nc, reg_x=kw["reg_x"], reg_u=kw["reg_u"], u_l=kw["u_l"], u_u=kw["u_u"], soc_W=W, soc_w0=w0, soc_v=v, soc_v0=v0, u_interior=u_int)
    x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
    common = dict(f=dev(f), fx=T(fx), fu=T(fu), X_prev=dev(X_prev), U_prev=dev(U_prev), Q=T(Q), R=T(R), X_ref=dev(X_ref), U_ref=dev(U_ref),
                  reg_x=kw["reg_x"], reg_u=kw["reg_u"], Nc=Nc, symmetric_cost=True, lu=dev(kw["u_l"]), uu=dev(kw["u_u"]))
    A = np.vstack([v[None], W]); c = np.concatenate([[v0], w0])
    for name, cones in (("shared", dict(sizes=[2], A=dev(A), c=dev(c))), ("tiled", dict(sizes=[2], A=dev(np.tile(A, (M, N, 1, 1))), c=dev(np.tile(c, (M, N, 1)))))):
        X, U, st = s.lsoc_solve(cones=cones, soc_u_interior=dev(u_int), cold_start=True, **common)
        s.sync()
        print("Nc", Nc, name, "status", st, "err", np.linalg.norm(U.cpu().numpy() - Uo) / np.linalg.norm(Uo), "rounds", s.last_info["active_set_rounds"], flush=True)
