import sys, numpy as np, torch
sys.path.insert(0, ".")
from tests.support.problems import rand_problem
from pmpc_amd.device import DeviceSolver
M, N, x, u, Nc, bu = 3, 6, 12, 4, 1, 0.8
args, kw = rand_problem(np.random.default_rng(7000), M, N, x, u, bu)
W = np.zeros((u - 1, u)); W[np.arange(u - 1), np.arange(1, u)] = 1.0
w0, v, v0 = np.zeros(u - 1), np.eye(u)[0] * 0.5, 0.05
u_int = np.eye(u)[0] * 0.2
x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda")
T = lambda a: dev(np.swapaxes(a, -1, -2))
s = DeviceSolver(0)
X, U, status = s.lsoc_solve(f=dev(f), fx=T(fx), fu=T(fu), X_prev=dev(X_prev), U_prev=dev(U_prev), Q=T(Q), R=T(R), X_ref=dev(X_ref),
                            U_ref=dev(U_ref), reg_x=kw["reg_x"], reg_u=kw["reg_u"], Nc=Nc, symmetric_cost=True, soc_W=dev(W),
                            soc_w0=dev(w0), soc_v=dev(v), soc_v0=v0, soc_u_interior=dev(u_int), lu=dev(kw["u_l"]), uu=dev(kw["u_u"]), verbose=2)
s.sync()
print("status", status, s.last_info)
