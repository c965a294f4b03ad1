"""One small stage-cone problem through the cone rounds with the debug dump (verbose = 2)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from oracle import lqp_oracle as orc
from pmpc_amd.device import DeviceSolver
from tests.support.problems import rand_problem
dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda")
T = lambda a: dev(np.swapaxes(a, -1, -2))
s = DeviceSolver(0)
M, N, x, u, Nc, q = [int(v) for v in sys.argv[1:7]] if len(sys.argv) > 6 else (2, 3, 12, 4, 0, 1)
bu = float(sys.argv[7]) if len(sys.argv) > 7 and sys.argv[7] != 'None' else (1.0 if len(sys.argv) <= 7 else None)
rng = np.random.default_rng(int(sys.argv[8]) if len(sys.argv) > 8 else 11)
args, kw = rand_problem(rng, M, N, x, u, bu)
W = np.zeros((q, u)); W[np.arange(q), np.arange(1, q + 1)] = 1.2
w0 = 0.02 * rng.standard_normal(q)
v = np.zeros(u); v[0] = 0.5
v0 = 0.1
u_int = np.zeros(u); u_int[0] = 0.15
Xo, Uo = orc.lsoc_solve_py(*args, Nc=Nc, reg_x=kw["reg_x"], reg_u=kw["reg_u"], u_l=kw.get("u_l"), u_u=kw.get("u_u"), soc_W=W, soc_w0=w0, soc_v=v, soc_v0=v0, u_interior=u_int)
print("oracle U", Uo.reshape(-1, u))
print("oracle s", (Uo @ np.vstack([v, W]).T + np.concatenate([[v0], w0])).reshape(-1, q + 1))
x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
X, U, status = s.lsoc_solve(f=dev(f), fx=T(fx), fu=T(fu), X_prev=dev(X_prev), U_prev=dev(U_prev), Q=T(Q), R=T(R), X_ref=dev(X_ref),
                            U_ref=dev(U_ref), reg_x=kw["reg_x"], reg_u=kw["reg_u"], Nc=Nc, symmetric_cost=True, soc_W=dev(W),
                            soc_w0=dev(w0), soc_v=dev(v), soc_v0=v0, soc_u_interior=dev(u_int), verbose=2, **(dict(lu=dev(kw["u_l"]), uu=dev(kw["u_u"])) if bu is not None else {}))
s.sync()
print("status", status, s.last_info)
Xn, Un = X.cpu().numpy(), U.cpu().numpy()
print("err", np.linalg.norm(Xn - Xo) / np.linalg.norm(Xo), np.linalg.norm(Un - Uo) / max(np.linalg.norm(Uo), 1.0))
print("device U", Un.reshape(-1, u)[:8])
