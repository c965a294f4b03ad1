import sys, numpy as np, torch
sys.path.insert(0, ".")
from pmpc_amd import dynamics as dyn
from pmpc_amd.device import MODEL_QUADROTOR, DeviceSolver, to_device_problem
M, N = 4096, 100
prob = dyn.make_quadrotor_problem(M=M, N=N, Nc=1)
d = to_device_problem(prob, "cuda")
s = DeviceSolver(0)
Xa, Ua = d["X_prev"].clone(), d["U_prev"].clone()
f = torch.empty((M, N, 12), dtype=torch.float64, device="cuda"); fx = torch.empty((M, N, 12, 12), dtype=torch.float64, device="cuda"); fu = torch.empty((M, N, 4, 12), dtype=torch.float64, device="cuda")
for it in range(12):
    s.linearize(MODEL_QUADROTOR, d["x0"], Xa, Ua, d["params"], f, fx, fu)
    X, U, st = s.lqp_solve(f=f, fx=fx, fu=fu, X_prev=Xa, U_prev=Ua, Q=d["Q"], R=d["R"], X_ref=d["X_ref"], U_ref=d["U_ref"], reg_x=prob["reg_x"], reg_u=prob["reg_u"], Nc=1, x0=d["x0"], lu=d["lu"], uu=d["uu"], symmetric_cost=True)
    s.sync()
    c = torch.linalg.vector_norm(U[..., 1:3], dim=-1) - 0.3 * U[..., 0]
    print(it, st, "max cone residual", float(c.max()), "violated stages", int((c > 1e-12).sum()), "of", M * N, "particles", int(((c > 1e-12).sum(1) > 0).sum()), "min T", float(U[..., 0].min()))
    Xa, Ua = X.clone(), U.clone()
