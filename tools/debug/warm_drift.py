import sys, numpy as np, torch
sys.path.insert(0, ".")
from pmpc_amd import dynamics as dyn
from pmpc_amd.device import MODEL_QUADROTOR, MODEL_UNICYCLE, DeviceSolver, to_device_problem
model_name, M, N, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
prob = dyn.make_quadrotor_problem(M=M, N=N) if model_name == "quadrotor" else dyn.make_unicycle_problem(M=M, N=N)
model = MODEL_QUADROTOR if model_name == "quadrotor" else MODEL_UNICYCLE
d = to_device_problem(prob, "cuda")
s = DeviceSolver(0)
Xp, Up = d["X_prev"].clone(), d["U_prev"].clone()
its, res = [], []
for k in range(steps):
    f, fx, fu = s.linearize(model, d["x0"], Xp, Up, d["params"])
    X, U, st = s.lqp_solve(f=f, fx=fx, fu=fu, X_prev=Xp, U_prev=Up, Q=d["Q"], R=d["R"], X_ref=d["X_ref"], U_ref=d["U_ref"],
                           reg_x=prob["reg_x"], reg_u=prob["reg_u"], Nc=1, x0=d["x0"], lu=d.get("lu"), uu=d.get("uu"), symmetric_cost=True)
    s.sync()
    assert st == 0
    its.append(s.last_info["ipm_iters"])
    res.append(float(torch.maximum(torch.linalg.vector_norm(X - Xp, dim=-1).max(), torch.linalg.vector_norm(U - Up, dim=-1).max())))
    Xp, Up = X.clone(), U.clone()
print("ipm iters:", its)
print("scp resid:", [round(r, 4) for r in res])
