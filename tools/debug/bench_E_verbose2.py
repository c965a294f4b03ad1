"""First cold solve of config E's size with the cone-record dump (verbose = 2)."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from pmpc_amd import dynamics as dyn
from pmpc_amd.device import MODEL_QUADROTOR, DeviceSolver, to_device_problem
M, N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 100
prob = dyn.make_quadrotor_problem(M=M, N=N, Nc=1)
d = to_device_problem(prob)
s = DeviceSolver(0)
Wc = torch.zeros((2, 4), dtype=torch.float64, device="cuda"); Wc[0, 1] = Wc[1, 2] = 1.0
soc_kw = dict(soc_W=Wc, soc_w0=torch.zeros(2, dtype=torch.float64, device="cuda"), soc_v=torch.tensor([0.3, 0, 0, 0.0], dtype=torch.float64, device="cuda"),
              soc_v0=0.0, soc_u_interior=torch.tensor([9.81, 0, 0, 0.0], dtype=torch.float64, device="cuda"))
Xa, Ua = d["X_prev"].clone(), d["U_prev"].clone()
f, fx, fu = s.linearize(MODEL_QUADROTOR, d["x0"], Xa, Ua, d["params"])
X, U, status = s.lsoc_solve(**soc_kw, f=f, fx=fx, fu=fu, X_prev=Xa, U_prev=Ua, Q=d["Q"], R=d["R"], X_ref=d["X_ref"], U_ref=d["U_ref"],
                            reg_x=prob["reg_x"], reg_u=prob["reg_u"], Nc=1, x0=d["x0"], lu=d["lu"], uu=d["uu"], symmetric_cost=True, verbose=2)
s.sync()
print("status", status, s.last_info)
