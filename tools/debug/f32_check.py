"""fp32-storage mode: an SCP loop with float32 fx, fu, Q, R against the same loop in fp64 (quadrotor, boxes and thrust cones)."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from pmpc_amd import dynamics as dyn
from pmpc_amd.device import MODEL_QUADROTOR, DeviceSolver, to_device_problem
M = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
prob = dyn.make_quadrotor_problem(M=M, N=N, Nc=1)
d = to_device_problem(prob)
Wc = torch.zeros((2, 4), dtype=torch.float64, device="cuda"); Wc[0, 1] = Wc[1, 2] = 1.0
soc_kw = dict(soc_W=Wc, soc_w0=torch.zeros(2, dtype=torch.float64, device="cuda"), soc_v=torch.tensor([0.3, 0, 0, 0.0], dtype=torch.float64, device="cuda"),
              soc_v0=0.0, soc_u_interior=torch.tensor([9.81, 0, 0, 0.0], dtype=torch.float64, device="cuda"))
for soc in (False, True):
    res = {}
    for dt in (torch.float64, torch.float32):
        s = DeviceSolver(0)
        Xa, Ua = d["X_prev"].clone(), d["U_prev"].clone()
        Xb, Ub = torch.empty_like(Xa), torch.empty_like(Ua)
        Q, R = d["Q"].to(dt).contiguous(), d["R"].to(dt).contiguous()
        fx = torch.empty((M, N, 12, 12), dtype=dt, device="cuda"); fu = torch.empty((M, N, 4, 12), dtype=dt, device="cuda")
        f = torch.empty((M, N, 12), dtype=torch.float64, device="cuda")
        log = []
        for it in range(steps):
            s.linearize(MODEL_QUADROTOR, d["x0"], Xa, Ua, d["params"], f, fx, fu)
            fn = s.lsoc_solve if soc else s.lqp_solve
            X, U, status = fn(**(soc_kw if soc else {}), f=f, fx=fx, fu=fu, X_prev=Xa, U_prev=Ua, Q=Q, R=R, X_ref=d["X_ref"], U_ref=d["U_ref"],
                              reg_x=prob["reg_x"], reg_u=prob["reg_u"], Nc=1, x0=d["x0"], lu=d["lu"], uu=d["uu"], X_out=Xb, U_out=Ub,
                              symmetric_cost=True, static_cons_bounds=True, prev_is_last_solution=it > 0)
            s.sync()
            assert status == 0, status
            log.append((s.last_info["active_set_rounds"], s.last_info["ipm_iters"]))
            Xa, Xb, Ua, Ub = Xb, Xa, Ub, Ua
        res[dt] = (Xa.cpu().numpy().copy(), Ua.cpu().numpy().copy(), log)
        s.close()
    X64, U64, l64 = res[torch.float64]
    X32, U32, l32 = res[torch.float32]
    print("soc" if soc else "box", "rel diff X", np.linalg.norm(X32 - X64) / np.linalg.norm(X64), "U", np.linalg.norm(U32 - U64) / np.linalg.norm(U64))
    print("   rounds f64", l64)
    print("   rounds f32", l32, flush=True)
