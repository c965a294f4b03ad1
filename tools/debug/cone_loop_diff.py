"""Library SCP loop with the cone objective vs a Python loop of lcone_solve calls: per-iteration differences (debugging aid)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from pmpc_amd import dynamics as dyn
from pmpc_amd.device import MODEL_UNICYCLE, DeviceSolver, to_device_problem

M, N, Nc = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (40, 12, 3)))
verbose = int(sys.argv[4]) if len(sys.argv) > 4 else 0
prob = dyn.make_unicycle_problem(M=M, N=N, Nc=Nc)
d = to_device_problem(prob)
common = dict(Q=d["Q"], R=d["R"], X_ref=d["X_ref"], U_ref=d["U_ref"], reg_x=prob["reg_x"], reg_u=prob["reg_u"], Nc=Nc, x0=d["x0"], lu=d["lu"], uu=d["uu"], symmetric_cost=True)
steps = 4
hist = {}
for mode in ("python", "library"):
    for k in range(1, steps + 1):
        solver = DeviceSolver(0)
        Xa, Ua = d["X_prev"].clone(), d["U_prev"].clone()
        Xb, Ub = torch.empty_like(Xa), torch.empty_like(Ua)
        if mode == "python":
            for it in range(k):
                f, fx, fu = solver.linearize(MODEL_UNICYCLE, d["x0"], Xa, Ua, d["params"])
                _, _, st = solver.lcone_solve(f=f, fx=fx, fu=fu, X_prev=Xa, U_prev=Ua, X_out=Xb, U_out=Ub, static_cons_bounds=True, prev_is_last_solution=it > 0,
                                              cold_start=it == 0, verbose=verbose if it == k - 1 else 0, **common)
                info = dict(solver.last_info)
                Xa, Xb, Ua, Ub = Xb, Xa, Ub, Ua
            hist[(mode, k)] = (Xa.clone(), Ua.clone(), info)
        else:
            x, u = Xa.shape[-1], Ua.shape[-1]
            mk = lambda *shape: torch.empty(shape, dtype=torch.float64, device="cuda")
            bufs = [mk(M, N, x), mk(M, N, x, x), mk(M, N, u, x), mk(M, N, x), mk(M, N, x, x), mk(M, N, u, x)]
            res, infos, last_in_out, done = solver.scp_loop(MODEL_UNICYCLE, d["params"], k, f=bufs[0], fx=bufs[1], fu=bufs[2], f2=bufs[3], fx2=bufs[4], fu2=bufs[5],
                                                            X_prev=Xa, U_prev=Ua, X_out=Xb, U_out=Ub, first_cold=True, cone_objective=True, **common)
            solver.sync()
            X_lib, U_lib = (Xb, Ub) if last_in_out else (Xa, Ua)
            hist[(mode, k)] = (X_lib.clone(), U_lib.clone(), infos[-1])
        solver.close()
for k in range(1, steps + 1):
    a, b = hist[("python", k)], hist[("library", k)]
    dU = (a[1] - b[1]).abs()
    i = int(dU.reshape(M, -1).max(1).values.argmax())
    print(k, "dX", float((a[0] - b[0]).abs().max()), "dU", float(dU.max()), "worst particle", i, {kk: a[2][kk] for kk in ("outer_solves", "active_set_rounds", "ipm_iters")},
          {kk: b[2][kk] for kk in ("outer_solves", "active_set_rounds", "ipm_iters")})
