"""An option of the active-set rounds off (0) against on (1) — default as_ckpt, the checkpointed restart of the later rounds' factor
sweeps; OPT=as_sens_min_m: the elementwise update of settled particles from the forward sweep's sensitivity records — : the same SCP
loop twice on one box, outputs compared iteration by iteration.  usage: [OPT=<option>] ckpt_check.py [M] [N] [steps] [soc|vmax=<v>]
PMPC_DUC_TRACE=1 prints, per round, the histogram of the highest changed stage among the unsettled particles."""
import os
import sys

import numpy as np
import torch

from pmpc_amd import dynamics as dyn
from pmpc_amd.device import MODEL_QUADROTOR, DeviceSolver, to_device_problem

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 50
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
extra = sys.argv[4] if len(sys.argv) > 4 else ""
dev = torch.device("cuda", 0)
prob = dyn.make_quadrotor_problem(M=M, N=N, Nc=1)
d = to_device_problem(prob, dev)
kw = {}
if extra == "soc":
    Wc = torch.zeros((2, 4), dtype=torch.float64, device=dev)
    Wc[0, 1] = Wc[1, 2] = 1.0
    kw = dict(soc_W=Wc, soc_w0=torch.zeros(2, dtype=torch.float64, device=dev), soc_v=torch.tensor([0.3, 0.0, 0.0, 0.0], dtype=torch.float64, device=dev),
              soc_v0=0.0, soc_u_interior=torch.tensor([9.81, 0.0, 0.0, 0.0], dtype=torch.float64, device=dev))
elif extra.startswith("vmax="):
    lx = torch.full((M, N, 12), -float("inf"), dtype=torch.float64, device=dev)
    lx[..., 3:6] = -float(extra[5:])
    kw = dict(lx=lx, ux=-lx)


def loop(ck):
    s = DeviceSolver(0)
    s.set_option(os.environ.get("OPT", "as_ckpt"), ck)
    Xa, Ua = d["X_prev"].clone(), d["U_prev"].clone()
    Xb, Ub = torch.empty_like(Xa), torch.empty_like(Ua)
    f = torch.empty((M, N, 12), dtype=torch.float64, device=dev)
    fx = torch.empty((M, N, 12, 12), dtype=torch.float64, device=dev)
    fu = torch.empty((M, N, 4, 12), dtype=torch.float64, device=dev)
    out = []
    for it in range(steps):
        s.linearize(MODEL_QUADROTOR, d["x0"], Xa, Ua, d["params"], f, fx, fu)
        fn = s.lsoc_solve if extra == "soc" else s.lqp_solve
        _, _, status = fn(**kw, f=f, fx=fx, fu=fu, X_prev=Xa, U_prev=Ua, Q=d["Q"], R=d["R"], X_ref=d["X_ref"], U_ref=d["U_ref"],
                          reg_x=prob["reg_x"], reg_u=prob["reg_u"], Nc=1, x0=d["x0"], lu=d.get("lu"), uu=d.get("uu"), X_out=Xb, U_out=Ub,
                          symmetric_cost=True, static_cons_bounds=True, prev_is_last_solution=it > 0)
        torch.cuda.synchronize()
        out.append((status, dict(s.last_info), Xb.cpu().numpy().copy(), Ub.cpu().numpy().copy()))
        Xa, Xb, Ua, Ub = Xb, Xa, Ub, Ua
    s.close()
    return out


a, b = loop(0), loop(1)
worst = 0.0
for it, (p, q) in enumerate(zip(a, b)):
    ex = np.abs(p[2] - q[2]).max() / max(1.0, np.abs(p[2]).max())
    eu = np.abs(p[3] - q[3]).max() / max(1.0, np.abs(p[3]).max())
    worst = max(worst, ex, eu)
    print(f"it {it}: status {p[0]} / {q[0]}  rounds {p[1]['active_set_rounds']} / {q[1]['active_set_rounds']}  ipm {p[1]['ipm_iters']} / {q[1]['ipm_iters']}"
          f"  max rel diff X {ex:.2e} U {eu:.2e}")
print("worst", worst)
sys.exit(0 if worst < 1e-9 and all(p[0] == 0 and q[0] == 0 for p, q in zip(a, b)) else 1)
