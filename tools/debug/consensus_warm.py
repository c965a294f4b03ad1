#!/usr/bin/env python3
"""Warm-started / no-rollout sub-problems with long consensus horizons against the exact oracle (debug aid)."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from oracle import lqp_oracle as orc  # noqa: E402
from pmpc_amd import dynamics as dyn  # noqa: E402
from pmpc_amd.device import MODEL_QUADROTOR, MODEL_UNICYCLE, DeviceSolver, to_device_problem  # noqa: E402

orc.build()
s = DeviceSolver(0)
rel = lambda a, b: np.linalg.norm(a - b) / max(np.linalg.norm(b), 1.0)
for (model, M, N, Nc) in [("unicycle", 16, 30, -1), ("unicycle", 16, 30, 5), ("quadrotor", 8, 20, -1), ("unicycle", 256, 30, -1)]:
    for variant in ("defect", "warm", "cold", "defect-generic"):
        prob = dyn.make_unicycle_problem(M=M, N=N, Nc=Nc) if model == "unicycle" else dyn.make_quadrotor_problem(M=M, N=N, Nc=Nc)
        mid = MODEL_UNICYCLE if model == "unicycle" else MODEL_QUADROTOR
        d = to_device_problem(prob)
        Xa, Ua = d["X_prev"].clone(), d["U_prev"].clone()
        Xb, Ub = torch.empty_like(Xa), torch.empty_like(Ua)
        errs = []
        for it in range(3):
            f, fx, fu = s.linearize(mid, d["x0"], Xa, Ua, d["params"])
            _, _, st = s.lqp_solve(f=f, fx=fx, fu=fu, X_prev=Xa, U_prev=Ua, Q=d["Q"], R=d["R"], X_ref=d["X_ref"], U_ref=d["U_ref"], reg_x=prob["reg_x"],
                                   reg_u=prob["reg_u"], Nc=Nc, x0=d["x0"], lu=d["lu"], uu=d["uu"], X_out=Xb, U_out=Ub, symmetric_cost=True,
                                   static_cons_bounds=True, prev_is_last_solution=(it > 0 and variant.startswith("defect")), cold_start=(variant == "cold"),
                                   force_generic=variant.endswith("generic"), verbose=(M <= 16 and it == 1 and variant == "defect"))
            s.sync()
            Xo, Uo = orc.lqp_solve_py(prob["x0"], f.cpu().numpy(), fx.cpu().numpy().swapaxes(-1, -2), fu.cpu().numpy().swapaxes(-1, -2), Xa.cpu().numpy(),
                                      Ua.cpu().numpy(), prob["Q"], prob["R"], prob["X_ref"], prob["U_ref"], reg_x=prob["reg_x"], reg_u=prob["reg_u"], Nc=Nc,
                                      u_l=prob["u_l"], u_u=prob["u_u"])
            errs.append((st, s.last_info["active_set_rounds"], s.last_info["ipm_iters"], f"{rel(Xb.cpu().numpy(), Xo):.1e}", f"{rel(Ub.cpu().numpy(), Uo):.1e}"))
            Xa, Xb, Ua, Ub = Xb, Xa, Ub, Ua
        print(model, M, N, Nc, variant, errs, flush=True)
