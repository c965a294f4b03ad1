"""The library's SCP loop (`pmpc_scp_loop_device`: linearise -> sub-problem -> residual -> swap inside the library, follow-up work
enqueued behind the rounds before their outcome is known) against a Python-driven loop of the same calls: random model, particle
count, horizon, consensus horizon, QP or cone objective (hard boxes / log-barrier smoothing), velocity limits on the quadrotor
(state boxes).  QP objective: the same iterates bit for bit; cone objective: 1e-6 (a point consistent to the acceptance test's
1e-9 is accepted on either path).   usage: fuzz_scp_loop.py SEED CASES"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from pmpc_amd import dynamics as dyn
from pmpc_amd.device import MODEL_QUADROTOR, MODEL_UNICYCLE, DeviceSolver, to_device_problem

seed, cases = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
fails, worst = 0, 0.0
for case in range(cases):
    model = str(rng.choice(["quadrotor", "unicycle"]))
    M = int(rng.choice([1, 2, 7, 33, 64, 200, 513]))
    N = int(rng.integers(4, 26))
    Nc = int(rng.choice([0, 1, 1, 2, 5, -1]))
    Nc = min(Nc, N) if Nc >= 0 else Nc
    kind = str(rng.choice(["qp", "qp", "cone", "smooth"])) if M > 1 else "qp"
    steps = int(rng.integers(1, 5)) if model == "quadrotor" else int(rng.integers(1, 3))
    if model == "unicycle" and kind != "qp":
        steps = 1  # (the unicycle's where(u >= 0, eps, -eps) turns the 1e-9 of two equally acceptable cone iterates into 1e-5 one linearisation later)
    vmax = float(rng.choice([0.0, 0.0, 4.0])) if model == "quadrotor" and kind == "qp" else 0.0
    prob = dyn.make_quadrotor_problem(M=M, N=N, Nc=Nc) if model == "quadrotor" else dyn.make_unicycle_problem(M=M, N=N, Nc=Nc)
    mid = MODEL_QUADROTOR if model == "quadrotor" else MODEL_UNICYCLE
    d = to_device_problem(prob)
    common = dict(Q=d["Q"], R=d["R"], X_ref=d["X_ref"], U_ref=d["U_ref"], reg_x=prob["reg_x"], reg_u=prob["reg_u"], Nc=Nc, x0=d["x0"], lu=d["lu"],
                  uu=d["uu"], symmetric_cost=True)
    if vmax > 0.0:
        lx = torch.full_like(d["X_prev"], -float("inf")); ux = torch.full_like(d["X_prev"], float("inf"))
        lx[..., 3:6], ux[..., 3:6] = -vmax, vmax
        common.update(lx=lx, ux=ux)
    skw = dict(smooth_alpha=10.0) if kind == "smooth" else {}
    tag = f"case {case}: {model} M{M} N{N} Nc{Nc} {kind} steps {steps} vmax {vmax}"
    outs = []
    try:
        for lib_loop in (False, True):
            solver = DeviceSolver(0)
            Xa, Ua = d["X_prev"].clone(), d["U_prev"].clone()
            Xb, Ub = torch.empty_like(Xa), torch.empty_like(Ua)
            if not lib_loop:
                res = []
                for it in range(steps):
                    f, fx, fu = solver.linearize(mid, d["x0"], Xa, Ua, d["params"])
                    fn = solver.lqp_solve if kind == "qp" else solver.lcone_solve
                    _, _, st = fn(f=f, fx=fx, fu=fu, X_prev=Xa, U_prev=Ua, X_out=Xb, U_out=Ub, static_cons_bounds=True, prev_is_last_solution=it > 0,
                                  cold_start=it == 0, **(skw if kind != "qp" else {}), **common)
                    assert st == 0, ("python loop", it, solver.last_info)
                    res.append(float(solver.scp_residual(Xb, Xa, Ub, Ua)[0].item()))
                    Xa, Xb, Ua, Ub = Xb, Xa, Ub, Ua
                outs.append((np.array(res), Xa.clone(), Ua.clone()))
            else:
                x, u = Xa.shape[-1], Ua.shape[-1]
                mk = lambda *shape: torch.empty(shape, dtype=torch.float64, device="cuda")
                bufs = [mk(M, N, x), mk(M, N, x, x), mk(M, N, u, x), mk(M, N, x), mk(M, N, x, x), mk(M, N, u, x)]
                lkw = dict(cone_objective=True, **({"barrier_mu": 0.1} if kind == "smooth" else {})) if kind != "qp" else {}
                res, infos, last_in_out, done = solver.scp_loop(mid, d["params"], steps, f=bufs[0], fx=bufs[1], fu=bufs[2], f2=bufs[3], fx2=bufs[4], fu2=bufs[5],
                                                                X_prev=Xa, U_prev=Ua, X_out=Xb, U_out=Ub, first_cold=True, **lkw, **common)
                solver.sync()
                assert done == steps and all(i["status"] == 0 for i in infos), ("library loop", done, [i["status"] for i in infos])
                X_lib, U_lib = (Xb, Ub) if last_in_out else (Xa, Ua)
                outs.append((res.cpu().numpy(), X_lib.clone(), U_lib.clone()))
            solver.close()
    except AssertionError as e:
        fails += 1
        print(tag + f": {str(e)[:200]}", flush=True)
        continue
    e = max(float((outs[1][1] - outs[0][1]).abs().max()), float((outs[1][2] - outs[0][2]).abs().max()))
    worst = max(worst, e)
    tol = 0.0 if kind == "qp" else 1e-6
    if e > tol or not np.allclose(outs[1][0], outs[0][0], rtol=1e-6 if kind != "qp" else 0.0, atol=1e-9 if kind != "qp" else 0.0):
        fails += 1
        print(tag + f": iterates differ by {e:.3e}; residuals {outs[0][0]} vs {outs[1][0]}", flush=True)
print(f"{cases} cases, {fails} failures, worst difference {worst:.3e}")
