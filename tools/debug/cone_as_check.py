"""Stage cones inside the active-set rounds (kernels_cone.hip) on the GPU: random problems against the cone oracle, then the
quadrotor SCP loop with thrust cones (rounds per iteration).  `python tools/debug/cone_as_check.py [nrand] [M] [N] [steps]`"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import lqp_oracle as orc
from pmpc_amd import dynamics as dyn
from pmpc_amd.device import MODEL_QUADROTOR, DeviceSolver, to_device_problem
from tests.support.problems import rand_problem

nrand = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda")
T = lambda a: dev(np.swapaxes(a, -1, -2))
s = DeviceSolver(0)
rng = np.random.default_rng(5)
worst, fails, paths = 0.0, 0, []
for k in range(nrand):
    x, u = [(12, 4), (4, 2), (3, 3), (5, 3), (8, 4), (6, 3)][rng.integers(6)]
    M, N = int(rng.integers(1, 9)), int(rng.integers(2, 12))
    Nc = int(rng.choice([0, 1, -1]))
    bu = None if rng.random() < 0.25 else float(rng.choice([0.4, 1.0]))
    args, kw = rand_problem(rng, M, N, x, u, bu)
    q = int(rng.integers(1, u))
    W = np.zeros((q, u)); W[np.arange(q), np.arange(1, q + 1)] = 1.0 + 0.3 * rng.random(q)
    w0 = 0.02 * rng.standard_normal(q)
    v = np.zeros(u); v[0] = 0.3 + 0.4 * rng.random()
    v0 = 0.05 + 0.1 * rng.random()
    u_int = np.zeros(u); u_int[0] = 0.15
    try:
        Xo, Uo = orc.lsoc_solve_py(*args, Nc=Nc, reg_x=kw["reg_x"], reg_u=kw["reg_u"], u_l=kw.get("u_l"), u_u=kw.get("u_u"), soc_W=W, soc_w0=w0,
                                   soc_v=v, soc_v0=v0, u_interior=u_int)
    except Exception as e:
        print("skip", type(e).__name__)
        continue
    x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
    bounds = dict(lu=dev(kw["u_l"]), uu=dev(kw["u_u"])) if bu is not None else {}
    for rep in range(2):  # second call: warm-started from the first one's set and multipliers
        X, U, status = s.lsoc_solve(f=dev(f), fx=T(fx), fu=T(fu), X_prev=dev(X_prev), U_prev=dev(U_prev), Q=T(Q), R=T(R), X_ref=dev(X_ref),
                                    U_ref=dev(U_ref), reg_x=kw["reg_x"], reg_u=kw["reg_u"], Nc=Nc, symmetric_cost=True, soc_W=dev(W),
                                    soc_w0=dev(w0), soc_v=dev(v), soc_v0=v0, soc_u_interior=dev(u_int), verbose=(k < 2), **bounds)
        s.sync()
        Xn, Un = X.cpu().numpy(), U.cpu().numpy()
        err = max(np.linalg.norm(Xn - Xo) / max(np.linalg.norm(Xo), 1e-300), np.linalg.norm(Un - Uo) / max(np.linalg.norm(Uo), 1.0)) if status == 0 else np.inf
        info = s.last_info
        paths.append((info["active_set_rounds"], info["ipm_iters"]))
        if not err < 1e-7:
            fails += 1
        print(("FAIL " if not err < 1e-7 else "ok   "), (M, N, x, u, Nc, bu, q), "rep", rep, "status", status, f"err {err:.2e}", "rounds", info["active_set_rounds"], "ipm", info["ipm_iters"], flush=True)
        worst = max(worst, err if np.isfinite(err) else 0)
print(f"random: {fails} failures, worst rel err {worst:.2e}; solves through the rounds alone: {sum(1 for a, b in paths if b == 0)} of {len(paths)}")

# ---- quadrotor SCP loop with thrust cones ---------------------------------------------------------------------------------------
M = int(sys.argv[2]) if len(sys.argv) > 2 else 256
N = int(sys.argv[3]) if len(sys.argv) > 3 else 100
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 12
prob = dyn.make_quadrotor_problem(M=M, N=N, Nc=1)
d = to_device_problem(prob)
Wc = torch.zeros((2, 4), dtype=torch.float64, device="cuda"); Wc[0, 1] = Wc[1, 2] = 1.0
soc_kw = dict(soc_W=Wc, soc_w0=torch.zeros(2, dtype=torch.float64, device="cuda"), soc_v=torch.tensor([0.3, 0, 0, 0.0], dtype=torch.float64, device="cuda"),
              soc_v0=0.0, soc_u_interior=torch.tensor([9.81, 0, 0, 0.0], dtype=torch.float64, device="cuda"))
Xa, Ua = d["X_prev"].clone(), d["U_prev"].clone()
Xb, Ub = torch.empty_like(Xa), torch.empty_like(Ua)
t0 = time.time()
for it in range(steps):
    f, fx, fu = s.linearize(MODEL_QUADROTOR, d["x0"], Xa, Ua, d["params"])
    X, U, status = s.lsoc_solve(**soc_kw, f=f, fx=fx, fu=fu, X_prev=Xa, U_prev=Ua, Q=d["Q"], R=d["R"], X_ref=d["X_ref"], U_ref=d["U_ref"],
                                reg_x=prob["reg_x"], reg_u=prob["reg_u"], Nc=1, x0=d["x0"], lu=d["lu"], uu=d["uu"], X_out=Xb, U_out=Ub,
                                symmetric_cost=True, static_cons_bounds=True, prev_is_last_solution=it > 0, verbose=(it in (0, 3)))
    s.sync()
    res = float(s.scp_residual(Xb, Xa, Ub, Ua)[0])
    Un = Ub.cpu().numpy()
    viol = float(np.max(np.linalg.norm(Un[..., 1:3], axis=-1) - 0.3 * Un[..., 0]))
    print(f"SCP it {it + 1}: status {status} rounds {s.last_info['active_set_rounds']} ipm {s.last_info['ipm_iters']} resid {res:.3e} cone viol {viol:.1e} Tmin {Un[..., 0].min():.2e}", flush=True)
    if M <= 16 and it in (1, 4):  # oracle check of one warm-started sub-problem
        fh, fxh, fuh = prob["f_fx_fu_fn"](np.concatenate([prob["x0"][:, None, :], Xa.cpu().numpy()[:, :-1]], 1), Ua.cpu().numpy())
        Xo, Uo = orc.lsoc_solve_py(prob["x0"], fh, fxh, fuh, Xa.cpu().numpy(), Ua.cpu().numpy(), prob["Q"], prob["R"], prob["X_ref"], prob["U_ref"],
                                   reg_x=prob["reg_x"], reg_u=prob["reg_u"], Nc=1, u_l=prob["u_l"], u_u=prob["u_u"], soc_W=Wc.cpu().numpy(),
                                   soc_w0=np.zeros(2), soc_v=np.array([0.3, 0, 0, 0.0]), soc_v0=0.0, u_interior=np.array([9.81, 0, 0, 0.0]))
        print(f"      vs oracle: X {np.linalg.norm(Xb.cpu().numpy() - Xo) / np.linalg.norm(Xo):.2e} U {np.linalg.norm(Un - Uo) / np.linalg.norm(Uo):.2e}")
    if status != 0:
        break
    Xa, Xb, Ua, Ub = Xb, Xa, Ub, Ua
print(f"loop wall {time.time() - t0:.2f}s")
