"""Slew problems: the increment form on the MFMA kernels against the generic kernels (cross block kept), time per solve and
difference of the answers.  usage: slew_paths.py  (GPU box)"""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from pmpc_amd.device import DeviceSolver
from tests.support.problems import rand_problem

dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda")
T = lambda a: dev(np.swapaxes(a, -1, -2))
for (M, N, x, u, Nc) in [(256, 30, 4, 2, 1), (1024, 30, 4, 2, 1), (256, 30, 4, 2, -1), (1024, 50, 8, 4, 1), (4096, 30, 2, 1, 1)]:
    rng = np.random.default_rng(1)
    args, kw = rand_problem(rng, M, N, x, u, 0.3, None, 1.0, 0.5)
    x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
    opt = dict(f=dev(f), fx=T(fx), fu=T(fu), X_prev=dev(X_prev), U_prev=dev(U_prev), Q=T(Q), R=T(R), X_ref=dev(X_ref), U_ref=dev(U_ref),
               reg_x=kw["reg_x"], reg_u=kw["reg_u"], Nc=Nc, symmetric_cost=True, lu=dev(kw["u_l"]), uu=dev(kw["u_u"]),
               slew_reg=dev(kw["slew_reg"]), slew_reg0=dev(kw["slew_reg0"]), slew_um1=dev(kw["slew_um1"]))
    s = DeviceSolver(0)
    out = {}
    for name, extra in (("increment form (MFMA)", {}), ("generic", dict(force_generic=True))):
        for cold in (True, False):
            ts = []
            for rep in range(4):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                X, U, st = s.lqp_solve(cold_start=cold, **opt, **extra)
                s.sync()
                ts.append(time.perf_counter() - t0)
            info = dict(s.last_info)
            print(f"M={M} N={N} x={x} u={u} Nc={Nc} {name:22s} cold={cold}: {1e3 * min(ts[1:]):8.3f} ms  ipm {info['ipm_iters']} rounds {info['active_set_rounds']} "
                  f"factorisations {info['structured_solves']} status {st}", flush=True)
        out[name] = (X.cpu().numpy(), U.cpu().numpy())
    a, b = out["increment form (MFMA)"], out["generic"]
    print("   difference of the two answers: X %.2e U %.2e" % (np.linalg.norm(a[0] - b[0]) / np.linalg.norm(b[0]), np.linalg.norm(a[1] - b[1]) / np.linalg.norm(b[1])))
    s.close()
