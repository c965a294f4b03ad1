"""One-off stress of the stage-cone path vs its oracle: random dims / consensus horizons / cone data."""
import faulthandler, sys, numpy as np, torch
faulthandler.enable()
sys.path.insert(0, ".")
from oracle import lqp_oracle as orc
from pmpc_amd.device import DeviceSolver
from tests.support.problems import rand_problem
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda")
T = lambda a: dev(np.swapaxes(a, -1, -2))
s = DeviceSolver(0)
worst, fails, worst_case = 0.0, 0, None
for k in range(int(sys.argv[2]) if len(sys.argv) > 2 else 50):
    x, u = [(12, 4), (4, 2), (3, 3), (5, 3), (6, 2), (8, 4), (7, 3)][rng.integers(7)]
    M, N = int(rng.integers(1, 7)), int(rng.integers(1, 9))
    Nc = int(rng.choice([0, min(1, N), -1]))
    bu = None if rng.random() < 0.3 else float(rng.choice([0.4, 1.0]))
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
    X, U, status = s.lsoc_solve(f=dev(f), fx=T(fx), fu=T(fu), X_prev=dev(X_prev), U_prev=dev(U_prev), Q=T(Q), R=T(R), X_ref=dev(X_ref),
                                U_ref=dev(U_ref), reg_x=kw["reg_x"], reg_u=kw["reg_u"], Nc=Nc, symmetric_cost=True, soc_W=dev(W),
                                soc_w0=dev(w0), soc_v=dev(v), soc_v0=v0, soc_u_interior=dev(u_int), **bounds)
    s.sync()
    X, U = X.cpu().numpy(), U.cpu().numpy()
    err = max(np.linalg.norm(X - Xo) / max(np.linalg.norm(Xo), 1e-300), np.linalg.norm(U - Uo) / max(np.linalg.norm(Uo), 1.0)) if status == 0 else np.inf
    if not err < 1e-6:
        fails += 1
        print("FAIL", (M, N, x, u, Nc, bu, q), status, err, s.last_info["ipm_iters"])
    if err > worst and np.isfinite(err):
        worst_case = ((M, N, x, u, Nc, bu, q), k)
    worst = max(worst, err)
    if "--kkt" in sys.argv and status == 0 and err > 1e-9:
        # which side is off?  The joint program's KKT conditions at BOTH points, from the data alone (tests/support/kkt_certificate.py)
        from tests.support.kkt_certificate import kkt_certificate
        lo, hi = (kw["u_l"], kw["u_u"]) if bu is not None else (np.full(U.shape, -np.inf), np.full(U.shape, np.inf))
        cs = [kkt_certificate(x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref, kw["reg_x"], kw["reg_u"], Nc, lo, hi, Xc, Uc, soc=dict(W=W, w0=w0, v=v, v0=v0), tol_act=1e-7)
              for Xc, Uc in ((X, U), (Xo, Uo))]
        print(f"case {k} {(M, N, x, u, Nc, bu, q)} err {err:.1e}: stationarity device {max(cs[0]['stationarity'], cs[0]['stationarity_shared']):.1e} oracle "
              f"{max(cs[1]['stationarity'], cs[1]['stationarity_shared']):.1e}; cone residual device {cs[0]['cone']:.1e} oracle {cs[1]['cone']:.1e}", flush=True)
print(f"{fails} failures, worst rel err {worst:.2e}" + (f" (case {worst_case[1]}: {worst_case[0]})" if worst > 0 else ""))
