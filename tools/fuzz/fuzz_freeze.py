"""Stage-cone rounds with one consensus stage: the freeze of a negligible shared-control step (option as_freeze_tol) must not change an
answer.  Warm SEQUENCES of related problems with tight boxes (shared controls often sit on a bound and are released in a later round)
through two contexts, as_freeze_tol = 0 against the default; every solve also against the oracle.
usage: fuzz_freeze.py [seed] [sequences] [solves per sequence]"""
import faulthandler, sys, numpy as np, torch
faulthandler.enable()
sys.path.insert(0, ".")
from oracle import lqp_oracle as orc
from pmpc_amd.device import DeviceSolver
from tests.support.problems import rand_problem
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
L = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda")
T = lambda a: dev(np.swapaxes(a, -1, -2))
sa, sb = DeviceSolver(0), DeviceSolver(0)
sa.set_option("as_freeze_tol", 0.0)
worst_ab, worst_o, fails, solves = 0.0, 0.0, 0, 0
for k in range(n):
    x, u = [(12, 4), (4, 2), (3, 3), (5, 3), (6, 2), (8, 4), (7, 3)][rng.integers(7)]
    M, N = int(rng.integers(2, 12)), int(rng.integers(2, 9))
    bu = float(rng.choice([0.05, 0.1, 0.3]))
    args, kw = rand_problem(rng, M, N, x, u, bu)
    q = int(rng.integers(1, u))
    W = np.zeros((q, u)); W[np.arange(q), np.arange(1, q + 1)] = 1.0 + 0.3 * rng.random(q)
    w0 = 0.02 * rng.standard_normal(q)
    v = np.zeros(u); v[0] = 0.3 + 0.4 * rng.random()
    v0 = 0.05 + 0.1 * rng.random()
    u_int = np.zeros(u); u_int[0] = min(0.15, 0.5 * bu)
    amp = float(rng.choice([1e-3, 1e-2, 1e-1]))
    for t in range(L):
        x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
        if t:
            f = f + amp * rng.standard_normal(f.shape)
            U_ref = U_ref + amp * rng.standard_normal(U_ref.shape)
            X_ref = X_ref + amp * rng.standard_normal(X_ref.shape)
            args = (x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref)
        try:
            Xo, Uo = orc.lsoc_solve_py(*args, Nc=1, reg_x=kw["reg_x"], reg_u=kw["reg_u"], u_l=kw["u_l"], u_u=kw["u_u"], soc_W=W, soc_w0=w0, soc_v=v, soc_v0=v0, u_interior=u_int)
        except Exception as e:
            print("skip", type(e).__name__)
            break
        out = []
        for s in (sa, sb):
            X, U, status = s.lsoc_solve(f=dev(f), fx=T(fx), fu=T(fu), X_prev=dev(X_prev), U_prev=dev(U_prev), Q=T(Q), R=T(R), X_ref=dev(X_ref), U_ref=dev(U_ref),
                                        reg_x=kw["reg_x"], reg_u=kw["reg_u"], Nc=1, symmetric_cost=True, soc_W=dev(W), soc_w0=dev(w0), soc_v=dev(v), soc_v0=v0,
                                        soc_u_interior=dev(u_int), lu=dev(kw["u_l"]), uu=dev(kw["u_u"]))
            s.sync()
            out.append((status, X.cpu().numpy(), U.cpu().numpy()))
        solves += 1
        rel = lambda A, B: max(np.linalg.norm(A[1] - B[0]) / max(np.linalg.norm(B[0]), 1e-300), np.linalg.norm(A[2] - B[1]) / max(np.linalg.norm(B[1]), 1.0))
        ok = out[0][0] == 0 and out[1][0] == 0
        eab = rel(out[1], (out[0][1], out[0][2])) if ok else np.inf
        eo = max(rel(out[0], (Xo, Uo)), rel(out[1], (Xo, Uo))) if ok else np.inf
        if not (eab < 1e-7 and eo < 1e-6):
            fails += 1
            print("FAIL", (k, t, M, N, x, u, bu, q, amp), [o[0] for o in out], f"freeze on/off {eab:.2e} vs oracle {eo:.2e}", flush=True)
        worst_ab, worst_o = max(worst_ab, eab), max(worst_o, eo)
print(f"{solves} solves, {fails} failures, worst freeze on/off {worst_ab:.2e}, worst rel err {worst_o:.2e}")
