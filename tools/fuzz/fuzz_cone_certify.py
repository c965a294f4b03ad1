"""The log-barrier cases of fuzz_cone.py (same seeds, same problems) with the optimality certificate next to the comparison: where the device
and the oracle differ by more than 1e-7, which of the two points satisfies the smoothed program's KKT conditions better?
(tests/support/kkt_certificate.py: from the ABI data alone; control boxes only, k = M)   usage: fuzz_cone_certify.py SEED CASES [MAX_M = 24] [MAX_M_DIRECT = 8]"""
import sys

import numpy as np

sys.path.insert(0, ".")
from oracle import lqp_oracle as orc
from pmpc_amd import backend
from tests.support.kkt_certificate import smoothed_cone_certificate
from tests.support.problems import abi_args, rand_problem

seed, cases = int(sys.argv[1]), int(sys.argv[2])
max_M = int(sys.argv[3]) if len(sys.argv) > 3 else 24
max_M_direct = int(sys.argv[4]) if len(sys.argv) > 4 else 8
rng = np.random.default_rng(seed)
for case in range(cases):  # (the draws of fuzz_cone.py, in its order)
    M, N = int(rng.integers(2, max_M + 1)), int(rng.integers(3, 9))
    x, u = [(4, 2), (3, 2), (6, 3), (5, 2), (4, 3), (6, 2)][int(rng.integers(0, 6))]
    Nc = int(rng.choice([0, 1, 1, 2, -1]))
    bu = float(rng.choice([0.4, 1.0, 2.5]))
    bx = 5.0 if rng.random() < 0.25 else None
    kind = str(rng.choice(["hard", "hard", "logbarrier", "logbarrier", "squareplus"]))
    alpha = float("nan") if kind == "hard" else float(rng.choice([1.0, 10.0, 100.0]))
    k = None if rng.random() < 0.7 else int(rng.integers(1, M + 1))
    if kind == "squareplus":
        Nc, M = (Nc if Nc in (0, 1) else 1), min(M, 6, max_M_direct)
    if k is not None:
        M = min(M, max_M_direct)
        k = min(k, M)
        k = None if k == M else k
    args, kw = rand_problem(rng, M, N, x, u, bu, bx)
    if kind != "logbarrier" or k is not None:
        continue
    try:
        Xo, Uo = orc.lcone_solve_py(*args, Nc=Nc, smooth_alpha=alpha, **kw)
    except BaseException:
        continue
    X, U = backend.lcone_solve(*abi_args(args, kw, Nc), smooth_alpha=alpha, solver="ecos", smooth_cstr="logbarrier", smooth_beta=1.0)
    rel = lambda a_, b_: np.linalg.norm(a_ - b_) / max(np.linalg.norm(b_), 1.0)
    e = max(rel(X, Xo), rel(U, Uo))
    if e <= 1e-7:
        continue
    if bx is not None:  # (the certificate knows control boxes only)
        print(f"case {case}: M{M} N{N} x{x} u{u} Nc{Nc} bu{bu} bx{bx} alpha {alpha}: device vs oracle {e:.2e} (state boxes: no certificate)", flush=True)
        continue
    worst = lambda c: max(v for kk, v in c.items() if isinstance(v, float) and kk not in ("slack",))
    cd = smoothed_cone_certificate(*args, kw["reg_x"], kw["reg_u"], Nc, kw["u_l"], kw["u_u"], X, U, alpha)
    cO = smoothed_cone_certificate(*args, kw["reg_x"], kw["reg_u"], Nc, kw["u_l"], kw["u_u"], Xo, Uo, alpha)
    print(f"case {case}: M{M} N{N} x{x} u{u} Nc{Nc} bu{bu} alpha {alpha}: device vs oracle {e:.2e};  worst KKT residual  device {worst(cd):.2e}  oracle {worst(cO):.2e}", flush=True)
    print("   device", {kk: f"{v:.1e}" for kk, v in cd.items() if isinstance(v, float)})
    print("   oracle", {kk: f"{v:.1e}" for kk, v in cO.items() if isinstance(v, float)})
