import sys; sys.path.insert(0, ".")
import numpy as np
from oracle import lqp_oracle as orc
from pmpc_amd import backend
from tests.support.problems import abi_args, rand_problem
orc.build()
rel = lambda a, b: np.linalg.norm(a - b) / max(np.linalg.norm(b), 1.0)
for M, k in [(40, 20), (24, 1), (40, 30), (600, 300)]:
    rng = np.random.default_rng(7 + M + k)
    args, kw = rand_problem(rng, M, 6, 4, 2, 0.4)
    args = list(args)
    scale = (1.0 + 2.0 * rng.permutation(M) / M)[:, None, None, None]
    args[6], args[7] = args[6] * scale, args[7] * scale
    args = tuple(args)
    Xo, Uo, info = orc.lcone_solve_py(*args, Nc=1, return_info=True, k=k, **kw)
    X, U = backend.lcone_solve(*abi_args(args, kw, 1), smooth_alpha=float("nan"), solver="ecos", k=k, verbose=0)
    print(M, k, "kink" if info.get("kink") else "-", f"{rel(X, Xo):.1e} {rel(U, Uo):.1e}", "theta", info.get("theta"), flush=True)
    if (M, k) == (40, 20):
        w = info["weights"]
        print("oracle fractional weights:", [(int(i), float(w[i])) for i in np.where((w > 1e-6) & (w < 1.0005))[0]])
    if (M, k) == (40, 20):
        pe = np.linalg.norm((X - Xo).reshape(M, -1), axis=1) / np.linalg.norm(Xo.reshape(M, -1), axis=1)
        print("per-particle rel err:", np.round(np.log10(pe + 1e-20), 1))
        print("weights (oracle):", np.round(w, 3))
        print("consensus control err:", np.abs(U[:, 0] - Uo[:, 0]).max(), "U0 dev", U[0, 0], "orc", Uo[0, 0])
        J = orc.particle_costs_py(X, U, *args[4:], reg_x=kw["reg_x"], reg_u=kw["reg_u"])
        Jo = orc.particle_costs_py(Xo, Uo, *args[4:], reg_x=kw["reg_x"], reg_u=kw["reg_u"])
        print("J dev[3,25]", J[3], J[25], "J orc", Jo[3], Jo[25])
