"""General stage cones (several rows / cones per stage) on the GPU against the conic oracle, with the rounds' trace."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from oracle import lqp_oracle as orc
from pmpc_amd.device import DeviceSolver
from pmpc_amd.extra_cstrs import stage_cones_from_extra_cstrs
from tests.support.problems import rand_problem
from tests.test_extra_cstrs_gpu import CASES, make_tuples, oracle_solve

dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda")
T = lambda a: dev(np.swapaxes(a, -1, -2))
s = DeviceSolver(0)
if len(sys.argv) > 1 and sys.argv[1] == "grid":
    CASES = [(4, 6, 5, 3, nc, bu, kinds) for nc in (0, 1, 2, -1) for bu in (None, 2.0) for kinds in (["soc"], ["lin2"], ["soc", "lin2"])]
    sys.argv = sys.argv[:1]
sel = [int(v) for v in sys.argv[1:]] or range(len(CASES))
for ci in sel:
    M, N, x, u, Nc, bu, kinds = CASES[ci]
    rng = np.random.default_rng(8200 + ci)
    args, kw = rand_problem(rng, M, N, x, u, bu)
    tuples, ncu = make_tuples(rng, M, N, x, u, Nc, kinds)
    Xo, Uo = oracle_solve(orc, args, kw, Nc, tuples, ncu)
    cn = stage_cones_from_extra_cstrs(tuples, M, N, x, u, Nc)
    x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
    boxes = dict(lu=dev(kw["u_l"]), uu=dev(kw["u_u"])) if bu is not None else {}
    X, U, status = s.lsoc_solve(f=dev(f), fx=T(fx), fu=T(fu), X_prev=dev(X_prev), U_prev=dev(U_prev), Q=T(Q), R=T(R), X_ref=dev(X_ref),
                                U_ref=dev(U_ref), reg_x=kw["reg_x"], reg_u=kw["reg_u"], Nc=Nc, symmetric_cost=True,
                                cones=dict(sizes=cn["sizes"], A=dev(cn["A"]), c=dev(cn["c"])), verbose=0, **boxes)
    s.sync()
    Un = U.cpu().numpy()
    err = np.linalg.norm(Un - Uo) / max(np.linalg.norm(Uo), 1.0)
    sv = np.einsum("mnru,mnu->mnr", cn["A"], Uo) + cn["c"]
    print("CASE", CASES[ci], "status", status, "err", err, "rounds", s.last_info["active_set_rounds"], flush=True)
