"""Sharded solve at bench shape on ONE GPU through the in-process communicator (tests/test_multirank_gpu.py machinery):
quadrotor N=50, M particles over 2 / 4 / 8 ranks, a few warm-started solves, against the single-rank result."""
import sys, numpy as np
sys.path.insert(0, ".")
from pmpc_amd import dynamics as dyn
from tests.test_multirank_gpu import _solve_sharded
M = int(sys.argv[1]) if len(sys.argv) > 1 else 512
prob = dyn.make_quadrotor_problem(M=M, N=50)
f, fx, fu = prob["f_fx_fu_fn"](np.concatenate([prob["x0"][:, None, :], prob["X_prev"][:, :-1, :]], 1), prob["U_prev"])
args = (prob["x0"], f, fx, fu, prob["X_prev"], prob["U_prev"], prob["Q"], prob["R"], prob["X_ref"], prob["U_ref"])
kw = dict(reg_x=prob["reg_x"], reg_u=prob["reg_u"], u_l=prob["u_l"], u_u=prob["u_u"])
X1, U1, i1 = _solve_sharded(args, kw, 1, 1, repeats=3)
print("single rank:", {k: i1[0][k] for k in ("ipm_iters", "active_set_rounds", "structured_solves", "fast_path")})
for world in (2, 4, 8):
    Xw, Uw, infos = _solve_sharded(args, kw, 1, world, repeats=3)
    ex = np.linalg.norm(Xw - X1) / np.linalg.norm(X1); eu = np.linalg.norm(Uw - U1) / np.linalg.norm(U1)
    same = len({(i["ipm_iters"], i["active_set_rounds"]) for i in infos}) == 1
    print(f"world {world}: rel diff X {ex:.2e} U {eu:.2e}; rounds {infos[0]['active_set_rounds']} ipm {infos[0]['ipm_iters']}; ranks agree: {same}; "
          f"consensus bit-identical: {bool(np.all(Uw[:, :1] == Uw[0:1, :1]))}")
    assert ex < 1e-9 and eu < 1e-9 and same
print("SHARDED_OK")
