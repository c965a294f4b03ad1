"""State boxes inside the active-set rounds (pmpc_amd/csrc/kernels_xbox.hip).  The reference hands
x_l <= x <= x_u to its sparse solver like every other row (PMPC.jl/src/lqp_utils.jl:318-333); here a binding state box is a row
of a semismooth Newton iteration that rides in the rounds of the control boxes — cold start, warm start and finish of the
interior-point iteration — instead of sending every such solve through the full interior-point iteration.
Problems: tests/support/problems.py::xbox_problem (boxes that bind, feasible by construction).  Tolerance: the module-wide 1e-7
of tests/test_parity_gpu.py (north star: 1e-6)."""
import numpy as np
import pytest

from tests.support.problems import abi_args, xbox_problem
from tests.test_slew_gpu import _rel, _solve

pytestmark = pytest.mark.gpu
TOL = 1e-7

# (M, N, x, u, Nc, u-bound, pull) — pull: how far the boxes let the optimum go towards the box-free one (smaller = more rows bind)
XBOX_CASES = [
    (6, 10, 4, 2, 1, None, 0.8),
    (6, 10, 4, 2, 1, 0.4, 0.8),
    (5, 12, 12, 4, 1, 0.4, 0.9),
    (5, 12, 12, 4, 0, None, 0.7),
    (4, 9, 6, 3, 2, 0.5, 0.8),
    (3, 8, 2, 1, -1, 0.5, 0.7),
    (7, 6, 3, 1, 1, None, 0.6),
    (1, 9, 5, 2, 0, 0.4, 0.8),
    (4, 8, 8, 4, 3, None, 0.5),     # many rows bind: the rounds may hand over to the interior-point iteration and finish it
    (8, 10, 6, 2, 1, 0.3, 0.45),
    (33, 7, 9, 3, 1, 0.5, 0.8),
    # edge cases: single-stage and two-stage horizons, one particle, full consensus, many particles
    (4, 1, 4, 2, 0, 0.4, 0.8),
    (4, 1, 4, 2, 1, 0.4, 0.8),
    (1, 1, 2, 1, 0, None, 0.8),
    (3, 2, 12, 4, 1, 0.4, 0.8),
    (5, 1, 3, 3, -1, None, 0.8),
    (2, 2, 6, 2, 2, 0.5, 0.8),
    (300, 3, 2, 1, 1, 0.5, 0.8),
]


@pytest.mark.parametrize("case", XBOX_CASES, ids=[str(c) for c in XBOX_CASES])
def test_binding_state_boxes_match_the_oracle_cold_and_warm(case, oracle):
    from pmpc_amd import backend

    M, N, x, u, Nc, bu, pull = case
    args, kw = xbox_problem(np.random.default_rng(5100 + XBOX_CASES.index(case)), oracle, M, N, x, u, Nc, bu, pull=pull)
    Xo, Uo = oracle.lqp_solve_py(*args, Nc=Nc, **kw)
    binding = int(np.sum((Xo <= kw["x_l"] + 1e-9) | (Xo >= kw["x_u"] - 1e-9)))
    assert binding > 0  # (the generator's promise)
    for rep in range(2):  # c_lqp_solve keeps its context: the second call is warm-started
        X, U = backend.lqp_solve(*abi_args(args, kw, Nc))
        assert _rel(X, Xo) <= TOL and _rel(U, Uo) <= TOL, (rep, binding, _rel(X, Xo), _rel(U, Uo))
        assert np.all(X >= kw["x_l"] - 1e-8) and np.all(X <= kw["x_u"] + 1e-8)
    if Nc != 0 and M > 1:
        k = N if Nc < 0 else Nc
        assert np.all(U[:, :k] == U[0:1, :k])


@pytest.mark.parametrize("pull,cold_in_rounds", [(0.97, True), (0.95, False)], ids=["1-row-binds", "26-rows-bind"])
def test_state_rows_cold_warm_and_next_problem(pull, cold_in_rounds, oracle):
    """A state limit touched here and there — the common case of an MPC problem — never sees the interior-point iteration: cold start
    in two phases (control boxes, then the state rows from that optimum), warm start from the previous set in one factorisation.
    With more rows binding the cold start may hand over to the interior-point iteration (whose last iterations the rounds replace);
    the warm start on the same problem still takes one factorisation."""
    from pmpc_amd.device import DeviceSolver

    M, N, x, u, Nc = 16, 20, 4, 2, 1
    args, kw = xbox_problem(np.random.default_rng(11), oracle, M, N, x, u, Nc, 0.5, pull=pull, margin=0.05)
    Xo, Uo = oracle.lqp_solve_py(*args, Nc=Nc, **kw)
    assert np.sum((Xo <= kw["x_l"] + 1e-9) | (Xo >= kw["x_u"] - 1e-9)) == (1 if cold_in_rounds else 26)
    s = DeviceSolver(0)
    X, U, status, cold = _solve(s, args, kw, Nc)
    assert status == 0 and cold["fast_path"] == 1 and cold["active_set_rounds"] >= 2, cold
    if cold_in_rounds:
        assert cold["ipm_iters"] == 0, cold
    assert _rel(X, Xo) <= TOL and _rel(U, Uo) <= TOL
    X, U, status, warm = _solve(s, args, kw, Nc)
    assert status == 0 and warm["ipm_iters"] == 0 and warm["structured_solves"] <= 2, warm
    assert _rel(X, Xo) <= TOL and _rel(U, Uo) <= TOL
    # a perturbed problem (what the next SCP iteration hands over): exact again; with one row binding still no interior-point
    # iteration (with 26 of these tight boxes binding, a 1 % change of f makes ~200 rows change status in the first round and the
    # rounds hand over — recorded in DESIGN.md as the limit of the state rows)
    x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
    rng = np.random.default_rng(12)
    args2 = (x0, f + 0.01 * rng.standard_normal(f.shape), fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref)
    Xo2, Uo2 = oracle.lqp_solve_py(*args2, Nc=Nc, **kw)
    X, U, status, nxt = _solve(s, args2, kw, Nc)
    assert status == 0 and (nxt["ipm_iters"] == 0 or not cold_in_rounds), nxt
    assert _rel(X, Xo2) <= TOL and _rel(U, Uo2) <= TOL
    s.close()


def test_state_box_rounds_can_be_switched_off_per_context(oracle):
    """Option xbox_as = 0 (pmpc_set_option; PMPC_XBOX_AS=0 sets the same default process-wide): the r02 behaviour — a binding state
    box sends the solve to the interior-point iteration — stays reachable, on one context while another keeps the rounds."""
    from pmpc_amd.device import DeviceSolver

    args, kw = xbox_problem(np.random.default_rng(11), oracle, 16, 20, 4, 2, 1, 0.5, pull=0.97, margin=0.05)
    Xo, Uo = oracle.lqp_solve_py(*args, Nc=1, **kw)
    off, on = DeviceSolver(0), DeviceSolver(0)
    assert off.get_option("xbox_as") == 1.0
    off.set_option("xbox_as", 0)
    with pytest.raises(KeyError):
        off.set_option("no_such_option", 1)
    X, U, status, info = _solve(off, args, kw, 1)
    assert status == 0 and info["ipm_iters"] > 0, info
    assert _rel(X, Xo) <= 1e-6 and _rel(U, Uo) <= 1e-6  # (the interior-point iteration alone: the north-star tolerance)
    X, U, status, info = _solve(on, args, kw, 1)
    assert status == 0 and info["ipm_iters"] == 0, info
    assert _rel(X, Xo) <= TOL and _rel(U, Uo) <= TOL
    off.close()
    on.close()


def test_library_scp_loop_with_a_binding_velocity_limit():
    """What `bench.py --vmax` times: the quadrotor SCP loop with |v| <= 2.5 m/s inside the library (pmpc_scp_loop_device: warm starts
    from the previous iteration's set and multipliers, the first round carries the linearisation point's dynamics defect, follow-up
    work speculated behind the first batch of rounds — and redone when a second phase follows).  The same iterates as a
    Python-driven loop of single calls, every iterate inside the limit, the limit touched."""
    import torch

    from pmpc_amd import dynamics as dyn
    from pmpc_amd.device import MODEL_QUADROTOR, DeviceSolver, to_device_problem

    M, N, Nc, vmax, steps = 32, 30, 1, 2.5, 6
    prob = dyn.make_quadrotor_problem(M=M, N=N, Nc=Nc)
    d = to_device_problem(prob)
    lx = torch.full((M, N, 12), -float("inf"), dtype=torch.float64, device="cuda")
    lx[..., 3:6] = -vmax
    common = dict(Q=d["Q"], R=d["R"], X_ref=d["X_ref"], U_ref=d["U_ref"], reg_x=prob["reg_x"], reg_u=prob["reg_u"], Nc=Nc, x0=d["x0"], lu=d["lu"],
                  uu=d["uu"], lx=lx, ux=-lx, symmetric_cost=True)
    s1, s2 = DeviceSolver(0), DeviceSolver(0)
    Xa, Ua = d["X_prev"].clone(), d["U_prev"].clone()
    Xb, Ub = torch.empty_like(Xa), torch.empty_like(Ua)
    res_py, rounds = [], 0
    for it in range(steps):
        f, fx, fu = s1.linearize(MODEL_QUADROTOR, d["x0"], Xa, Ua, d["params"])
        _, _, st = s1.lqp_solve(f=f, fx=fx, fu=fu, X_prev=Xa, U_prev=Ua, X_out=Xb, U_out=Ub, static_cons_bounds=True, prev_is_last_solution=it > 0,
                                cold_start=it == 0, **common)
        assert st == 0 and s1.last_info["fast_path"] == 1
        rounds += s1.last_info["active_set_rounds"]
        res_py.append(float(s1.scp_residual(Xb, Xa, Ub, Ua)[0].item()))
        Xa, Xb, Ua, Ub = Xb, Xa, Ub, Ua
        v = Xa[..., 3:6].abs().max().item()
        assert v <= vmax + 1e-8, (it, v)
    X_py, U_py = Xa.clone(), Ua.clone()
    assert rounds > 0 and (X_py[..., 3:6].abs() > vmax - 1e-9).sum().item() > 0  # (the limit binds)
    Xa, Ua = d["X_prev"].clone(), d["U_prev"].clone()
    Xb, Ub = torch.empty_like(Xa), torch.empty_like(Ua)
    mk = lambda *shape: torch.empty(shape, dtype=torch.float64, device="cuda")
    bufs = [mk(M, N, 12), mk(M, N, 12, 12), mk(M, N, 4, 12), mk(M, N, 12), mk(M, N, 12, 12), mk(M, N, 4, 12)]
    res, infos, last_in_out, done = s2.scp_loop(MODEL_QUADROTOR, d["params"], steps, f=bufs[0], fx=bufs[1], fu=bufs[2], f2=bufs[3], fx2=bufs[4],
                                                fu2=bufs[5], X_prev=Xa, U_prev=Ua, X_out=Xb, U_out=Ub, first_cold=True, **common)
    s2.sync()
    assert done == steps and all(i["status"] == 0 for i in infos)
    X_lib, U_lib = (Xb, Ub) if last_in_out else (Xa, Ua)
    rel = lambda a, b: (a - b).norm().item() / max(b.norm().item(), 1.0)
    assert rel(X_lib, X_py) <= 1e-9 and rel(U_lib, U_py) <= 1e-9, (rel(X_lib, X_py), rel(U_lib, U_py))
    np.testing.assert_allclose(res.cpu().numpy(), np.array(res_py), rtol=1e-7, atol=1e-12)
    s1.close()
    s2.close()
