"""Slew penalties on the MFMA path: with exactly symmetric cost blocks, N >= 2 and
(xdim + udim, udim) among the dimensions the register-resident kernels are built for, a problem with slew penalties is
restated in control increments (pmpc_amd/csrc/kernels_slew.hip) and solved by the same MFMA sweeps as a plain one.  Without
boxes that is one Newton step, exact.  With boxes the restated problem has STATE boxes (the control boxes act on the u-part of
the state [x; u]): since r03 those are rows of the active-set rounds (kernels_xbox.hip), cold and warm — measured 2.1x - 5.4x
faster than the generic kernels (tools/debug/slew_paths.py); PMPC_SLEW_INCREMENT_BOXES=0 puts them back there.
Checked against the oracle (the reference's tridiagonal slew block, PMPC.jl/src/lqp_utils.jl:17-102) and against the generic
kernels, which keep the cross block."""
import numpy as np
import pytest

from tests.support.problems import CASES, rand_problem

pytestmark = pytest.mark.gpu
TOL = 1e-7

SLEW_CASES = [c for c in CASES if (c[7] is not None or c[8] is not None) and c[1] >= 2] + [
    # (M, N, x, u, Nc, u-bound, x-bound, slew, slew0)
    (16, 20, 4, 2, 1, 0.3, None, 0.8, 0.5),     # unicycle dims, consensus on the first control
    (16, 20, 4, 2, -1, 0.3, None, 0.8, None),   # full consensus
    (12, 15, 2, 1, 0, 0.4, None, 2.0, 1.0),     # double-integrator dims (the canonical slew_reg = 1 example)
    (10, 12, 8, 4, 1, 0.3, None, 0.6, 0.2),     # quadrotor-like control count
    (6, 10, 4, 2, 2, 0.3, 8.0, 0.7, None),      # with state boxes
    (5, 8, 3, 1, 1, None, None, None, 0.5),     # first-stage penalty only
    (64, 30, 4, 2, -1, None, None, 1.0, 0.5),   # no boxes, full consensus (the generic kernels' slowest case)
    (33, 25, 8, 4, 1, None, None, 0.4, None),   # no boxes, (12, 4) restated
    (20, 12, 2, 1, 3, None, None, 2.0, 1.0),    # no boxes, (3, 1) restated
    (7, 9, 10, 2, 1, 0.3, None, 0.5, 0.5),      # (12, 2) restated
]


def _solve(s, args, kw, Nc, **extra):
    import torch

    dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda")
    T = lambda a: dev(np.swapaxes(a, -1, -2))
    x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
    opt = dict(f=dev(f), fx=T(fx), fu=T(fu), X_prev=dev(X_prev), U_prev=dev(U_prev), Q=T(Q), R=T(R), X_ref=dev(X_ref), U_ref=dev(U_ref),
               reg_x=kw["reg_x"], reg_u=kw["reg_u"], Nc=Nc, symmetric_cost=True)
    if "u_l" in kw:
        opt.update(lu=dev(kw["u_l"]), uu=dev(kw["u_u"]))
    if "x_l" in kw:
        opt.update(lx=dev(kw["x_l"]), ux=dev(kw["x_u"]))
    if "slew_reg" in kw:
        opt["slew_reg"] = dev(kw["slew_reg"])
    if "slew_reg0" in kw:
        opt.update(slew_reg0=dev(kw["slew_reg0"]), slew_um1=dev(kw["slew_um1"]))
    opt.update(extra)
    X, U, status = s.lqp_solve(**opt)
    s.sync()
    return X.cpu().numpy(), U.cpu().numpy(), status, dict(s.last_info)


def _rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1.0)


@pytest.mark.parametrize("regs", [(1.0, 0.1), (0.05, 0.6), (0.3, 0.3)], ids=["regx>regu", "regx<regu", "equal"])
@pytest.mark.parametrize("case", SLEW_CASES, ids=[str(c) for c in SLEW_CASES])
def test_slew_problems_take_the_mfma_path_and_match_the_oracle(case, regs, oracle):
    from pmpc_amd.device import DeviceSolver

    M, N, x, u, Nc, bu, bx, sl, sl0 = case
    rng = np.random.default_rng(4000 + SLEW_CASES.index(case))
    args, kw = rand_problem(rng, M, N, x, u, bu, bx, sl, sl0)
    kw["reg_x"], kw["reg_u"] = regs
    Xo, Uo = oracle.lqp_solve_py(*args, Nc=Nc, **kw)
    s = DeviceSolver(0)
    X, U, status, info = _solve(s, args, kw, Nc)
    boxed = bu is not None or bx is not None
    assert status == 0 and info["fast_path"] == 1, info
    tol = TOL if boxed else 1e-10  # (no boxes: one Newton step)
    assert _rel(X, Xo) <= tol and _rel(U, Uo) <= tol, (_rel(X, Xo), _rel(U, Uo), info)
    if Nc != 0 and M > 1:
        k = N if Nc < 0 else Nc
        assert np.all(U[:, :k] == U[0:1, :k])
    Xg, Ug, status, info = _solve(s, args, kw, Nc, force_generic=True)  # the kernels that keep the cross block
    assert status == 0 and info["fast_path"] == 0
    assert _rel(Xg, Xo) <= TOL and _rel(Ug, Uo) <= TOL
    s.close()


def test_slew_with_weights_and_an_scp_like_sequence(oracle):
    """Per-particle cost weights scale the slew penalties with everything else (main.jl:96-112), and a sequence of related
    problems through one context (what an SCP loop with slew_reg = 1 does) stays on the oracle."""
    import torch

    from pmpc_amd.device import DeviceSolver

    M, N, x, u, Nc = 12, 14, 4, 2, 1
    rng = np.random.default_rng(91)
    args, kw = rand_problem(rng, M, N, x, u, None, None, 1.0, 0.4)
    wts = 0.5 + rng.random(M)
    s = DeviceSolver(0)
    for t in range(3):
        if t:
            x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
            args = (x0, f + 0.03 * rng.standard_normal(f.shape), fx * (1 + 0.03 * rng.standard_normal(fx.shape)), fu, X_prev, U_prev,
                    Q, R, X_ref, U_ref)
        Xo, Uo = oracle.lqp_solve_py(*args, Nc=Nc, weights=wts, **kw)
        X, U, status, info = _solve(s, args, kw, Nc, weights=torch.tensor(wts, dtype=torch.float64, device="cuda"))
        assert status == 0 and info["fast_path"] == 1
        assert _rel(X, Xo) <= 1e-10 and _rel(U, Uo) <= 1e-10, (t, _rel(X, Xo), _rel(U, Uo))
    s.close()


@pytest.mark.parametrize("case", [(16, 20, 4, 2, 1, 0.3, None), (12, 15, 2, 1, 0, 0.4, None), (6, 10, 4, 2, 2, 0.3, 8.0), (3, 6, 3, 2, 2, 0.3, 5.0)],
                         ids=lambda c: str(c))
def test_boxed_slew_problems_on_the_generic_kernels_when_switched_back(case, oracle):
    """Option slew_increment_boxes = 0 (PMPC_SLEW_INCREMENT_BOXES=0 process-wide): boxed slew problems stay on the generic
    kernels — the r02 default and the comparison leg of tools/debug/slew_paths.py."""
    from pmpc_amd.device import DeviceSolver

    M, N, x, u, Nc, bu, bx = case
    args, kw = rand_problem(np.random.default_rng(300 + M), M, N, x, u, bu, bx, 0.8, 0.5)
    Xo, Uo = oracle.lqp_solve_py(*args, Nc=Nc, **kw)
    s = DeviceSolver(0)
    s.set_option("slew_increment_boxes", 0)
    s.set_option("warn_slow_path", 0)
    X, U, status, info = _solve(s, args, kw, Nc)
    assert status == 0 and info["fast_path"] == 0, info
    assert _rel(X, Xo) <= TOL and _rel(U, Uo) <= TOL
    if Nc > 0:
        assert np.all(U[:, :Nc] == U[0:1, :Nc])
    s.close()


def test_the_reference_canonical_abi_example_runs_on_the_mfma_path():
    """tests/pmpcjl_test.py:164-219 (x2 u1 N=30 Nc=3, slew_reg = 1, u in +-0.4 saturating, x in +-20 holding): VERDICT r02 named it as
    the boxed slew problem that still ran on the generic kernels.  Increment form (3, 1) on the MFMA kernels: the committed golden
    solution, and a warm start of one factorisation.  (Its cold start is hard for ANY primal-dual active-set rule — a double
    integrator whose control saturates over a long window: the rounds hand over to the interior-point iteration on both paths,
    18 iterations + 12 rounds here against 14 + 5 on the generic kernels, 3.6 ms against 6.6 ms, tools/debug/canonical_example.py.)"""
    from pmpc_amd.device import DeviceSolver
    from tests.test_oracle_golden import load_qp

    args, kw, Nc, Xg, Ug, _ = load_qp("qp_double_integrator_u04.npz")
    s = DeviceSolver(0)
    X, U, status, info = _solve(s, args, kw, Nc)
    assert status == 0 and info["fast_path"] == 1, info
    assert _rel(X, Xg) <= TOL and _rel(U, Ug) <= TOL
    assert np.sum(np.abs(np.abs(U) - 0.4) < 1e-9) > 0  # (the control box binds)
    X, U, status, info = _solve(s, args, kw, Nc)
    assert status == 0 and info["ipm_iters"] == 0 and info["structured_solves"] <= 2, info
    assert _rel(X, Xg) <= TOL and _rel(U, Ug) <= TOL
    s.close()


def test_boxed_slew_warm_start_takes_one_round(oracle):
    """A boxed slew problem solved twice through one context: the second solve starts from the first one's set and multipliers
    (state-box rounds of the restated problem) and is accepted after one factorisation."""
    from pmpc_amd.device import DeviceSolver

    M, N, x, u, Nc = 16, 20, 4, 2, 1
    args, kw = rand_problem(np.random.default_rng(77), M, N, x, u, 0.3, None, 0.8, 0.5)
    Xo, Uo = oracle.lqp_solve_py(*args, Nc=Nc, **kw)
    s = DeviceSolver(0)
    X, U, status, cold = _solve(s, args, kw, Nc)
    assert status == 0 and cold["fast_path"] == 1 and cold["ipm_iters"] == 0 and cold["active_set_rounds"] >= 2, cold
    assert _rel(X, Xo) <= TOL and _rel(U, Uo) <= TOL
    X, U, status, warm = _solve(s, args, kw, Nc)
    assert status == 0 and warm["ipm_iters"] == 0 and warm["structured_solves"] <= 2, warm
    assert _rel(X, Xo) <= TOL and _rel(U, Uo) <= TOL
    s.close()


def test_single_stage_slew_stays_on_the_generic_kernels(oracle):
    """N = 1: the reference's diagonal rule (s0 + s on the only stage, lqp_utils.jl:31-39) is not the plain penalty; the
    generic kernels reproduce it, the restated problem would not."""
    from pmpc_amd.device import DeviceSolver

    args, kw = rand_problem(np.random.default_rng(5), 3, 1, 3, 2, 0.3, None, 0.5, 0.3)
    Xo, Uo = oracle.lqp_solve_py(*args, Nc=-1, **kw)
    s = DeviceSolver(0)
    X, U, status, info = _solve(s, args, kw, -1)
    assert status == 0 and info["fast_path"] == 0
    assert _rel(X, Xo) <= TOL and _rel(U, Uo) <= TOL
    s.close()
