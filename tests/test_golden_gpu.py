"""GPU vs the committed golden fixtures (tests/golden/make_golden.py): sub-problems through `c_lqp_solve`,
and the full SCP loop (pmpc_amd.solve, HIP back end) against the hist rows of the reference's own loop."""
from pathlib import Path

import numpy as np
import pytest

from tests.support.problems import abi_args
from tests.test_oracle_golden import CONE_FILES, QP_FILES, load_qp

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).resolve().parent / "golden"


@pytest.mark.parametrize("name", QP_FILES)
def test_c_lqp_solve_matches_golden(name):
    from pmpc_amd import backend

    args, kw, Nc, Xg, Ug, _ = load_qp(name)
    X, U = backend.lqp_solve(*abi_args(args, kw, Nc))
    assert np.linalg.norm(X - Xg) / np.linalg.norm(Xg) < 1e-7
    assert np.linalg.norm(U - Ug) / max(np.linalg.norm(Ug), 1.0) < 1e-7


@pytest.mark.parametrize("name", CONE_FILES)
def test_c_lcone_solve_matches_golden(name):
    """smooth_alpha = NaN => hard constraints (PMPC.jl/src/main.jl:242-244): the epsilon-anchored epigraph objective of
    the cone path, M = 1 (same minimiser as the QP) and the 24-particle chain (cheapest particle at weight 1+eps-2 eps M)."""
    from pmpc_amd import backend

    args, kw, Nc, Xg, Ug, _ = load_qp(name)
    X, U = backend.lcone_solve(*abi_args(args, kw, Nc), smooth_alpha=float("nan"), solver="ecos")
    assert np.linalg.norm(X - Xg) / np.linalg.norm(Xg) < 1e-7 and np.linalg.norm(U - Ug) / max(np.linalg.norm(Ug), 1.0) < 1e-7


@pytest.mark.parametrize("name,regs", [("scp_unicycle_simple.npz", {}), ("scp_unicycle_remote.npz", dict(reg_x=1.0, reg_u=1.0))])
def test_scp_solve_on_gpu_matches_reference_loop(name, regs):
    import pmpc_amd
    from pmpc_amd import dynamics as dyn

    g = np.load(GOLD / name)
    N = int(g["N"])
    p = np.array([1.0, 1.0, 0.3])
    Q, R = np.tile(np.eye(4), (N, 1, 1)), np.tile(1e-2 * np.eye(2), (N, 1, 1))
    X, U, data = pmpc_amd.solve(lambda X, U: dyn.unicycle(X, U, p), Q, R, np.ones(4), np.zeros((N, 4)), np.zeros((N, 2)),
                                np.zeros((N, 4)), np.zeros((N, 2)), u_l=-np.ones((N, 2)), u_u=np.ones((N, 2)),
                                max_it=int(g["max_it"]), solver_settings=dict(solver="osqp"), **regs)
    hist = np.array([[h["it"], h["obj"], h["resid"]] for h in data["hist"]])
    assert hist.shape[0] == g["hist"].shape[0]
    # tolerances as in tests/test_host_logic.py (first linearisation sits on the u2 = 1e-6 singular point)
    np.testing.assert_allclose(hist[:, 1], g["hist"][:, 1], rtol=2e-4)
    np.testing.assert_allclose(hist[5:, 1], g["hist"][5:, 1], rtol=1e-5)
    np.testing.assert_allclose(X, g["X"], atol=2e-5)
    np.testing.assert_allclose(U, g["U"], atol=2e-5)


def test_failure_convention_nan_outputs():
    """A QP whose Hessian is not positive definite fails loudly: NaN outputs (osqp_solver.jl:65-71) ->
    (None, None, None) from the host loop (pmpc/scp_mpc.py:391-394)."""
    from pmpc_amd import backend
    from tests.support.problems import rand_problem

    args, kw = rand_problem(np.random.default_rng(3), 2, 5, 3, 2)
    args = list(args)
    args[7] = -np.tile(np.eye(2), (2, 5, 1, 1))  # R = -I, reg_u = 0.1  ->  Huu indefinite
    X, U = backend.lqp_solve(*abi_args(tuple(args), kw, 0))
    assert np.all(np.isnan(X)) and np.all(np.isnan(U))


@pytest.mark.parametrize("solver", ["ecos", "osqp"])
def test_scp_on_gpu_reproduces_reference_notebook_table(solver):
    """The HIP back end through `pmpc_amd.solve` against the table the reference's own Julia + ECOS stack printed
    (examples/gpu_solver.ipynb, 50 SCP iterations, 4 digits) — cone path ("ecos", as in the notebook) and QP path."""
    import pmpc_amd
    from tests.support import notebook_problem as nbp

    args, kw, table = nbp.load()
    X, U, data = pmpc_amd.solve(*args, solver_settings=dict(solver=solver), **kw)
    nbp.check_rows(data["hist"], table)


def test_batched_independent_problems_match_sequential_solves():
    """`solve_problems(batched=True)`: K unrelated problems as K particles with Nc = 0 give the
    trajectories of K separate `solve` calls run for the same number of SCP iterations."""
    import pmpc_amd
    from pmpc_amd import dynamics as dyn

    N, xdim, udim, K = 15, 4, 2, 7
    rng = np.random.default_rng(11)
    params = np.array([1.0, 1.0, 0.3])

    def f_fx_fu_fn(X, U):
        return dyn.unicycle(X, U, params)

    base = dict(f_fx_fu_fn=f_fx_fu_fn, Q=np.tile(np.eye(xdim), (N, 1, 1)), R=np.tile(1e-2 * np.eye(udim), (N, 1, 1)),
                u_l=-np.ones((N, udim)), u_u=np.ones((N, udim)), reg_x=1.0, reg_u=1.0, max_it=6, res_tol=0.0,
                solver_settings=dict(solver="osqp"))
    problems = [dict(base, x0=np.ones(xdim) + 0.3 * rng.standard_normal(xdim), X_ref=0.2 * rng.standard_normal((N, xdim)),
                     U_ref=np.zeros((N, udim))) for _ in range(K)]
    seq = pmpc_amd.solve_problems(problems)
    bat = pmpc_amd.solve_problems(problems, batched=True)
    assert len(bat) == K
    for (Xs, Us, _), (Xb, Ub, _) in zip(seq, bat):
        assert Xs.shape == Xb.shape == (N + 1, xdim)
        assert np.linalg.norm(Xs - Xb) / np.linalg.norm(Xs) < 1e-7 and np.linalg.norm(Us - Ub) / max(np.linalg.norm(Us), 1.0) < 1e-7


@pytest.mark.parametrize("mode", ["torch_callable", "builtin_model", "cone"])
def test_device_resident_scp_loop_matches_host_loop(mode):
    """`solve(..., device="cuda")` (pmpc_amd/scp_device.py: torch tensors in HBM end to end) against the host loop
    (numpy callable through the C ABI): same hist rows and trajectories, consensus problem with M = 6 particles."""
    import torch

    import pmpc_amd
    from pmpc_amd import dynamics as dyn

    M, N, xdim, udim = 6, 20, 4, 2
    rng = np.random.default_rng(5)
    P = np.stack([1.0 + 0.1 * rng.standard_normal(M), 1.0 + 0.1 * rng.standard_normal(M), np.full(M, 0.3)], -1)
    Pt = torch.tensor(P, device="cuda")
    Q, R = np.tile(np.eye(xdim), (M, N, 1, 1)), np.tile(1e-2 * np.eye(udim), (M, N, 1, 1))
    x0 = 1.0 + 0.05 * rng.standard_normal((M, xdim))
    kw = dict(u_l=-np.ones((M, N, udim)), u_u=np.ones((M, N, udim)), reg_x=1.0, reg_u=1.0, max_it=8, res_tol=0.0, verbose=False,
              solver_settings=dict(solver="ecos" if mode == "cone" else "osqp", Nc=2))
    Xh, Uh, dh = pmpc_amd.solve(lambda X, U: dyn.unicycle(X, U, P[:, None, :]), Q, R, x0, **kw)
    if mode == "builtin_model":
        Xd, Ud, dd = pmpc_amd.solve(None, Q, R, x0, device="cuda", builtin_model="unicycle", params=P, **kw)
    else:
        Xd, Ud, dd = pmpc_amd.solve(lambda X, U: dyn.unicycle_torch(X, U, Pt[:, None, :]), Q, R, x0, device="cuda", **kw)
    assert Xd.shape == (M, N + 1, xdim) and len(dd["hist"]) == len(dh["hist"]) == 8
    # the two loops evaluate the unicycle with different sin/cos implementations (numpy / torch-ROCm / HIP); its closed
    # form divides an O(u2^2) difference by u2^2 with u2 ~ eps at U = 0, which amplifies those last-bit differences to
    # ~1e-6 per linearisation — hence 1e-4 here (layout or plumbing mistakes would show up as O(1))
    assert np.linalg.norm(Xd - Xh) / np.linalg.norm(Xh) < 1e-4 and np.linalg.norm(Ud - Uh) / np.linalg.norm(Uh) < 1e-4
    for a, b in zip(dd["hist"], dh["hist"]):
        assert abs(a["obj"] - b["obj"]) <= 1e-4 * abs(b["obj"]) and abs(a["resid"] - b["resid"]) <= 1e-3 * max(b["resid"], 1e-3)


def test_device_loop_with_thrust_cones():
    """`solve(..., device="cuda", builtin_model="quadrotor", soc=...)`: the SCP loop with the thrust cone on every stage runs,
    keeps every iterate inside the cones and boxes, and its residual falls."""
    import pmpc_amd
    from pmpc_amd import dynamics as dyn

    M, N = 16, 30
    prob = dyn.make_quadrotor_problem(M=M, N=N)
    W = np.zeros((2, 4)); W[0, 1] = W[1, 2] = 1.0
    params = prob["params"]
    X, U, data = pmpc_amd.solve(None, prob["Q"], prob["R"], prob["x0"], X_ref=prob["X_ref"], U_ref=prob["U_ref"], X_prev=prob["X_prev"],
                                U_prev=prob["U_prev"], u_l=prob["u_l"], u_u=prob["u_u"], reg_x=prob["reg_x"], reg_u=prob["reg_u"], max_it=6,
                                res_tol=0.0, verbose=False, solver_settings=dict(solver="osqp", Nc=1), device="cuda",
                                builtin_model="quadrotor", params=params,
                                soc=dict(W=W, w0=np.zeros(2), v=[0.3, 0, 0, 0], v0=0.0, u_interior=[9.81, 0, 0, 0]))
    # the same cone handed over in the reference's `extra_cstrs` tuple format gives the same trajectories
    from pmpc_amd.extra_cstrs import stage_soc_to_extra_cstrs

    cstr = stage_soc_to_extra_cstrs(W, np.zeros(2), [0.3, 0, 0, 0], 0.0, M, N, 12, 4, 1)
    X2, U2, _ = pmpc_amd.solve(None, prob["Q"], prob["R"], prob["x0"], X_ref=prob["X_ref"], U_ref=prob["U_ref"], X_prev=prob["X_prev"],
                               U_prev=prob["U_prev"], u_l=prob["u_l"], u_u=prob["u_u"], reg_x=prob["reg_x"], reg_u=prob["reg_u"], max_it=6,
                               res_tol=0.0, verbose=False, device="cuda", builtin_model="quadrotor", params=params,
                               solver_settings=dict(solver="osqp", Nc=1, extra_cstrs=[cstr], soc_u_interior=[9.81, 0, 0, 0]))
    assert np.allclose(X2, X, rtol=0, atol=1e-6) and np.allclose(U2, U, rtol=0, atol=1e-6)  # (warm-start histories differ: not bitwise)
    assert X.shape == (M, N + 1, 12) and len(data["hist"]) == 6
    assert (0.3 * U[..., 0] - np.linalg.norm(U[..., 1:3], axis=-1)).min() > -1e-9
    assert np.all(U >= prob["u_l"] - 1e-9) and np.all(U <= prob["u_u"] + 1e-9) and np.all(U[:, 0] == U[0:1, 0])
    assert data["hist"][-1]["resid"] < data["hist"][0]["resid"]


@pytest.mark.parametrize("name", ["ref_root_testing_single", "ref_root_testing_consensus", "ref_logbarrier_tests"])
def test_scp_on_gpu_reproduces_reference_tables_on_consensus_slew_and_smoothing(name):
    """The HIP back end (`c_lcone_solve` through `pmpc_amd.solve`) against the tables the reference's Julia stack printed in
    tests/root_testing.ipynb (M = 1: slew + log barrier; M = 20: consensus Nc = 5 + slew + log barrier + eps-anchored
    particle weights) and tests/logbarrier_tests.ipynb — row by row, 4 printed digits (tolerances in notebook_problem)."""
    import pmpc_amd
    from tests.support import notebook_problem as nbp

    args, kw, settings, table = nbp.load_table(name)
    X, U, data = pmpc_amd.solve(*args, solver_settings=settings, **kw)
    assert X is not None, "solver failed"
    nbp.check_table(name, data["hist"], table)


@pytest.mark.parametrize("name", ["ref_experimental_cpu", "ref_demo_cost_convex", "ref_demo_cost_external"])
def test_scp_on_gpu_reaches_the_fixed_points_of_the_remaining_reference_tables(name):
    import pmpc_amd
    from tests.support import notebook_problem as nbp

    args, kw, settings, table = nbp.load_table(name)
    if name == "ref_experimental_cpu":
        kw["max_it"] = 150
    X, U, data = pmpc_amd.solve(*args, solver_settings=settings, **kw)
    assert X is not None, "solver failed"
    nbp.check_fixed_point(name, data["hist"], table)
