"""CPU tests (no GPU): the oracle against the committed golden fixtures (tests/golden/make_golden.py) and
the numpy model of the device algorithm against the oracle."""
from pathlib import Path

import numpy as np
import pytest

from tests.support.problems import CASES, rand_problem

GOLD = Path(__file__).resolve().parent / "golden"
QP_FILES = sorted(p.name for p in GOLD.glob("qp_*.npz"))
CONE_FILES = sorted(p.name for p in GOLD.glob("cone_*.npz"))
ARG_NAMES = ["x0", "f", "fx", "fu", "X_prev", "U_prev", "Q", "R", "X_ref", "U_ref"]
KW_NAMES = ["reg_x", "reg_u", "u_l", "u_u", "x_l", "x_u", "slew_reg", "slew_reg0", "slew_um1"]


def load_qp(name):
    z = np.load(GOLD / name)
    args = tuple(z[k] for k in ARG_NAMES)
    kw = {k: (float(z[k]) if z[k].ndim == 0 else z[k]) for k in KW_NAMES if k in z.files}
    return args, kw, int(z["Nc"]), z["X"], z["U"], z["cert"]


@pytest.mark.parametrize("name", QP_FILES)
def test_oracle_reproduces_golden(name, oracle):
    args, kw, Nc, X, U, cert = load_qp(name)
    Xo, Uo, info = oracle.lqp_solve_py(*args, Nc=Nc, return_info=True, **kw)
    np.testing.assert_allclose(Xo, X, rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(Uo, U, rtol=1e-9, atol=1e-10)
    assert max(info["cert"].values()) < 1e-8 and cert.max() < 1e-8


@pytest.mark.parametrize("name", CONE_FILES)
def test_cone_oracle_reproduces_golden(name, oracle):
    """cone path: the fixture's minimiser, and the optimality condition of the epigraph problem — the particles that
    are not at the full weight 1 + eps attain min_i J_i (multipliers lambda_i = w_i), deficit 2 eps M in total"""
    args, kw, Nc, X, U, cert = load_qp(name)
    z = np.load(GOLD / name)
    Xo, Uo, info = oracle.lcone_solve_py(*args, Nc=Nc, return_info=True, **kw)
    np.testing.assert_allclose(Xo, X, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(Uo, U, rtol=1e-8, atol=1e-9)
    M, eps = X.shape[0], oracle.COST_ANCHOR_EPS
    w, J = z["weights"], z["J"]
    np.testing.assert_allclose(np.sum(1 + eps - w), 2 * eps * M, rtol=1e-12)
    assert np.all(J[w < 1 + eps - 1e-12] <= J.min() * (1 + 1e-9)) and cert.max() < 1e-8
    np.testing.assert_allclose(oracle.cone_objective(J), float(np.sum(w * J)), rtol=1e-9)


def test_golden_double_integrator_structure():
    """tests/pmpcjl_test.py:164-219: Nc=3 shared controls, |u| <= 0.4 saturates, |x| <= 20 holds."""
    args, kw, Nc, X, U, _ = load_qp("qp_double_integrator_u04.npz")
    assert Nc == 3 and np.all(np.abs(U) <= 0.4 + 1e-9) and np.all(np.abs(X) <= 20 + 1e-9)
    assert np.sum(np.abs(np.abs(U) - 0.4) < 1e-8) >= 5  # the control bound is active early on


def test_golden_consensus_property():
    """examples/simple_demo.ipynb:402-421: first Nc controls identical across particles, then they differ."""
    for name, Nc in (("qp_chain_Nc1.npz", 1), ("qp_chain_Nc3.npz", 3)):
        _, _, nc, _, U, _ = load_qp(name)
        assert nc == Nc and np.all(U[:, :Nc] == U[0:1, :Nc]) and np.max(np.abs(U[1:, Nc:] - U[0:1, Nc:])) > 1e-3


@pytest.mark.parametrize("case", CASES, ids=[str(c) for c in CASES])
def test_structured_model_matches_oracle(case, oracle):
    """The Riccati / consensus-condensing / IPM algorithm the GPU runs, in numpy, against the oracle."""
    from tests.support import structured_np as snp

    M, N, x, u, Nc, bu, bx, sl, sl0 = case
    rng = np.random.default_rng(1000 + CASES.index(case))
    args, kw = rand_problem(rng, M, N, x, u, bu, bx, sl, sl0)
    Xo, Uo = oracle.lqp_solve_py(*args, Nc=Nc, **kw)
    X, U, _ = snp.ipm_solve(snp.Problem(*args[1:], Nc=Nc, **kw))
    assert np.linalg.norm(X - Xo) / np.linalg.norm(Xo) < 1e-7
    assert np.linalg.norm(U - Uo) / max(np.linalg.norm(Uo), 1.0) < 1e-7


def test_oracle_assembly_shapes(oracle):
    """Sizes of the joint QP (lqp_utils.jl:4-15, :223-226): the (120,180)/(180,180) shapes printed at
    tests/jax_solver.ipynb:192-195 for x4 u2 N=30 M=1."""
    rng = np.random.default_rng(0)
    args, kw = rand_problem(rng, 1, 30, 4, 2, bounds_u=1.0)
    x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
    nan = np.full((1, 30, 4), np.nan)
    qp = oracle.assemble_abi(4, 2, 30, 1, -1, f, oracle.to_abi_mat(fx), oracle.to_abi_mat(fu), X_prev, U_prev,
                             oracle.to_abi_mat(Q), oracle.to_abi_mat(R), X_ref, U_ref, nan, nan, kw["u_l"], kw["u_u"], 1.0, 0.1,
                             np.full(1, np.nan), np.full(1, np.nan), np.full((1, 2), np.nan))
    assert qp.A.shape == (120, 180) and qp.P.shape == (180, 180) and qp.G.shape == (60, 180)
    assert qp.A.nnz == 4 * 2 * 30 + 30 * 4 + 29 * 16


@pytest.mark.parametrize("case", [c for c in CASES if c[7] is None and c[8] is None], ids=lambda c: str(c))
def test_structured_cpu_baseline_matches_oracle(case):
    """oracle/structured_cpu.c (the OpenMP baseline bench.py times) solves the same QP as the exact oracle."""
    from oracle import lqp_oracle as orc

    M, N, x, u, Nc, bu, bx, _, _ = case
    args, kw = rand_problem(np.random.default_rng(1000 + CASES.index(case)), M, N, x, u, bu, bx)
    Xo, Uo = orc.lqp_solve_py(*args, Nc=Nc, **kw)
    X, U, info = orc.structured_cpu_solve_py(*args, kw["reg_x"], kw["reg_u"], Nc=Nc, x_l=kw.get("x_l"), x_u=kw.get("x_u"),
                                             u_l=kw.get("u_l"), u_u=kw.get("u_u"), threads=2)
    assert info["status"] == 0
    assert np.linalg.norm(X - Xo) <= 1e-7 * np.linalg.norm(Xo)
    assert np.linalg.norm(U - Uo) <= 1e-7 * max(1.0, np.linalg.norm(Uo))


@pytest.mark.parametrize("case", [(5, 8, 4, 2, 1, 0.3), (3, 9, 5, 3, -1, 0.1), (6, 7, 3, 2, 0, 0.2), (4, 10, 12, 4, 2, 0.4)],
                         ids=["Nc1", "NcN", "Nc0", "quadrotor-dims-Nc2"])
def test_active_set_model_reaches_the_oracle_optimum(case, oracle):
    """The numpy model of the device's primal-dual active-set iteration (penalty `big` on the step of the held controls,
    multipliers = -/+ big du, KKT sign check) ends exactly on the oracle's optimum — cold, and warm-started from the set of a
    perturbed problem (the SCP use)."""
    from tests.support import structured_np as snp
    from tests.support.problems import rand_problem

    M, N, x, u, Nc, bu = case
    rng = np.random.default_rng(31)
    args, kw = rand_problem(rng, M, N, x, u, bu)
    Xo, Uo = oracle.lqp_solve_py(*args, Nc=Nc, **kw)
    p = snp.Problem(*args[1:], Nc=Nc, **kw)
    X, U, info = snp.active_set_solve(p)
    assert np.linalg.norm(X - Xo) / np.linalg.norm(Xo) < 1e-10 and np.linalg.norm(U - Uo) / max(np.linalg.norm(Uo), 1.0) < 1e-10
    assert (info["act"] > 0).any()  # the boxes are active in these cases
    # next sub-problem of an SCP-like sequence: perturbed linearisation, warm start from the previous set and solution
    x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
    args2 = (x0, f + 0.05 * rng.standard_normal(f.shape), fx * (1 + 0.05 * rng.standard_normal(fx.shape)),
             fu * (1 + 0.05 * rng.standard_normal(fu.shape)), X_prev, U_prev, Q, R, X_ref, U_ref)
    Xo2, Uo2 = oracle.lqp_solve_py(*args2, Nc=Nc, **kw)
    p2 = snp.Problem(*args2[1:], Nc=Nc, **kw)
    X2, U2, info2 = snp.active_set_solve(p2, act0=info["act"], U0=U)
    assert np.linalg.norm(X2 - Xo2) / np.linalg.norm(Xo2) < 1e-10 and np.linalg.norm(U2 - Uo2) / max(np.linalg.norm(Uo2), 1.0) < 1e-10
