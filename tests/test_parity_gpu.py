"""GPU parity tests proper: libpmpc_hip.so (through the C ABI) against the oracle on the same seeded
inputs.  Tolerance (BASELINE.json north_star): fp64, relative error
|X - X*| / |X*| and |U - U*| / max(|U*|, 1) <= 1e-6 (we assert 1e-7; typical is 1e-10)."""
import numpy as np
import pytest

from tests.support.problems import CASES, abi_args, rand_problem

pytestmark = pytest.mark.gpu
TOL = 1e-7


def _rel(a, b, floor=1e-300):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), floor)


@pytest.mark.parametrize("case", CASES, ids=[str(c) for c in CASES])
def test_lqp_solve_matches_oracle(case, oracle):
    from pmpc_amd import backend

    M, N, x, u, Nc, bu, bx, sl, sl0 = case
    rng = np.random.default_rng(1000 + CASES.index(case))
    args, kw = rand_problem(rng, M, N, x, u, bu, bx, sl, sl0)
    Xo, Uo = oracle.lqp_solve_py(*args, Nc=Nc, **kw)
    X, U = backend.lqp_solve(*abi_args(args, kw, Nc))
    assert X.shape == (M, N, x) and U.shape == (M, N, u)
    assert _rel(X, Xo) <= TOL, _rel(X, Xo)
    assert _rel(U, Uo, 1.0) <= TOL, _rel(U, Uo, 1.0)
    if Nc != 0 and M > 1:  # consensus: the first Nc controls are shared exactly (examples/simple_demo.ipynb:402-421)
        k = N if Nc < 0 else Nc
        assert np.all(U[:, :k] == U[0:1, :k])


@pytest.mark.parametrize("case", [CASES[2], CASES[13], CASES[17]], ids=lambda c: str(c))
@pytest.mark.parametrize("symmetric", [True, False])
def test_column_major_and_row_major_entries_agree(case, symmetric, oracle):
    """`c_lqp_solve` proper (Fortran-contiguous arrays, the layout of c_interface.jl:28-46) and `pmpc_lqp_solve_host`
    (row-major Jacobian / cost blocks, transposed in HBM) against the oracle, with cost blocks that are exactly symmetric
    or not (then OSQP's triu(P) semantics apply and the transposition of Q, R matters)."""
    from pmpc_amd import backend

    M, N, x, u, Nc, bu, bx, sl, sl0 = case
    rng = np.random.default_rng(77)
    args, kw = rand_problem(rng, M, N, x, u, bu, bx, sl, sl0)
    args = list(args)
    if symmetric:
        args[6], args[7] = [0.5 * (a + np.swapaxes(a, -1, -2)) for a in (args[6], args[7])]
    else:
        args[6] = args[6] + 0.05 * np.triu(rng.standard_normal((M, N, x, x)), 1)
        args[7] = args[7] + 0.05 * np.triu(rng.standard_normal((M, N, u, u)), 1)
    Xo, Uo = oracle.lqp_solve_py(*args, Nc=Nc, **kw)
    a = abi_args(tuple(args), kw, Nc)
    assert not a[3].flags.f_contiguous  # py2jl view of the C-ordered fx stack -> row-major entry point
    fa = tuple(np.asfortranarray(v) if isinstance(v, np.ndarray) else v for v in a)
    for arrs in (a, fa):
        X, U = backend.lqp_solve(*arrs)
        assert _rel(X, Xo) <= TOL, _rel(X, Xo)
        assert _rel(U, Uo, 1.0) <= TOL, _rel(U, Uo, 1.0)


_IPM_ONLY_SCRIPT = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from oracle import lqp_oracle as orc
from pmpc_amd import backend
from tests.support.problems import abi_args, rand_problem
worst = 0.0
for k, (M, N, x, u, Nc, bu) in enumerate([(8, 10, 12, 4, 1, 0.4), (6, 8, 5, 3, -1, 0.3), (8, 7, 4, 2, 3, 0.3), (5, 9, 13, 2, 0, 0.2), (1, 30, 2, 1, 1, 0.4)]):
    args, kw = rand_problem(np.random.default_rng(900 + k), M, N, x, u, bu)
    Xo, Uo = orc.lqp_solve_py(*args, Nc=Nc, **kw)
    for rep in range(2):  # the second call is warm-started (interior-point warm start only: the active-set paths are off)
        X, U = backend.lqp_solve(*abi_args(args, kw, Nc))
        e = max(np.linalg.norm(X - Xo) / np.linalg.norm(Xo), np.linalg.norm(U - Uo) / max(np.linalg.norm(Uo), 1.0))
        worst = max(worst, e)
        print((M, N, x, u, Nc), rep, e, flush=True)
assert worst < 1e-7, worst
print("IPM_ONLY_OK")
"""


def test_interior_point_path_alone_still_matches_the_oracle():
    """With PMPC_POLISH_MU=0 every active-set use is off and control boxes go through the Mehrotra interior-point
    iteration to mu = 1e-12 again — the fallback of the active-set iteration must stay parity-green on its own.  Own
    process: the switch is read once per process."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    env = dict(os.environ, PMPC_POLISH_MU="0")
    r = subprocess.run([sys.executable, "-c", _IPM_ONLY_SCRIPT], cwd=str(Path(__file__).resolve().parents[1]), env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "IPM_ONLY_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_empty_box_fails_like_the_reference(oracle):
    """lo > hi on one control: no feasible point.  The reference's OSQP reports infeasibility and the outputs are NaN
    (osqp_solver.jl:65-71); the active-set rounds must not 'solve' it by holding the control on one side."""
    from pmpc_amd import backend

    args, kw = rand_problem(np.random.default_rng(5), 4, 6, 4, 2, 0.3)
    kw["u_l"] = kw["u_l"].copy()
    kw["u_l"][1, 2, 0] = kw["u_u"][1, 2, 0] + 0.5
    X, U = backend.lqp_solve(*abi_args(args, kw, 1))
    assert np.isnan(X).all() and np.isnan(U).all()
    args2, kw2 = rand_problem(np.random.default_rng(5), 4, 6, 4, 2, 0.3)  # and the context recovers
    Xo, Uo = oracle.lqp_solve_py(*args2, Nc=1, **kw2)
    X, U = backend.lqp_solve(*abi_args(args2, kw2, 1))
    assert np.linalg.norm(X - Xo) / np.linalg.norm(Xo) < 1e-7 and np.linalg.norm(U - Uo) / max(np.linalg.norm(Uo), 1.0) < 1e-7


def test_host_abi_notices_in_place_changes_of_arrays_it_did_not_resend(oracle):
    """The host-pointer ABI skips the upload of chunks whose bytes equal the previous call's (Q, R, references, boxes inside
    an SCP loop).  The comparison is exact: a single entry changed IN PLACE in the caller's own array must reach the solve."""
    from pmpc_amd import backend

    args, kw = rand_problem(np.random.default_rng(8), 6, 8, 4, 2, 0.3)
    rel = lambda a, b: np.linalg.norm(a - b) / max(np.linalg.norm(b), 1.0)
    a = list(abi_args(args, kw, 1))
    for step in range(4):
        if step == 1:  # a new linearisation, everything else untouched (the SCP case: those chunks are not sent again)
            a[2] = a[2] + 0.01
            args = (args[0], args[1] + 0.01) + tuple(args[2:])
        if step == 2:  # one cost entry changed in place, same array object, same address
            Q = args[6]
            Q[2, 3, 1, 1] += 0.5
            a[7][...] = abi_args(args, kw, 1)[7]
        if step == 3:  # and one bound
            kw["u_u"][1, 2, 0] *= 0.5
            a[14][...] = abi_args(args, kw, 1)[14]
        Xo, Uo = oracle.lqp_solve_py(*args, Nc=1, **kw)
        X, U = backend.lqp_solve(*a)
        assert rel(X, Xo) < 1e-7 and rel(U, Uo) < 1e-7, step


# (M, N, x, u, Nc, u-bound): the dense consensus system has Nc u unknowns — 17 .. 255 of them go through k_cons_solve_reg
DENSE_CONS = [(3, 9, 4, 2, -1, 0.3), (3, 16, 4, 2, -1, 0.3), (3, 17, 4, 2, -1, 0.3), (2, 27, 5, 3, -1, 0.4), (3, 40, 4, 2, 33, 0.3),
              (2, 50, 12, 4, -1, 0.4), (2, 52, 4, 2, -1, 0.3), (2, 64, 2, 1, -1, 0.5), (2, 63, 12, 4, -1, 0.5), (2, 70, 12, 4, 64, 0.5)]


@pytest.mark.parametrize("case", DENSE_CONS, ids=[f"nc={(c[1] if c[4] < 0 else c[4]) * c[3]}-{c}" for c in DENSE_CONS])
def test_dense_consensus_systems_of_every_panel_count(case, oracle):
    """Several shared stages with boxes tight enough that shared controls are held (1e30 on the diagonal of the consensus system): 18 .. 256
    unknowns — whole and ragged last panels, both instantiations of the register-resident solve (up to 12 / 17 blocks per wave), and one size
    beyond it (256: the blocked factorisation in global memory) — against the oracle; cold and warm (the warm solve re-factors with the
    accepted set held)."""
    from pmpc_amd import backend

    M, N, x, u, Nc, bu = case
    args, kw = rand_problem(np.random.default_rng(7000 + N * u + M), M, N, x, u, bu)
    Xo, Uo = oracle.lqp_solve_py(*args, Nc=Nc, **kw)
    k = N if Nc < 0 else Nc
    assert np.any(np.abs(np.abs(Uo[0, :k]) - bu) <= 1e-9), "no shared control on its bound: tighten the box"
    for rep in ("cold", "warm"):
        X, U = backend.lqp_solve(*abi_args(args, kw, Nc))
        assert _rel(X, Xo) <= TOL and _rel(U, Uo, 1.0) <= TOL, (rep, _rel(X, Xo), _rel(U, Uo, 1.0))
        assert np.all(U[:, :k] == U[0:1, :k])
