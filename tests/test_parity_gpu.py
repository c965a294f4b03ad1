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
