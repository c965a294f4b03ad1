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
