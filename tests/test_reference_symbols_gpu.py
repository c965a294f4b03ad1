"""The two symbols the reference's pybind11 module binds (PMPC.jl/pmpcjl/module.cpp:9-23) — `c_lqp_solve` and
`c_lcone_solve(..., smooth_alpha, char *solver)` — called exactly as pmpc/static_backend.py:71-102,:159-189 calls them:
every array Fortran-contiguous in the Julia shapes, nothing through the row-major extension entries.  Checked against the
oracle on seeded problems (hard boxes, log-barrier smoothing, consensus, slew) and against one table the reference's own
Julia + ECOS stack printed (tests/root_testing.ipynb cells 3-4).  Tolerance: fp64, 1e-7 relative (north star: 1e-6)."""
import ctypes

import numpy as np
import pytest

from tests.support.problems import abi_args, rand_problem

pytestmark = pytest.mark.gpu
TOL = 1e-7


def _rel(a, b, floor=1e-300):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), floor)


def _raw_call(symbol, a, extra=()):
    """ctypes straight onto the exported symbol: no pmpc_amd.backend in between."""
    from pmpc_amd import _lib

    lib = _lib.load()
    Nc, arrs = a[0], [np.asfortranarray(np.asarray(v, dtype=np.float64)) for v in a[1:15]]
    reg_x, reg_u = a[15], a[16]
    tail = [np.asfortranarray(np.asarray(v, dtype=np.float64)) for v in a[17:20]]
    assert all(v.flags.f_contiguous for v in arrs + tail)
    xdim, M = arrs[0].shape
    N, udim = arrs[1].shape[1], arrs[3].shape[1]
    X, U = np.empty(xdim * N * M), np.empty(udim * N * M)
    p = lambda v: v.ctypes.data_as(_lib.c_dp)
    getattr(lib, symbol)(p(X), p(U), xdim, udim, N, M, int(Nc), *[p(v) for v in arrs], float(reg_x), float(reg_u), *[p(v) for v in tail],
                         0, *extra)
    return X.reshape(M, N, xdim), U.reshape(M, N, udim)


QP_CASES = [  # (M, N, x, u, Nc, u-bound, x-bound, slew, slew0)
    (5, 9, 12, 4, 1, 0.4, None, None, None),
    (4, 8, 4, 2, -1, 0.2, 30.0, None, None),
    (3, 6, 3, 2, 2, 0.3, 5.0, 0.5, 0.3),
    (6, 10, 4, 2, 0, None, None, None, None),
]


@pytest.mark.parametrize("case", QP_CASES, ids=str)
def test_c_lqp_solve_symbol_with_fortran_arrays(case, oracle):
    M, N, x, u, Nc, bu, bx, sl, sl0 = case
    args, kw = rand_problem(np.random.default_rng(4100 + QP_CASES.index(case)), M, N, x, u, bu, bx, sl, sl0)
    Xo, Uo = oracle.lqp_solve_py(*args, Nc=Nc, **kw)
    X, U = _raw_call("c_lqp_solve", abi_args(args, kw, Nc))
    assert _rel(X, Xo) <= TOL and _rel(U, Uo, 1.0) <= TOL, (_rel(X, Xo), _rel(U, Uo, 1.0))


CONE_CASES = [  # (M, N, x, u, Nc, u-bound, smooth_alpha, solver string)
    (1, 12, 4, 2, 0, 0.3, float("nan"), b"ecos"),
    (24, 8, 4, 2, 1, 0.3, float("nan"), b"ecos"),
    (24, 8, 4, 2, 1, 0.3, 1e2, b"ecos"),
    (6, 10, 12, 4, 2, 0.4, 1e1, b"cosmo"),
    (9, 7, 5, 3, -1, 0.4, float("nan"), b"mosek"),
]


@pytest.mark.parametrize("case", CONE_CASES, ids=str)
def test_c_lcone_solve_symbol_with_fortran_arrays(case, oracle):
    """Hard boxes (smooth_alpha = NaN, main.jl:242-244) and log-barrier smoothing; the `solver` string only selects the
    conic back end upstream (main.jl:320) — every value must give the oracle's optimum."""
    M, N, x, u, Nc, bu, alpha, solver = case
    args, kw = rand_problem(np.random.default_rng(4200 + CONE_CASES.index(case)), M, N, x, u, bu)
    Xo, Uo = oracle.lcone_solve_py(*args, Nc=Nc, smooth_alpha=alpha, **kw)
    X, U = _raw_call("c_lcone_solve", abi_args(args, kw, Nc), extra=(ctypes.c_double(alpha), ctypes.c_char_p(solver)))
    assert _rel(X, Xo) <= 1e-6 and _rel(U, Uo, 1.0) <= 1e-6, (_rel(X, Xo), _rel(U, Uo, 1.0))


def test_reference_table_through_c_lcone_solve_proper():
    """tests/root_testing.ipynb cells 3-4 (M = 1, slew 1e2, log barrier alpha 0.1, solver "ecos"): the whole SCP loop with
    every sub-problem going through the REFERENCE symbol `c_lcone_solve` (Fortran arrays), row by row against the table the
    reference's Julia stack printed."""
    import pmpc_amd
    from pmpc_amd import backend
    from tests.support import notebook_problem as nbp

    args, kw, settings, table = nbp.load_table("ref_root_testing_single")
    backend.REFERENCE_SYMBOLS_ONLY = True
    before = dict(backend.CALLS)
    try:
        X, U, data = pmpc_amd.solve(*args, solver_settings=settings, **kw)
    finally:
        backend.REFERENCE_SYMBOLS_ONLY = False
    assert X is not None, "solver failed"
    nbp.check_table("ref_root_testing_single", data["hist"], table)
    assert backend.CALLS.get("c_lcone_solve", 0) - before.get("c_lcone_solve", 0) == len(table)
    assert backend.CALLS.get("pmpc_lcone_solve_host", 0) == before.get("pmpc_lcone_solve_host", 0)


def test_reference_qp_golden_through_c_lqp_solve_proper():
    """The canonical ABI example of the reference (tests/pmpcjl_test.py:164-219: x2 u1 N=30 Nc=3, u in +-0.4, x in +-20,
    slew_reg = 1) through `c_lqp_solve` itself."""
    from tests.test_oracle_golden import load_qp

    args, kw, Nc, Xg, Ug, _ = load_qp("qp_double_integrator_u04.npz")
    X, U = _raw_call("c_lqp_solve", abi_args(args, kw, Nc))
    assert _rel(X, Xg) <= TOL and _rel(U, Ug, 1.0) <= TOL
