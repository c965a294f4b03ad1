"""pmpc_amd — MI355X-native back end for particle SCP-MPC with the call surface of StanfordASL/pmpc.

    from pmpc_amd import solve
    X, U, data = solve(f_fx_fu_fn, Q, R, x0, X_ref=..., U_ref=..., u_l=..., u_u=..., solver_settings=dict(Nc=1))

The convex sub-problem of every SCP iteration is solved by hand-written HIP kernels behind the
reference's own C ABI (`c_lqp_solve` / `c_lcone_solve`, include/pmpc_abi.h).  There is no CPU path.
"""
from .scp_mpc import AA_method, FILTER_MAP, aff_solve, scp_solve, select_method, smooth_method, solve, solve_problems, tune_scp  # noqa: F401
from .backend import is_precompiled_backend_available, lcone_solve, lqp_solve  # noqa: F401
from .problem_struct import Problem  # noqa: F401
from .problem_matrices import lqp_generate_problem_matrices  # noqa: F401


def scp_solve_device(*args, **kw):
    """Device-resident SCP loop (torch tensors in HBM end to end); see pmpc_amd/scp_device.py."""
    from .scp_device import scp_solve_device as _impl

    return _impl(*args, **kw)

# keyword-compatible arguments of `solve` (pmpc/__init__.py:5-31)
SOLVE_KWS = {
    "X_ref", "U_ref", "X_prev", "U_prev", "x_l", "x_u", "u_l", "u_u", "verbose", "debug", "max_it", "time_limit",
    "res_tol", "reg_x", "reg_u", "slew_rate", "u_slew", "cost_fn", "extra_cstrs_fns", "method", "solver_settings",
    "solver_state", "filter_method", "filter_window", "filter_it0",
}
