"""SCP host loop — the counterpart of the reference's pmpc/scp_mpc.py.

`solve` / `scp_solve` keep the reference's call surface (pmpc/scp_mpc.py:205-456, argument glossary
README.md:174-249): per iteration  linearise (user `f_fx_fu_fn`) -> convex sub-problem (`aff_solve`,
here the HIP back end) -> residual / objective row, until `resid < res_tol`, `max_it` or the time
limit.  Returned `data` carries the same keys (`hist`, `solver_data`, `t_aff_solve`, optionally
`sol_hist`, `min_viol_sol`).
"""
from __future__ import annotations

import math
import time
from copy import copy
from typing import Any, Callable, Dict, List, Optional, Tuple

import numpy as np

from .backend import aff_solve as _backend_aff_solve
from .utils import TablePrinter, atleast_nd, to_numpy_f64

print_fn = print


def aff_solve(*args, **kw):
    """Module-level hook (as pmpc/scp_mpc.py:78 — `scp_solve` looks it up at call time, :370)."""
    return _backend_aff_solve(*args, **kw)


# ---- fixed-point filters (pmpc/scp_mpc.py:37-62) ----------------------------------------------------
def _stack(Fs: List[np.ndarray]) -> np.ndarray:
    return np.stack([np.reshape(f, -1) for f in Fs], -1)


def AA_method(Fs: List[np.ndarray]) -> np.ndarray:
    """Anderson-acceleration weights over the residual window."""
    F = _stack(Fs)
    dF = F[:, :-1] - F[:, -1:]
    theta = np.linalg.solve(dF.T @ dF + 1e-10 * np.eye(dF.shape[-1]), -dF.T @ F[:, -1:]).reshape(-1)
    return np.concatenate([theta, [1.0 - np.sum(theta)]])


def smooth_method(Fs: List[np.ndarray]) -> np.ndarray:
    k = len(Fs)
    return np.full(k, 1.0 / k)


def select_method(Fs: List[np.ndarray]) -> np.ndarray:
    """min sum_i alf_i^2 |F_i|^2  s.t. sum alf = 1."""
    F = _stack(Fs)
    k = F.shape[-1]
    A = np.zeros((k + 1, k + 1))
    A[:k, :k] = np.diag(np.linalg.norm(F, axis=0) ** 2)
    A[:k, k] = 1.0
    A[k, :k] = 1.0
    b = np.zeros(k + 1)
    b[k] = 1.0
    return np.linalg.solve(A, b)[:k]


FILTER_MAP = dict(smooth=smooth_method, select=select_method, AA=AA_method)


# ---- linear cost augmentation (pmpc/scp_mpc.py:171-185) ---------------------------------------------
def _augment_cost(lin_cost_fn, X_prev, U_prev, Q, R, X_ref, U_ref, problems):
    if lin_cost_fn is None:
        return X_ref, U_ref
    cx, cu = lin_cost_fn(X_prev, U_prev, problems)
    if cx is not None:
        X_ref = X_ref - np.linalg.solve(Q, np.array(cx)[..., None])[..., 0]
    if cu is not None:
        U_ref = U_ref - np.linalg.solve(R, np.array(cu)[..., None])[..., 0]
    return X_ref, U_ref


def _bmv(A, x):
    return (A @ x[..., None])[..., 0]


def scp_solve(
    f_fx_fu_fn: Callable,
    Q: np.ndarray,
    R: np.ndarray,
    x0: np.ndarray,
    X_ref: Optional[np.ndarray] = None,
    U_ref: Optional[np.ndarray] = None,
    X_prev: Optional[np.ndarray] = None,
    U_prev: Optional[np.ndarray] = None,
    x_l: Optional[np.ndarray] = None,
    x_u: Optional[np.ndarray] = None,
    u_l: Optional[np.ndarray] = None,
    u_u: Optional[np.ndarray] = None,
    verbose: bool = False,
    debug: bool = False,
    max_it: int = 100,
    time_limit: float = 1000.0,
    res_tol: float = 1e-5,
    reg_x: float = 1e0,
    reg_u: float = 1e-2,
    slew_rate: float = 0.0,
    u0_slew: Optional[np.ndarray] = None,
    lin_cost_fn: Optional[Callable] = None,
    cost_fn: Optional[Callable] = None,
    extra_cstrs_fns: Optional[Callable] = None,
    solver_settings: Optional[Dict[str, Any]] = None,
    solver_state: Optional[Dict[str, Any]] = None,
    filter_method: str = "",
    filter_window: int = 5,
    filter_it0: int = 20,
    return_min_viol: bool = False,
    min_viol_it0: int = -1,
    **extra_kw,
) -> Tuple[Optional[np.ndarray], Optional[np.ndarray], Optional[Dict[str, Any]]]:
    """SCP solution of a nonlinear-dynamics / quadratic-cost control problem
    (semantics of pmpc/scp_mpc.py:205-442; defaults identical).  `device="cuda"` (extra keyword, cf. the `device` option
    of the reference's pmpc/experimental solver) runs the whole loop on the GPU with torch tensors: pmpc_amd/scp_device.py."""
    if extra_kw.get("device") is not None:
        from .scp_device import scp_solve_device

        return scp_solve_device(f_fx_fu_fn, Q, R, x0, X_ref=X_ref, U_ref=U_ref, X_prev=X_prev, U_prev=U_prev, x_l=x_l, x_u=x_u,
                                u_l=u_l, u_u=u_u, verbose=verbose, debug=debug, max_it=max_it, time_limit=time_limit, res_tol=res_tol,
                                reg_x=reg_x, reg_u=reg_u, slew_rate=slew_rate, u0_slew=u0_slew, lin_cost_fn=lin_cost_fn,
                                cost_fn=cost_fn, extra_cstrs_fns=extra_cstrs_fns, solver_settings=solver_settings,
                                solver_state=solver_state, filter_method=filter_method, return_min_viol=return_min_viol, **extra_kw)
    if cost_fn is not None:
        raise ValueError("cost_fn is deprecated, use lin_cost_fn instead.")
    t_start = time.time()

    x0, reg_x, reg_u = np.array(to_numpy_f64(x0)), float(reg_x), float(reg_u)
    Q, R = np.array(to_numpy_f64(Q)), np.array(to_numpy_f64(R))
    single = x0.ndim == 1  # pmpc/scp_mpc.py:297-309
    if single:
        assert R.ndim == 3 and Q.ndim == 3
        opt = lambda z: None if z is None else np.asarray(z)
        Q, R, x0 = atleast_nd(Q, 4), atleast_nd(R, 4), atleast_nd(x0, 2)
        X_ref, U_ref, X_prev, U_prev, x_l, x_u, u_l, u_u = [atleast_nd(opt(z), 3) for z in
                                                           (X_ref, U_ref, X_prev, U_prev, x_l, x_u, u_l, u_u)]
    else:
        assert x0.ndim == 2 and R.ndim == 4 and Q.ndim == 4
    M, N, xdim, udim = Q.shape[:3] + R.shape[-1:]

    # references default to zero, the previous iterate to the references (pmpc/scp_mpc.py:311-321); everything in (M, N, d)
    def batch(z, d, fallback):
        return fallback if z is None else np.array(to_numpy_f64(z)).reshape((M, N, d))

    X_ref, U_ref = batch(X_ref, xdim, np.zeros((M, N, xdim))), batch(U_ref, udim, np.zeros((M, N, udim)))
    X_prev, U_prev = batch(X_prev, xdim, X_ref), batch(U_prev, udim, U_ref)
    no_box = np.zeros((0, 0, 0))  # "absent" as the back end understands it (size 0 -> NaN sentinels at the ABI)
    x_l, x_u, u_l, u_u = (no_box if z is None else np.array(z) for z in (x_l, x_u, u_l, u_u))
    if slew_rate is not None:
        slew_rate = float(slew_rate)
    if u0_slew is not None:
        u0_slew = np.array(u0_slew)

    data: Dict[str, Any] = dict(solver_data=[], hist=[], sol_hist=[])
    field_names = ["it", "elaps", "obj", "resid", "reg_x", "reg_u"]
    tp = TablePrinter(field_names, fmts=["%04d"] + ["%8.3e"] * 5)
    solver_settings = copy(solver_settings) if solver_settings is not None else dict()
    Fs: List[np.ndarray] = []
    min_viol, max_res = math.inf, math.inf

    if verbose:
        print_fn(tp.make_header())
    it = 0
    X = U = None
    while it < max_it:
        # -- linearise about [x0, X_prev[:-1]], U_prev (:338-342) ----------------------------------
        X_lin = np.concatenate([x0[..., None, :], X_prev[..., :-1, :]], -2)
        f, fx, fu = f_fx_fu_fn(X_lin, U_prev)
        f = to_numpy_f64(f).reshape((M, N, xdim))
        fx = to_numpy_f64(fx).reshape((M, N, xdim, xdim))
        fu = to_numpy_f64(fu).reshape((M, N, xdim, udim))

        problems = dict(extra_kw, f_fx_fu_fn=f_fx_fu_fn, f=f, fx=fx, fu=fu, x0=x0, X_prev=X_prev, U_prev=U_prev,
                        slew_rate=slew_rate, u0_slew=u0_slew, x_l=x_l, x_u=x_u, u_l=u_l, u_u=u_u, Q=Q, R=R,
                        X_ref=X_ref, U_ref=U_ref)
        X_ref_, U_ref_ = _augment_cost(lin_cost_fn, X_prev, U_prev, Q, R, X_ref, U_ref, problems)
        # user cones: re-evaluated about the current iterate when given as a callback; numpy / scipy members become plain
        # nested lists, the form the pyjulia transport of the reference needs (:353-361) and the back end here accepts
        cstrs = solver_settings.get("extra_cstrs") if extra_cstrs_fns is None else extra_cstrs_fns(X_prev, U_prev, problems)
        if cstrs is not None:
            solver_settings["extra_cstrs"] = tuple([m.tolist() if hasattr(m, "tolist") else m for m in c] for c in cstrs)
        solver_settings["solver_state"] = solver_state

        # -- convex sub-problem (:369-371) -----------------------------------------------------------
        t_aff = time.time()
        X, U, solver_data = aff_solve(f, fx, fu, x0, X_prev, U_prev, Q, R, X_ref_, U_ref_, reg_x, reg_u, slew_rate,
                                      u0_slew, x_l, x_u, u_l, u_u, solver_settings=solver_settings)
        t_aff = time.time() - t_aff
        solver_state = solver_data.get("solver_state", None)
        X, U = X.reshape((M, N + 1, xdim)), U.reshape((M, N, udim))
        if debug or filter_method != "":
            data["sol_hist"].append((X, U))

        # -- optional filtering over the last `filter_window` iterates (:380-387) --------------------
        if filter_method != "":
            X_base = np.concatenate([x0[..., None, :], X_prev], -2)
            Fs.append(np.concatenate([(X - X_base).reshape(-1), (U - U_prev).reshape(-1)]))
            if it >= filter_it0:
                k = min(filter_window, len(Fs))
                alfs = FILTER_MAP[filter_method](Fs[-k:])
                window = data["sol_hist"][-k:]
                X = sum(alf * Xw for alf, (Xw, _) in zip(alfs, window))
                U = sum(alf * Uw for alf, (_, Uw) in zip(alfs, window))

        if np.any(np.isnan(X)) or np.any(np.isnan(U)):  # solver failure (:391-394)
            if verbose:
                print_fn("Solver failed...")
            return None, None, None

        Xn = X[..., 1:, :]
        if filter_method != "":
            dX, dU = data["sol_hist"][-1][0][..., 1:, :] - X_prev, data["sol_hist"][-1][1] - U_prev
        else:
            dX, dU = Xn - X_prev, U - U_prev
        max_res = max(np.max(np.linalg.norm(dX, 2, -1)), np.max(np.linalg.norm(dU, 2, -1)))  # (:397-403)
        eX, eU = Xn - X_ref, U - U_ref
        obj = (np.sum(eX * _bmv(Q, eX)) + np.sum(eU * _bmv(R, eU))) / N / M  # (:404-405)
        X_prev, U_prev = Xn, U

        vals = (it + 1, time.time() - t_start, obj, max_res, reg_x, reg_u)
        if verbose:
            print_fn(tp.make_values(vals))
        data["solver_data"].append(solver_data)
        data["hist"].append(dict(zip(field_names, vals)))
        data.setdefault("t_aff_solve", []).append(t_aff)
        if return_min_viol and (it >= min_viol_it0 or min_viol_it0 < 0) and min_viol > max_res:
            data["min_viol_sol"], min_viol = (X, U), max_res

        if max_res < res_tol:
            break
        it += 1
        if (time.time() - t_start) * (it + 1) / it > time_limit:
            break

    if verbose:
        print_fn(tp.make_footer())
        if max_res > 1e-2:
            print_fn("#" * 73)
            print_fn("Bad solution found, the solution is approximate to a residual:", "%9.4e" % max_res)
            print_fn("#" * 73)
    if not debug:
        del data["sol_hist"]
    if single:
        return X.reshape((N + 1, xdim)), U.reshape((N, udim)), data
    return X.reshape((M, N + 1, xdim)), U.reshape((M, N, udim)), data


def solve(*args, **kwargs):
    """pmpc/scp_mpc.py:446-456 (`profile=True` needs line_profiler, as upstream)."""
    if kwargs.pop("profile", False):
        from line_profiler import LineProfiler

        lp = LineProfiler()
        lp.add_function(scp_solve)
        ret = lp.wrap_function(scp_solve)(*args, **kwargs)
        lp.print_stats(output_unit=1e-3)
        return ret
    return scp_solve(*args, **kwargs)


_BATCH_ARRAY_KEYS = ("Q", "R", "x0", "X_ref", "U_ref", "X_prev", "U_prev", "x_l", "x_u", "u_l", "u_u", "u_slew")


def solve_problems(problems: List[Dict[str, Any]], verbose: bool = False, batched: bool = False, split: bool = True, **kw):
    """pmpc/scp_mpc.py:504-511: one `solve` per problem dictionary.

    `batched=True` is the independent-problem mode the reference offers through other means
    (pmpc/experimental/remote_like_interface.py:35-106 stacks the list and runs ONE `scp_solve`; pmpc/remote.py farms the
    list out): the problems — same dimensions, the first one's batch-capable `f_fx_fu_fn`, scalar options of the first —
    are stacked along the particle axis with consensus horizon `Nc = 0`, so every SCP iteration is one launch of the same
    kernels over all of them.  All problems then take the same number of SCP iterations (the loop stops on the largest
    residual).  `split=True` returns a list of `(X, U, data)`, else the stacked arrays."""
    if not batched:
        return [solve(**dict(p, verbose=verbose)) for p in problems]
    assert len(problems) > 0
    first = problems[0]
    stacked: Dict[str, Any] = {k: v for k, v in first.items() if k not in _BATCH_ARRAY_KEYS}
    for k in _BATCH_ARRAY_KEYS:
        present = [k in p and p[k] is not None for p in problems]
        if any(present):
            assert all(present), f"`{k}` must be given for every problem or for none"
            stacked[k] = np.stack([to_numpy_f64(p[k]) for p in problems], 0)
    stacked["solver_settings"] = dict(copy(first.get("solver_settings") or {}), Nc=0)
    stacked["verbose"] = verbose
    f_fx_fu_fn, Q, R, x0 = (stacked.pop(k) for k in ("f_fx_fu_fn", "Q", "R", "x0"))
    X, U, data = solve(f_fx_fu_fn, Q, R, x0, **stacked, **kw)
    if not split or X is None:
        return X, U, data
    return [(X[i], U[i], data) for i in range(X.shape[0])]


def tune_scp(*args, sample_nb: int = 14, reg_rng: Tuple[int, int] = (-3, 3), solve_fn: Callable = scp_solve,
             savefig: Optional[str] = None, **kwargs):
    """Sweep `reg_x` over `logspace(*reg_rng, sample_nb)` with `reg_u = reg_ratio * reg_x` and return the pair with
    the smallest final SCP residual (pmpc/scp_mpc.py:460-497).  The residual curve is plotted only when matplotlib
    is importable (upstream requires it)."""
    reg_ratio = kwargs.pop("reg_ratio", 1e-1)
    reg_list = kwargs.pop("reg_list", np.logspace(*reg_rng, sample_nb))
    res_list = []
    for reg in reg_list:
        kwargs["reg_x"], kwargs["reg_u"] = reg, reg * reg_ratio
        kwargs["verbose"] = False
        X, U, data = solve_fn(*args, **kwargs)
        res_list.append(1e2 if data is None else data["hist"][-1]["resid"])
    try:
        import matplotlib.pyplot as plt
    except ImportError:
        plt = None
    if plt is not None:
        plt.figure()
        plt.loglog(reg_list, res_list)
        plt.ylabel("final residual"), plt.xlabel("reg_x"), plt.title("reg_u = reg_x * %6.1e" % reg_ratio)
        plt.tight_layout()
        plt.grid(True, which="both")
        if savefig is not None:
            plt.savefig(savefig, dpi=200)
    reg_x = float(reg_list[int(np.argmin(res_list))])
    return reg_x, reg_ratio * reg_x
