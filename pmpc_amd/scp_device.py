"""Device-resident SCP loop: `scp_solve` (pmpc/scp_mpc.py:205-442) with every array kept in HBM.

    X, U, data = pmpc_amd.solve(f_fx_fu_fn, Q, R, x0, ..., device="cuda")

`f_fx_fu_fn(X_, U_prev)` receives float64 torch tensors on the GPU (`X_ = [x0, X_prev[:-1]]`, the reference's
contract, README.md:136-145 / scp_mpc.py:338-342) and returns `f (M,N,x)`, `fx (M,N,x,x)`, `fu (M,N,x,u)` as torch GPU
tensors in the usual (row, col) layout — or already in the ABI layout `(M,N,col,row)` with `jacobians_abi_layout=True`,
which saves one transposing copy of the Jacobian stacks per iteration.  A built-in model (`builtin_model="unicycle" |
"quadrotor"`, `params=...`) is linearised by the HIP kernel of csrc/dynamics.hip instead of a Python callable.
Nothing crosses PCIe inside the loop except the scalars of the `hist` row (one small read per SCP iteration).

`solver_settings["extra_cstrs"]` in the reference's tuple format is accepted for that case (pmpc_amd/extra_cstrs.py).
`soc=dict(W=(q,u), w0=(q,), v=(u,), v0=float, u_interior=(u,))` adds the stage-wise second-order cone
`||W u + w0|| <= v'u + v0` on every stage's controls (thrust cones; `DeviceSolver.lsoc_solve`).

Host-only features of the reference loop that need the sub-problem on the host (`lin_cost_fn`, `extra_cstrs_fns`,
filters, `solver_state`) are not offered here; `pmpc_amd.scp_mpc.scp_solve` (the default) has them.
"""
from __future__ import annotations

import math
import time
from typing import Any, Callable, Dict, Optional

import numpy as np
import torch

from .device import MODEL_QUADROTOR, MODEL_UNICYCLE, DeviceSolver
from .utils import TablePrinter

_MODELS = {"unicycle": MODEL_UNICYCLE, "quadrotor": MODEL_QUADROTOR}
_solvers: Dict[int, DeviceSolver] = {}


def _solver_for(device: torch.device) -> DeviceSolver:
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _solvers:
        _solvers[idx] = DeviceSolver(idx)
    return _solvers[idx]


def scp_solve_device(f_fx_fu_fn: Optional[Callable], Q, R, x0, X_ref=None, U_ref=None, X_prev=None, U_prev=None, x_l=None,
                     x_u=None, u_l=None, u_u=None, verbose: bool = False, max_it: int = 100, time_limit: float = 1000.0,
                     res_tol: float = 1e-5, reg_x: float = 1e0, reg_u: float = 1e-2, slew_rate: Optional[float] = None,
                     u0_slew=None, solver_settings: Optional[Dict[str, Any]] = None, device="cuda",
                     jacobians_abi_layout: bool = False, builtin_model: Optional[str] = None, params=None,
                     return_torch: bool = False, solver: Optional[DeviceSolver] = None, soc: Optional[Dict[str, Any]] = None,
                     lin_cost_fn=None, cost_fn=None,
                     extra_cstrs_fns=None, solver_state=None, filter_method: str = "", debug: bool = False,
                     return_min_viol: bool = False, **ignored):
    host_only = dict(lin_cost_fn=lin_cost_fn, cost_fn=cost_fn, extra_cstrs_fns=extra_cstrs_fns, solver_state=solver_state,
                     filter_method=filter_method or None, debug=debug or None, return_min_viol=return_min_viol or None)
    bad = [k for k, v in host_only.items() if v is not None]
    if bad:
        raise ValueError(f"scp_solve(device=...) does not support {bad}; use the host loop (device=None)")
    dev = torch.device(device)
    t_start = time.time()
    T = lambda z: None if z is None else torch.as_tensor(np.asarray(z) if not torch.is_tensor(z) else z, dtype=torch.float64, device=dev)
    Q, R, x0 = T(Q), T(R), T(x0)
    single = x0.ndim == 1  # pmpc/scp_mpc.py:297-309
    if single:
        assert Q.ndim == 3 and R.ndim == 3
        Q, R, x0 = Q[None], R[None], x0[None]
    M, N, xdim, udim = Q.shape[0], Q.shape[1], Q.shape[-1], R.shape[-1]
    vec = lambda z, d: None if z is None else T(z).reshape(M, N, d).contiguous()
    X_ref = torch.zeros((M, N, xdim), dtype=torch.float64, device=dev) if X_ref is None else vec(X_ref, xdim)
    U_ref = torch.zeros((M, N, udim), dtype=torch.float64, device=dev) if U_ref is None else vec(U_ref, udim)
    X_prev = X_ref.clone() if X_prev is None else vec(X_prev, xdim).clone()  # default: X_ref (:313)
    U_prev = U_ref.clone() if U_prev is None else vec(U_prev, udim).clone()
    has = lambda z: z is not None and np.size(z) > 0
    lx, ux = (vec(x_l, xdim), vec(x_u, xdim)) if has(x_l) and has(x_u) else (None, None)
    lu, uu = (vec(u_l, udim), vec(u_u, udim)) if has(u_l) and has(u_u) else (None, None)
    Qa, Ra = Q.transpose(-1, -2).contiguous(), R.transpose(-1, -2).contiguous()  # ABI: column-major blocks
    sym = bool(torch.equal(Qa, Q) and torch.equal(Ra, R))
    settings = dict(solver_settings or {})
    solver_name = str(settings.get("solver", "ecos")).lower()  # static_backend.py:242-253
    cone = solver_name in ("ecos", "gurobi", "mosek", "cosmo") or "smooth_cstr" in settings or "smooth_alpha" in settings
    smooth_alpha = float(settings.get("smooth_alpha", math.nan))
    Nc = int(settings.get("Nc", -1))
    slew = None
    if slew_rate is not None and float(slew_rate) != 0.0:
        slew = torch.full((M,), float(slew_rate), dtype=torch.float64, device=dev)
    slew0 = um1 = None
    if "slew_reg" in settings and u0_slew is not None:  # static_backend.py:263-272
        slew0 = torch.full((M,), float(settings["slew_reg"]), dtype=torch.float64, device=dev)
        um1 = T(u0_slew).reshape(-1, udim).expand(M, udim).contiguous()
    s = solver or _solver_for(dev)
    model = _MODELS[builtin_model] if builtin_model is not None else None
    if model is not None:
        assert params is not None, "builtin_model needs `params` (M, 3) unicycle / (M, 4) quadrotor"
        params = T(params).reshape(M, -1).contiguous()
    x0c = x0.contiguous()
    soc_kw = {}
    if soc is None and settings.get("extra_cstrs"):  # the reference's tuple format, stage-wise SOC case only
        from .extra_cstrs import stage_soc_from_extra_cstrs

        tuples = list(settings["extra_cstrs"])
        if len(tuples) != 1:
            raise ValueError("one extra_cstrs tuple (the stage-wise second-order cone) is supported")
        soc = stage_soc_from_extra_cstrs(tuples[0], M, N, xdim, udim, Nc)
        if "soc_u_interior" not in settings:
            raise ValueError("solver_settings['soc_u_interior'] (a control strictly inside the boxes and the cone) is required")
        soc["u_interior"] = settings["soc_u_interior"]
    if soc is not None:
        soc_kw = dict(soc_W=T(soc["W"]).reshape(-1, udim).contiguous(), soc_w0=T(soc["w0"]).reshape(-1).contiguous(),
                      soc_v=T(soc["v"]).reshape(udim).contiguous(), soc_v0=float(soc.get("v0", 0.0)),
                      soc_u_interior=T(soc["u_interior"]).reshape(udim).contiguous())
    Xs = torch.empty((M, N, xdim), dtype=torch.float64, device=dev)
    Us = torch.empty((M, N, udim), dtype=torch.float64, device=dev)

    data: Dict[str, Any] = dict(solver_data=[], hist=[], t_aff_solve=[])
    fields = ["it", "elaps", "obj", "resid", "reg_x", "reg_u"]
    tp = TablePrinter(fields, fmts=["%04d"] + ["%8.3e"] * 5)
    if verbose:
        print(tp.make_header())
    it, max_res = 0, math.inf
    s.stream.wait_stream(torch.cuda.current_stream(dev))  # the set-up above ran on the caller's stream
    while it < max_it:
        if model is not None:
            f, fxa, fua = s.linearize(model, x0c, X_prev, U_prev, params)
        else:
            X_lin = torch.cat([x0[:, None, :], X_prev[:, :-1, :]], 1)
            f, fx, fu = f_fx_fu_fn(X_lin if not single else X_lin[0], U_prev if not single else U_prev[0])
            f = f.reshape(M, N, xdim).contiguous()
            if jacobians_abi_layout:
                fxa, fua = fx.reshape(M, N, xdim, xdim).contiguous(), fu.reshape(M, N, udim, xdim).contiguous()
            else:
                fxa = fx.reshape(M, N, xdim, xdim).transpose(-1, -2).contiguous()
                fua = fu.reshape(M, N, xdim, udim).transpose(-1, -2).contiguous()
        t_aff = time.time()
        kw = dict(f=f, fx=fxa, fu=fua, X_prev=X_prev, U_prev=U_prev, Q=Qa, R=Ra, X_ref=X_ref, U_ref=U_ref, reg_x=float(reg_x),
                  reg_u=float(reg_u), Nc=Nc, x0=x0c, lx=lx, ux=ux, lu=lu, uu=uu, slew_reg=slew, slew_reg0=slew0, slew_um1=um1,
                  X_out=Xs, U_out=Us, symmetric_cost=sym, verbose=bool(settings.get("verbose", False)),
                  static_cons_bounds=it > 0,  # the boxes are the same in every iteration of this loop (scp_mpc.py:338-376)
                  prev_is_last_solution=it > 0)  # and X_prev, U_prev are the previous iteration's X, U (scp_mpc.py:430)
        if soc is not None:
            _, _, status = s.lsoc_solve(**soc_kw, **kw)
        elif cone:
            _, _, status = s.lcone_solve(smooth_alpha=smooth_alpha, **kw)
        else:
            _, _, status = s.lqp_solve(**kw)
        with torch.cuda.stream(s.stream):  # residual / objective row of scp_mpc.py:397-405 on the solver's stream
            res = s.scp_residual(Xs, X_prev, Us, U_prev)[0]
            eX, eU = Xs - X_ref, Us - U_ref
            obj = (torch.sum(eX * torch.einsum("mnrt,mnt->mnr", Q, eX)) + torch.sum(eU * torch.einsum("mnrt,mnt->mnr", R, eU))) / N / M
            row = torch.stack([res, obj]).cpu()  # the only device->host read of the iteration (synchronises the stream)
            X_prev, U_prev = Xs.clone(), Us.clone()
        torch.cuda.current_stream(dev).wait_stream(s.stream)
        t_aff = time.time() - t_aff
        if status != 0 or not bool(torch.isfinite(row).all()):  # solver failure (:391-394)
            if verbose:
                print("Solver failed...")
            return None, None, None
        max_res, obj_v = float(row[0]), float(row[1])
        vals = (it + 1, time.time() - t_start, obj_v, max_res, float(reg_x), float(reg_u))
        if verbose:
            print(tp.make_values(vals))
        data["hist"].append(dict(zip(fields, vals)))
        data["solver_data"].append(dict(s.last_info))
        data["t_aff_solve"].append(t_aff)
        if max_res < res_tol:
            break
        it += 1
        if (time.time() - t_start) * (it + 1) / it > time_limit:
            break
    if verbose:
        print(tp.make_footer())
    X = torch.cat([x0[:, None, :], X_prev], 1)
    U = U_prev
    if single:
        X, U = X[0], U[0]
    if return_torch:
        return X, U, data
    return X.cpu().numpy(), U.cpu().numpy(), data
