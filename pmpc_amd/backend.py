"""Python side of the C ABI — the counterpart of the reference's pmpc/static_backend.py.

Same function names, argument order, shape asserts, sentinel conventions and return layout:

* `lqp_solve`  (pmpc/static_backend.py:24-104)   -> `c_lqp_solve`
* `lcone_solve`(pmpc/static_backend.py:107-191)  -> `c_lcone_solve`
* `aff_solve`  (pmpc/static_backend.py:198-312)  -> marshal one SCP sub-problem, prepend x0

Arrays handed to `lqp_solve` / `lcone_solve` use the "Julia" shapes of the reference
(x0 (xdim,M), f (xdim,N,M), fx (xdim,xdim,N,M) ...; pmpc/julia_utils.py:69-77 `py2jl`); they are
made Fortran-contiguous, which yields exactly the column-major buffers the ABI expects
(PMPC.jl/src/c_interface.jl:28-46).  The compute happens in libpmpc_hip.so on the GPU; there
is no CPU path here.
"""
from __future__ import annotations

import math
from copy import copy
from typing import Any, Dict, Optional, Tuple

import numpy as np

from . import _lib
from .utils import atleast_nd, to_numpy_f64


def is_precompiled_backend_available() -> bool:
    """pmpc/static_backend.py:14-21 — here: does libpmpc_hip.so load."""
    try:
        _lib.load()
        return True
    except (ImportError, OSError):
        return False


# ---- layout conversion (pmpc/julia_utils.py:69-88) ---------------------------------------------------
def py2jl(x, keep: int = 1):
    n = x.ndim
    return np.transpose(x, tuple(range(n - keep, n)) + tuple(range(n - keep - 1, -1, -1)))


def jl2py(x, keep: int = 1):
    n = x.ndim
    return np.transpose(x, tuple(range(n - 1, keep - 1, -1)) + tuple(range(0, keep)))


_BLOCK_BIT = {2: 0, 3: 1, 6: 2, 7: 3}  # positions of fx, fu, Q, R -> bit of the `rowmajor` mask (include/pmpc_abi.h)
# True: always go through the reference's two symbols `c_lqp_solve` / `c_lcone_solve` with Fortran-contiguous arrays, exactly as
# pmpc/static_backend.py:71-102 calls pmpcjl (host transposition included) — never through the row-major extension entries.
REFERENCE_SYMBOLS_ONLY = False
CALLS: Dict[str, int] = {}  # library entry point -> number of calls (tests assert which symbol a path used)


def _check_shapes(x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref, lx, ux, lu, uu, slew_reg, slew_reg0, slew_um1):
    assert x0.ndim == 2
    xdim, M = x0.shape
    assert f.ndim == 3
    N = f.shape[1]
    assert fu.ndim == 4
    udim = fu.shape[1]
    assert f.shape == (xdim, N, M), f.shape
    assert fx.shape == (xdim, xdim, N, M), fx.shape
    assert fu.shape == (xdim, udim, N, M), fu.shape
    assert X_prev.shape == (xdim, N, M) and U_prev.shape == (udim, N, M)
    assert Q.shape == (xdim, xdim, N, M) and R.shape == (udim, udim, N, M)
    assert X_ref.shape == (xdim, N, M) and U_ref.shape == (udim, N, M)
    assert lx.shape == (xdim, N, M), lx.shape
    assert ux.shape == (xdim, N, M), ux.shape
    assert lu.shape == (udim, N, M), lu.shape
    assert uu.shape == (udim, N, M), uu.shape
    assert slew_um1.shape == (udim, M)
    assert slew_reg.shape == (M,)
    assert slew_reg0.shape == (M,), slew_reg0.shape
    return xdim, udim, N, M


def _call(entry, Nc, x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref, lx, ux, lu, uu, reg_x, reg_u, slew_reg, slew_reg0,
          slew_um1, verbose, extra=(), cone_k=None, smooth=None):
    lib = _lib.load()
    xdim, udim, N, M = _check_shapes(x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref, lx, ux, lu, uu, slew_reg,
                                     slew_reg0, slew_um1)
    arrs, rowmajor = [], 0
    for k, z in enumerate((x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref, lx, ux, lu, uu, slew_reg, slew_reg0, slew_um1)):
        z = np.asarray(z, dtype=np.float64)
        bit = _BLOCK_BIT.get(k)
        if bit is not None and not REFERENCE_SYMBOLS_ONLY and not z.flags.f_contiguous and z.transpose(3, 2, 0, 1).flags.c_contiguous:
            # `py2jl` view of a C-ordered (M, N, row, col) stack: the blocks are row-major.  The reference transposes them on
            # the host here (f_style cast, static_backend.py:83-101); libpmpc_hip does it in HBM after the upload instead.
            rowmajor |= 1 << bit
            arrs.append(z)
        else:
            arrs.append(np.asfortranarray(z))
    X_out, U_out = np.empty(xdim * N * M), np.empty(udim * N * M)
    ptr = lambda a: a.ctypes.data_as(_lib.c_dp)
    cone_k = int(cone_k) if cone_k is not None else 0
    if rowmajor or cone_k > 0 or smooth is not None:
        entry = {"c_lqp_solve": "pmpc_lqp_solve_host", "c_lcone_solve": "pmpc_lcone_solve_host"}[entry]
        extra = tuple(extra[:1]) + (rowmajor,)  # the `solver` string only selects the conic back end upstream
        if entry == "pmpc_lcone_solve_host":
            extra = extra + (cone_k,)  # the `k` setting cannot cross the reference's C ABI; the extension entry carries it
            if smooth is not None:  # smooth_cstr / smooth_beta (main.jl:247-279): a second extension entry
                entry, extra = "pmpc_lcone_solve_host_ex", extra + (int(smooth[0]), float(smooth[1]))
    CALLS[entry] = CALLS.get(entry, 0) + 1
    getattr(lib, entry)(ptr(X_out), ptr(U_out), xdim, udim, N, M, int(Nc), *[ptr(a) for a in arrs[:14]], float(reg_x),
                        float(reg_u), *[ptr(a) for a in arrs[14:]], int(verbose), *extra)
    # pmpc/static_backend.py:103
    return np.reshape(X_out, (M, N, xdim)), np.reshape(U_out, (M, N, udim))


def lqp_solve(Nc, x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref, lx, ux, lu, uu, reg_x, reg_u, slew_reg, slew_reg0,
              slew_um1, verbose=False):
    """pmpc/static_backend.py:24-104."""
    return _call("c_lqp_solve", Nc, x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref, lx, ux, lu, uu, reg_x, reg_u,
                 slew_reg, slew_reg0, slew_um1, verbose)


def lcone_solve(Nc, x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref, lx, ux, lu, uu, reg_x, reg_u, slew_reg, slew_reg0,
                slew_um1, smooth_alpha=1e1, verbose=False, solver="ecos", k=None, smooth_cstr=None, smooth_beta=1.0):
    """pmpc/static_backend.py:107-191.  `k` (the reference's worst-k setting, PMPC.jl/src/main.jl:204-227) is reachable only
    through pyjulia upstream; here it rides on the extension entry point `pmpc_lcone_solve_host`."""
    if k is not None:
        # main.jl:204-205: k = k >= 0 ? k : M.  k = 0 is legal upstream (the objective loses its epigraph offset) and k > M is
        # not clamped there; neither is implemented here — refused instead of silently read as k = M
        k = int(k)
        if k < 0:
            k = None
        elif k == 0 or k > x0.shape[-1]:
            raise ValueError(f"lcone_solve: k = {k} is not supported (1 <= k <= M = {x0.shape[-1]}, or negative / None for k = M)")
    extra = (float(smooth_alpha), str(solver).encode())
    # `smooth_cstr` (main.jl:247-279; pyjulia-only upstream): "logbarrier" is what a finite smooth_alpha means anyway, "squareplus" (soft
    # boxes, slope smooth_beta) rides on the second extension entry, "" = hard boxes
    smooth = None
    if smooth_cstr is not None:
        if smooth_cstr not in ("", "logbarrier", "squareplus"):
            raise ValueError(f"Unknown smoothing method: [{smooth_cstr}]")  # main.jl:289
        if smooth_cstr == "":
            extra = (float("nan"), extra[1])
        elif smooth_cstr == "squareplus" and smooth_alpha == smooth_alpha:
            smooth = (1, float(smooth_beta))
    return _call("c_lcone_solve", Nc, x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref, lx, ux, lu, uu, reg_x, reg_u,
                 slew_reg, slew_reg0, slew_um1, verbose, extra, cone_k=k, smooth=smooth)


_SOC_SOLVER = None


def _aff_solve_stage_cone(f, fx, fu, x0, X_prev, U_prev, Q, R, X_ref, U_ref, reg_x, reg_u, slew_rate, u_slew, x_l, x_u, u_l, u_u,
                          solver_settings, soc):
    """Sub-problem with one second-order cone ||W u + w0|| <= v'u + v0 on the controls of every (particle, stage) next to the
    control boxes — the structured case of the reference's `extra_cstrs` (README.md:219-239) — through the device solver's
    path-following method (`pmpc_lsoc_solve_device`).  py-layout host arrays in, py-layout out.  The objective is the plain
    sum of the particle costs (the QP's), as for solver = "osqp"."""
    import torch

    from .device import DeviceSolver

    global _SOC_SOLVER
    if slew_rate or u_slew is not None or (x_l is not None and np.size(x_l)):
        raise ValueError("extra_cstrs (stage-wise second-order cone): slew penalties and state boxes are not supported on this path")
    general = "cones" in soc
    if not general and (u_l is None or np.size(u_l) == 0):
        raise ValueError("extra_cstrs (stage-wise second-order cone): control boxes are required (the method starts strictly inside them)")
    if _SOC_SOLVER is None:
        _SOC_SOLVER = DeviceSolver(0)
    s = _SOC_SOLVER
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda")
    tm = lambda a: torch.as_tensor(np.ascontiguousarray(np.swapaxes(a, -1, -2)), dtype=torch.float64, device="cuda")
    # the reference applies extra_cstrs INSIDE its cone solve, i.e. together with the eps-anchored objective, smoothing and the
    # worst-k setting (PMPC.jl/src/main.jl:293-316); this path minimises the plain sum with hard boxes: say so, do not ignore
    ignored = [k for k in ("smooth_alpha", "smooth_cstr", "k", "weights") if solver_settings.get(k) is not None
               and not (k == "smooth_alpha" and isinstance(solver_settings[k], float) and math.isnan(solver_settings[k]))]
    if ignored:
        raise ValueError(f"extra_cstrs (stage cones): solver_settings {ignored} are not supported together with stage cones "
                         "(this path minimises the plain sum of the particle costs with hard boxes)")
    Nc = solver_settings.get("Nc", -1)
    boxes = dict(lu=t(u_l), uu=t(u_u)) if u_l is not None and np.size(u_l) else {}
    sym = bool(np.array_equal(Q, np.swapaxes(Q, -1, -2)) and np.array_equal(R, np.swapaxes(R, -1, -2)))
    common = dict(f=t(f), fx=tm(fx), fu=tm(fu), X_prev=t(X_prev), U_prev=t(U_prev), Q=tm(Q), R=tm(R), X_ref=t(X_ref), U_ref=t(U_ref),
                  reg_x=float(reg_x), reg_u=float(reg_u), Nc=Nc, x0=t(x0), symmetric_cost=sym, **boxes)
    u_int = solver_settings.get("soc_u_interior")
    if general:
        cn = soc["cones"]
        X, U, status = s.lsoc_solve(cones=dict(sizes=cn["sizes"], A=t(cn["A"]), c=t(cn["c"])),
                                    soc_u_interior=None if u_int is None else t(np.asarray(u_int, dtype=np.float64).reshape(-1)), **common)
    else:
        lo_all, hi_all = np.asarray(u_l, dtype=np.float64), np.asarray(u_u, dtype=np.float64)
        if u_int is None:  # a point strictly inside EVERY stage's box: the centre of the intersection of the boxes
            u_int = 0.5 * (lo_all.reshape(-1, lo_all.shape[-1]).max(0) + hi_all.reshape(-1, hi_all.shape[-1]).min(0))
        u_int = np.asarray(u_int, dtype=np.float64).reshape(-1)
        margin = float(np.dot(soc["v"], u_int) + soc["v0"] - np.linalg.norm(np.asarray(soc["W"]) @ u_int + soc["w0"]))
        if not (np.all(lo_all < u_int) and np.all(u_int < hi_all) and margin > 0.0):
            raise ValueError("extra_cstrs (stage-wise second-order cone): solver_settings['soc_u_interior'] must lie strictly inside "
                             f"every stage's control box and the cone (cone margin {margin:.3e}); the default — the centre of the "
                             "boxes' intersection — does not: pass one explicitly")
        X, U, status = s.lsoc_solve(soc_W=t(soc["W"]), soc_w0=t(soc["w0"]), soc_v=t(soc["v"]), soc_v0=float(soc["v0"]), soc_u_interior=t(u_int),
                                    **common)
    s.sync()
    X, U = X.cpu().numpy(), U.cpu().numpy()
    if status != 0:
        X[:], U[:] = np.nan, np.nan
    return np.concatenate([np.asarray(x0)[:, None, :], X], -2), U, dict()


def aff_solve(
    f: np.ndarray, fx: np.ndarray, fu: np.ndarray, x0: np.ndarray, X_prev: np.ndarray, U_prev: np.ndarray,
    Q: np.ndarray, R: np.ndarray, X_ref: np.ndarray, U_ref: np.ndarray, reg_x: float, reg_u: float,
    slew_rate: Optional[float], u_slew: Optional[np.ndarray], x_l: np.ndarray, x_u: np.ndarray, u_l: np.ndarray,
    u_u: np.ndarray, solver_settings: Optional[Dict[str, Any]] = None,
) -> Tuple[np.ndarray, np.ndarray, Any]:
    """Solve a single instance of a linearized MPC problem (pmpc/static_backend.py:198-312)."""
    solver_settings = copy(solver_settings) if solver_settings is not None else dict()
    f = atleast_nd(to_numpy_f64(f), 3)
    fx, fu = atleast_nd(to_numpy_f64(fx), 4), atleast_nd(to_numpy_f64(fu), 4)
    x0 = atleast_nd(to_numpy_f64(x0), 2)
    X_prev, U_prev = atleast_nd(to_numpy_f64(X_prev), 3), atleast_nd(to_numpy_f64(U_prev), 3)
    Q, R = atleast_nd(to_numpy_f64(Q), 4), atleast_nd(to_numpy_f64(R), 4)
    X_ref, U_ref = atleast_nd(to_numpy_f64(X_ref), 3), atleast_nd(to_numpy_f64(U_ref), 3)
    x_l, x_u, u_l, u_u = [None if z is None else atleast_nd(to_numpy_f64(z), 3) for z in (x_l, x_u, u_l, u_u)]

    # user constraints in the reference's tuple format (pyjulia only upstream, PMPC.jl/src/main.jl:293-316): the cases this
    # back end implements are folded in here; anything else is REFUSED, never dropped silently
    soc = None
    aux_from = None  # (set when rows on the states were restated as auxiliary state components: stripped from the result)
    if solver_settings.get("extra_cstrs"):
        from .extra_cstrs import linear_rows_to_boxes, stage_cones_from_extra_cstrs, stage_soc_from_extra_cstrs

        Mb, Nb, xd, ud = f.shape[0], f.shape[1], f.shape[2], fu.shape[-1]
        Ncb = solver_settings.get("Nc", -1)
        # ---- cost terms and exponential cones of the tuples, normalised first ---------------------------------------------------------
        from .extra_cstrs import split_linear_cost

        cone_method = str(solver_settings.get("solver", "ecos")).lower() != "osqp" or "smooth_cstr" in solver_settings or "smooth_alpha" in solver_settings
        tuples = []
        for cstr in solver_settings["extra_cstrs"]:
            cu_, cx_, cstr = split_linear_cost(cstr, Mb, Nb, xd, ud, Ncb)
            if cu_ is not None:
                # c_left sits OUTSIDE the epigraph rows (cone_utils.jl:152-154).  One particle: the epigraph row is degenerate, the program is
                # min (1 - eps) J + c'z (+ smoothing), i.e. J with its references shifted by Q^-1 c / (1 - eps) — the reference's own device for
                # linear cost terms (pmpc/scp_mpc.py:171-185).  Several particles: a particle whose epigraph multiplier is zero would be left
                # with c'z alone — a linear program the Riccati structure cannot carry: refused.
                if not cone_method:
                    raise ValueError("extra_cstrs act in the cone program only (solver 'ecos' / 'mosek' / 'gurobi' / 'cosmo'); c_left with solver 'osqp' has no meaning upstream")
                if Mb != 1:
                    raise ValueError("extra_cstrs: c_left (a linear cost outside the epigraph rows) is supported for one particle only: with several, the "
                                     "particles of multiplier zero are left with a linear program in their own variables")
                if solver_settings.get("weights") is not None:
                    raise ValueError("extra_cstrs: c_left together with `weights` is not supported")
                COST_ANCHOR_EPS = 1e-3  # main.jl:223
                X_ref = X_ref - np.linalg.solve(Q, cx_[..., None])[..., 0] / (1.0 - COST_ANCHOR_EPS)
                U_ref = U_ref - np.linalg.solve(R, cu_[..., None])[..., 0] / (1.0 - COST_ANCHOR_EPS)
            if int(cstr[2]) > 0:
                # Exponential cones (cone_solver.jl:178-188).  Upstream they appear as the log-barrier triples of `make_logbarrier_constraint`
                # (cone_utils.jl:173-200), each with a new epigraph variable in G_right.  Under smooth_cstr = "logbarrier" the reference itself
                # refuses any tuple with G_right (main.jl:300, "We only support left matrix reformulation") and smooths a tuple's LINEAR rows
                # instead — which is supported here; with hard or squareplus boxes such rows would be barrier terms NEXT TO hard / hinged rows:
                # a barrier weight per row, which neither the active-set rounds nor the single-mu barrier iteration carries.
                raise ValueError("extra_cstrs: exponential cones are not supported (under smooth_cstr='logbarrier' hand the rows over as LINEAR rows: "
                                 "the reference smooths them into exactly those cones, main.jl:298-312, and refuses G_right there itself; with hard or "
                                 "squareplus boxes a log-barrier row next to them needs a barrier weight of its own)")
            if int(cstr[0]) + (sum(int(v) for v in np.atleast_1d(cstr[1])) if np.size(cstr[1]) else 0) > 0:
                tuples.append(cstr)
        rest = []
        for cstr in tuples:
            try:  # single-variable linear rows: boxes
                bx = linear_rows_to_boxes(cstr, Mb, Nb, xd, ud, Ncb)
            except ValueError:
                rest.append(cstr)
                continue
            merged = []
            for cur, new, is_lo, like in ((x_l, bx[0], True, X_prev), (x_u, bx[1], False, X_prev), (u_l, bx[2], True, U_prev), (u_u, bx[3], False, U_prev)):
                if np.all(np.isinf(new)):
                    merged.append(cur)
                    continue
                base = np.full(like.shape, -np.inf if is_lo else np.inf) if cur is None or cur.size == 0 else cur
                merged.append(np.maximum(base, new) if is_lo else np.minimum(base, new))
            x_l, x_u, u_l, u_u = merged
            # a one-sided box needs its other side too (the ABI takes lower and upper together)
            if (x_l is None or x_l.size == 0) != (x_u is None or x_u.size == 0):
                x_l = np.full(X_prev.shape, -np.inf) if x_l is None or x_l.size == 0 else x_l
                x_u = np.full(X_prev.shape, np.inf) if x_u is None or x_u.size == 0 else x_u
            if (u_l is None or u_l.size == 0) != (u_u is None or u_u.size == 0):
                u_l = np.full(U_prev.shape, -np.inf) if u_l is None or u_l.size == 0 else u_l
                u_u = np.full(U_prev.shape, np.inf) if u_u is None or u_u.size == 0 else u_u
        # linear rows that involve STATES (obstacle half-spaces, rows coupling x and u of a stage): restated as upper bounds on auxiliary
        # states that the dynamics produce (extra_cstrs.aux_state_problem) — the solve below then sees state boxes only
        from scipy.sparse import csr_matrix as sp_csr

        ncu_b = (Nb if Ncb < 0 else min(int(Ncb), Nb))
        ncu_b = ncu_b * ud + Mb * (Nb - ncu_b) * ud
        on_states = lambda c_: int(c_[0]) > 0 and sp_csr(c_[3]).shape[1] > ncu_b and sp_csr(c_[3])[:, ncu_b:].count_nonzero() > 0
        state_rows = [c_ for c_ in rest if on_states(c_)]
        if state_rows:
            from .extra_cstrs import aux_state_problem, stage_rows_from_extra_cstrs

            if len(state_rows) != len(rest):
                raise ValueError("extra_cstrs: rows on the states together with cones on the controls are not supported in one solve")
            if slew_rate or u_slew is not None or "slew_reg" in solver_settings:
                raise ValueError("extra_cstrs: rows on the states are not supported together with slew penalties")
            if str(solver_settings.get("smooth_cstr", "")).lower() == "squareplus" and np.isfinite(float(solver_settings.get("smooth_alpha", np.nan))):
                # the reference smooths extra_cstrs' linear rows only under "logbarrier" (main.jl:298-312); under "squareplus" they stay HARD
                # rows next to softened boxes — restated as state boxes they would be softened with them: refused, not solved differently
                raise ValueError('extra_cstrs: rows on the states together with smooth_cstr="squareplus" are not supported '
                                 "(the reference keeps them hard there; this path would soften them with the boxes)")
            try:
                rows = stage_rows_from_extra_cstrs(state_rows, Mb, Nb, xd, ud, Ncb)
            except ValueError as e_rows:
                raise ValueError(f"extra_cstrs: rows on the states must be linear rows on the state and control of ONE stage of one "
                                 f"particle; this is not ({e_rows})") from None
            aug = aux_state_problem(rows, x0, f, fx, fu, X_prev, U_prev, Q, X_ref, reg_x, x_l, x_u)
            x0, f, fx, fu, X_prev, Q, X_ref, x_l, x_u = (aug[k_] for k_ in ("x0", "f", "fx", "fu", "X_prev", "Q", "X_ref", "x_l", "x_u"))
            aux_from = xd
            rest = []
        if rest:
            # conic rows inside one stage's controls: ONE second-order cone with the same data on every stage keeps the
            # path-following fallback (soc); anything else stage-local goes to the general form (several cones / linear rows
            # per stage, stage-dependent data); what is neither is REFUSED with the reason, never dropped silently
            try:
                if len(rest) != 1:
                    raise ValueError("several tuples")
                soc = stage_soc_from_extra_cstrs(rest[0], Mb, Nb, xd, ud, Ncb)
            except ValueError as e_soc:
                try:
                    soc = dict(cones=stage_cones_from_extra_cstrs(rest, Mb, Nb, xd, ud, Ncb))
                except ValueError as e_gen:
                    raise ValueError("extra_cstrs: supported are linear rows and second-order cones on the controls of ONE stage (the same "
                                     f"list of cones on every stage); this is not ({e_gen}; as a single uniform cone: {e_soc})") from None
    if soc is not None:
        return _aff_solve_stage_cone(f, fx, fu, x0, X_prev, U_prev, Q, R, X_ref, U_ref, reg_x, reg_u, slew_rate, u_slew, x_l, x_u, u_l, u_u,
                                     solver_settings, soc)

    x_l, x_u, u_l, u_u = [None if z is None else py2jl(z, 1) for z in (x_l, x_u, u_l, u_u)]
    x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = [
        py2jl(z, d) for z, d in zip((x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref), (1, 1, 2, 2, 1, 1, 2, 2, 1, 1))]

    solver_settings.setdefault("solver", "ecos")
    solver = solver_settings["solver"].lower()
    assert solver in ("ecos", "gurobi", "mosek", "cosmo", "osqp")
    if solver in ("ecos", "gurobi", "mosek", "cosmo") or "smooth_cstr" in solver_settings or "smooth_alpha" in solver_settings:
        method, smooth_alpha = "cone", solver_settings.get("smooth_alpha", math.nan)  # static_backend.py:244-250
    else:
        method, smooth_alpha = "qp", None

    Nc = -1 if "Nc" not in solver_settings else solver_settings["Nc"]
    nan_like = lambda z: np.nan * np.zeros_like(z)
    x_l = nan_like(X_prev) if x_l is None or x_l.size == 0 else x_l
    x_u = nan_like(X_prev) if x_u is None or x_u.size == 0 else x_u
    u_l = nan_like(U_prev) if u_l is None or u_l.size == 0 else u_l
    u_u = nan_like(U_prev) if u_u is None or u_u.size == 0 else u_u
    M = x0.shape[-1]
    slew_reg = (math.nan if slew_rate is None else slew_rate) * np.ones(M)
    # the reference feeds solver_settings["slew_reg"] into slew_reg0 (static_backend.py:263-267)
    slew_reg0 = np.nan * np.zeros(M) if "slew_reg" not in solver_settings else solver_settings["slew_reg"] * np.ones(M)
    slew_um1 = nan_like(U_prev[:, 0, :]) if u_slew is None else py2jl(atleast_nd(to_numpy_f64(u_slew), 2), 1)

    args = (Nc, x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref, x_l, x_u, u_l, u_u, reg_x, reg_u, slew_reg, slew_reg0,
            slew_um1)
    verbose = solver_settings.get("verbose", False)
    if method == "qp":
        X, U = lqp_solve(*args, verbose=verbose)
    else:
        skw = {}  # (only when asked for: the call is the reference's otherwise, static_backend.py:189)
        if solver_settings.get("smooth_cstr") is not None:
            skw = dict(smooth_cstr=solver_settings["smooth_cstr"], smooth_beta=solver_settings.get("smooth_beta", 1.0))
        X, U = lcone_solve(*args, smooth_alpha, verbose=verbose, solver=solver_settings["solver"], k=solver_settings.get("k"), **skw)
    X_traj = np.concatenate([np.swapaxes(x0, -1, -2)[:, None, :], X], -2)  # static_backend.py:311
    if aux_from is not None:
        X_traj = np.ascontiguousarray(X_traj[..., :aux_from])
    return X_traj, U, dict()
