"""Device-resident API: the same solve with every buffer already in HBM (torch ROCm tensors are
used purely as device-memory handles), the on-device linearisation of the built-in models and
particle sharding over RCCL (one process per GPU, torch.distributed only bootstraps the
communicator).  This is the path bench.py times; it has no reference counterpart — the reference
copies the Jacobian stacks host<->Julia every SCP iteration (pmpc/static_backend.py:71-74)."""
from __future__ import annotations

import ctypes
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib

MODEL_UNICYCLE, MODEL_QUADROTOR = 0, 1


def _p(t: Optional[torch.Tensor], dtype=torch.float64):
    if t is None:
        return None
    assert t.is_cuda and t.dtype == dtype and t.is_contiguous(), (t.device, t.dtype, t.is_contiguous())
    return ctypes.c_void_p(t.data_ptr())


class DeviceSolver:
    """Owns a pmpc_ctx (HIP stream + workspace cache + optional RCCL communicator)."""

    def __init__(self, device: Optional[int] = None):
        if not torch.cuda.is_available():
            raise RuntimeError("pmpc_amd.DeviceSolver needs a HIP device: there is no CPU path")
        self.lib = _lib.load()
        self.device = torch.cuda.current_device() if device is None else int(device)
        h = ctypes.c_void_p()
        rc = self.lib.pmpc_create(ctypes.byref(h), self.device)
        if rc != 0:
            raise RuntimeError(f"pmpc_create failed ({rc})")
        self.h = h
        self.stream = torch.cuda.ExternalStream(self.lib.pmpc_stream(h), device=self.device)
        self.rank, self.world = 0, 1
        self.last_info: Dict[str, float] = {}

    def set_option(self, key: str, value: float) -> None:
        """Per-context algorithm switch (include/pmpc_abi.h, pmpc_set_option): e.g. ``set_option("xbox_as", 0)``."""
        if self.lib.pmpc_set_option(self.h, key.encode(), float(value)) != 0:
            raise KeyError(f"pmpc_set_option: unknown option {key!r}")

    def get_option(self, key: str) -> float:
        v = ctypes.c_double()
        if self.lib.pmpc_get_option(self.h, key.encode(), ctypes.byref(v)) != 0:
            raise KeyError(f"pmpc_get_option: unknown option {key!r}")
        return v.value

    def close(self):
        if getattr(self, "h", None):
            self.lib.pmpc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- RCCL bootstrap over an existing torch.distributed group ------------------------------------
    def init_comm(self):
        import torch.distributed as dist

        if not dist.is_initialized() or dist.get_world_size() == 1:
            return
        rank, world = dist.get_rank(), dist.get_world_size()
        uid = [None]
        if rank == 0:
            buf = ctypes.create_string_buffer(128)
            rc = self.lib.pmpc_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p))
            if rc != 0:
                raise RuntimeError(f"pmpc_comm_unique_id failed ({rc})")
            uid[0] = bytes(buf.raw)
        dist.broadcast_object_list(uid, src=0)
        buf = ctypes.create_string_buffer(uid[0], 128)
        rc = self.lib.pmpc_comm_init(self.h, rank, world, ctypes.cast(buf, ctypes.c_void_p))
        if rc != 0:
            raise RuntimeError(f"pmpc_comm_init failed ({rc})")
        self.rank, self.world = rank, world

    # ---- stream contract -----------------------------------------------------------------------------
    # The solver runs on its own non-blocking HIP stream.  By default every entry point orders that stream BEHIND the
    # caller's current torch stream before it enqueues (inputs produced by torch ops are complete when read) and orders the
    # caller's current stream behind the solver stream when it returns (outputs can be consumed by torch ops / .cpu()
    # straight away).  `wait_current_stream=False` drops both edges for callers that stay on `self.stream` themselves
    # (bench.py's loop: every producer and consumer is a library call on the solver's stream) and `sync()` at the end.
    def _before(self, on=True):
        if on:
            self.stream.wait_stream(torch.cuda.current_stream(self.device))

    def _after(self, on=True):
        if on:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)

    # ---- solve -----------------------------------------------------------------------------------------
    def _problem(self, *, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref, reg_x, reg_u, Nc=-1, x0=None, lx=None, ux=None,
                 lu=None, uu=None, slew_reg=None, slew_reg0=None, slew_um1=None, X_out=None, U_out=None, weights=None,
                 barrier_mu=0.0, force_generic=False, symmetric_cost=False, cold_start=False, static_cons_bounds=False, prev_is_last_solution=False, soc_W=None, soc_w0=None,
                 soc_v=None, soc_v0=0.0, soc_u_interior=None, cone_k=0, cones=None, cone_objective=False, smooth_cstr="logbarrier", smooth_beta=1.0):
        M, N, x = f.shape
        u = U_prev.shape[-1]
        assert fx.shape == (M, N, x, x) and fu.shape == (M, N, u, x) and Q.shape == (M, N, x, x) and R.shape == (M, N, u, u)
        assert weights is None or weights.shape == (M,)
        if X_out is None:
            X_out = torch.empty((M, N, x), dtype=torch.float64, device=f.device)
        if U_out is None:
            U_out = torch.empty((M, N, u), dtype=torch.float64, device=f.device)
        flags = 0
        if lx is not None and ux is not None:
            flags |= _lib.HAS_XBOUNDS
        if lu is not None and uu is not None:
            flags |= _lib.HAS_UBOUNDS
        if slew_reg is not None:
            flags |= _lib.HAS_SLEW
        if slew_reg0 is not None and slew_um1 is not None:
            flags |= _lib.HAS_SLEW0
        if force_generic:
            flags |= _lib.FORCE_GENERIC
        if cold_start:  # ignore the interior-point iterate remembered from the previous solve of this shape
            flags |= _lib.COLD_START
        if static_cons_bounds:  # sharded SCP loops: the consensus controls' bounds are those of the previous solve (no re-broadcast)
            flags |= _lib.STATIC_CONS_BOUNDS
        if prev_is_last_solution:  # SCP loops: X_prev / U_prev are the previous solve's outputs, boxes unchanged (no rollout)
            flags |= _lib.PREV_IS_LAST_SOLUTION
        if symmetric_cost:  # Q_j, R_j exactly symmetric (enables the register-resident MFMA path)
            flags |= _lib.SYMMETRIC_COST
        if cone_objective:  # scp_loop: the sub-problem is the reference's default path (c_lcone_solve semantics; barrier_mu = 1 / smooth_alpha)
            flags |= _lib.CONE_OBJECTIVE
        # fp32-STORAGE mode (include/pmpc_abi.h PMPC_F32_MATRICES): fx, fu, Q, R as float32 tensors — all four or none
        md = torch.float32 if fx.dtype == torch.float32 else torch.float64
        assert fx.dtype == fu.dtype == Q.dtype == R.dtype == md, "fx, fu, Q, R must share one dtype (float64, or float32 for the fp32-storage mode)"
        if md == torch.float32:
            flags |= _lib.F32_MATRICES
        prob = _lib.PmpcProblem(
            xdim=x, udim=u, N=N, M=M, Nc=int(Nc), flags=flags, reg_x=float(reg_x), reg_u=float(reg_u),
            x0=_p(x0), f=_p(f), fx=_p(fx, md), fu=_p(fu, md), X_prev=_p(X_prev), U_prev=_p(U_prev), Q=_p(Q, md), R=_p(R, md),
            X_ref=_p(X_ref), U_ref=_p(U_ref), lx=_p(lx), ux=_p(ux), lu=_p(lu), uu=_p(uu), slew_reg=_p(slew_reg),
            slew_reg0=_p(slew_reg0), slew_um1=_p(slew_um1), X_out=_p(X_out), U_out=_p(U_out), weights=_p(weights), barrier_mu=float(barrier_mu),
            soc_q=0 if soc_W is None else int(soc_W.shape[0]), soc_W=_p(soc_W), soc_w0=_p(soc_w0), soc_v=_p(soc_v), soc_v0=float(soc_v0),
            soc_u_interior=_p(soc_u_interior), cone_k=int(cone_k), smooth_cstr={"logbarrier": 0, "squareplus": 1}[smooth_cstr], smooth_beta=float(smooth_beta))
        if cones is not None:  # general form of the stage cones: dict(sizes=[q_k], A=tensor, c=tensor); per-stage data if A is (M, N, rows, udim)
            sizes = [int(v) for v in cones["sizes"]]
            rows = sum(sizes) + len(sizes)
            A, cvec = cones["A"], cones["c"]
            per_stage = A.dim() == 4
            assert A.shape == ((M, N, rows, u) if per_stage else (rows, u)) and cvec.shape == ((M, N, rows) if per_stage else (rows,)), (A.shape, cvec.shape)
            self._cone_sizes = (ctypes.c_int * len(sizes))(*sizes)  # (kept alive: the struct holds a bare pointer)
            prob.cone_count, prob.cone_sizes = len(sizes), ctypes.cast(self._cone_sizes, ctypes.POINTER(ctypes.c_int))
            prob.cone_A, prob.cone_c, prob.cone_per_stage = _p(A), _p(cvec), int(per_stage)
        return prob, X_out, U_out

    def lqp_solve(self, *, verbose=False, wait_current_stream=True, **kw):
        """All tensors float64 CUDA, ABI layout: vectors (M,N,d); matrices (M,N,col,row) i.e. the
        transpose of the py layout.  Returns X (M,N,x), U (M,N,u) (steps 1..N, no x0).  `weights` (M,): per-particle
        cost weights (the reference's `weights` setting, PMPC.jl/src/main.jl:96-112)."""
        prob, X_out, U_out = self._problem(**kw)
        info = _lib.PmpcInfo()
        self._before(wait_current_stream)
        status = self.lib.pmpc_lqp_solve_device(self.h, ctypes.byref(prob), ctypes.byref(info), int(verbose))
        self.last_info = {k: getattr(info, k) for k, _ in _lib.PmpcInfo._fields_}
        self._after(wait_current_stream)
        return X_out, U_out, status

    def lcone_solve(self, *, smooth_alpha=float("nan"), verbose=False, wait_current_stream=True, **kw):
        """The cone-path objective of `c_lcone_solve` (epsilon-anchored epigraph, PMPC.jl/src/main.jl:194-354) with
        every buffer in HBM; same tensor conventions as `lqp_solve`.  `cone_k` = the reference's `k` setting (worst-k
        objective for k < M; not reachable through its C ABI)."""
        prob, X_out, U_out = self._problem(**kw)
        info = _lib.PmpcInfo()
        self._before(wait_current_stream)
        status = self.lib.pmpc_lcone_solve_device(self.h, ctypes.byref(prob), float(smooth_alpha), ctypes.byref(info), int(verbose))
        self.last_info = {k: getattr(info, k) for k, _ in _lib.PmpcInfo._fields_}
        self._after(wait_current_stream)
        return X_out, U_out, status

    def lsoc_solve(self, *, verbose=False, wait_current_stream=True, **kw):
        """`lqp_solve` plus one second-order cone ||W u + w0||_2 <= v'u + v0 on the controls of every (particle, stage) —
        `soc_W (q, udim)`, `soc_w0 (q)`, `soc_v (udim)`, `soc_v0`, and `soc_u_interior (udim)`, a control strictly inside
        the boxes and the cone (all float64 CUDA tensors).  Config E's thrust cones; the structured case of the reference's
        pyjulia-only `extra_cstrs` (README.md:219-239).  General form: `cones=dict(sizes=[q_k, ...], A=..., c=...)` — several
        cones per stage, cone k with q_k + 1 rows of `s = A u + c` (q_k = 0: a linear row s >= 0; q_k >= 1: |s[1:]| <= s[0]),
        `A (rows, udim)` / `c (rows,)` shared by all stages or `A (M, N, rows, udim)` / `c (M, N, rows)` per stage; `weights` allowed."""
        prob, X_out, U_out = self._problem(**kw)
        info = _lib.PmpcInfo()
        self._before(wait_current_stream)
        status = self.lib.pmpc_lsoc_solve_device(self.h, ctypes.byref(prob), ctypes.byref(info), int(verbose))
        self.last_info = {k: getattr(info, k) for k, _ in _lib.PmpcInfo._fields_}
        self._after(wait_current_stream)
        return X_out, U_out, status

    def particle_costs(self, X, U, **kw):
        """J_i(X, U) of PMPC.jl/src/qp_utils.jl:60-162 for every particle (device tensor, (M,))."""
        prob, _, _ = self._problem(X_out=X, U_out=U, **kw)
        J = torch.empty((X.shape[0],), dtype=torch.float64, device=X.device)
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        self.lib.pmpc_particle_costs_device(self.h, ctypes.byref(prob), _p(X), _p(U), _p(J))
        self.sync()
        return J

    def linearize(self, model: int, x0, X_prev, U_prev, params, f=None, fx=None, fu=None, wait_current_stream=True):
        """f, fx, fu (ABI layout) at X_ = [x0, X_prev[:-1]], U_prev for a built-in model."""
        M, N, x = X_prev.shape
        u = U_prev.shape[-1]
        dev = X_prev.device
        f = torch.empty((M, N, x), dtype=torch.float64, device=dev) if f is None else f
        fx = torch.empty((M, N, x, x), dtype=torch.float64, device=dev) if fx is None else fx
        fu = torch.empty((M, N, u, x), dtype=torch.float64, device=dev) if fu is None else fu
        self._before(wait_current_stream)
        if fx.dtype == torch.float32:  # fp32-storage mode: the Jacobian stacks are written as float32
            self.lib.pmpc_linearize_device_f32(self.h, int(model), N, M, _p(x0), _p(X_prev), _p(U_prev), _p(params), _p(f), _p(fx, torch.float32),
                                               _p(fu, torch.float32))
        else:
            self.lib.pmpc_linearize_device(self.h, int(model), N, M, _p(x0), _p(X_prev), _p(U_prev), _p(params), _p(f), _p(fx),
                                           _p(fu))
        self._after(wait_current_stream)
        return f, fx, fu

    def scp_residual(self, X, X_prev, U, U_prev, out=None, wait_current_stream=True):
        """max(max_ij ||X - X_prev||_2, max_ij ||U - U_prev||_2) of pmpc/scp_mpc.py:397-403 as a one-element device tensor
        (one fused pass on the solver's stream; inf if a trajectory holds a NaN)."""
        M, N, x = X.shape
        out = torch.empty((1,), dtype=torch.float64, device=X.device) if out is None else out
        self._before(wait_current_stream)
        self.lib.pmpc_scp_residual_device(self.h, x, U.shape[-1], N, M, _p(X), _p(X_prev), _p(U), _p(U_prev), _p(out))
        self._after(wait_current_stream)
        return out

    def scp_loop(self, model: int, params, steps: int, *, f2, fx2, fu2, first_cold=True, res=None, wait_current_stream=True, **kw):
        """`steps` SCP iterations (linearise -> sub-problem -> residual -> swap) for a built-in dynamics model in ONE library
        call: the host is out of the loop body (no Python / ctypes work between iterations, the residual and the next
        linearisation are enqueued behind the sub-problem's rounds before their outcome is read back).  `kw` as for
        `lqp_solve` / `lsoc_solve` with `f, fx, fu` (scratch the linearisation writes), `X_prev, U_prev` (start iterate,
        OVERWRITTEN) and `X_out, U_out`; `f2, fx2, fu2`: a second scratch set.  Returns (res, infos, last_in_out, done):
        per-iteration residuals (device tensor), per-iteration info dicts, whether the final iterate is in (X_out, U_out)
        (else in (X_prev, U_prev)), iterations completed."""
        prob, X_out, U_out = self._problem(**kw)
        res = torch.empty((steps,), dtype=torch.float64, device=f2.device) if res is None else res
        infos = (_lib.PmpcInfo * steps)()
        last = ctypes.c_int(0)
        self._before(wait_current_stream)
        jd = kw["fx"].dtype
        done = self.lib.pmpc_scp_loop_device(self.h, int(model), _p(params), ctypes.byref(prob), _p(f2), _p(fx2, jd), _p(fu2, jd), int(steps),
                                             int(bool(first_cold)), _p(res), infos, ctypes.byref(last))
        self._after(wait_current_stream)
        out = [{k: getattr(infos[i], k) for k, _ in _lib.PmpcInfo._fields_} for i in range(min(done + 1, steps))]
        if out:
            self.last_info = out[min(done, steps) - 1] if done > 0 else out[0]
        return res, out, bool(last.value), done

    def sync(self):
        self.lib.pmpc_sync(self.h)

    def profile(self, level):
        """0 / False: off; 1 / True: HIP events around the dominant kernel (factor sweep) only; 2: every launch class."""
        self.lib.pmpc_profile_enable(self.h, int(level))

    def profile_read(self):
        """{class: (sum_ms, launches)} of HIP-event timings on the solver stream since the last read."""
        ms = (ctypes.c_double * 4)()
        n = (ctypes.c_longlong * 4)()
        self.lib.pmpc_profile_read(self.h, ms, n)
        names = ("bwd_factor", "bwd_vec", "fwd", "consensus")
        out = {k: (ms[i], n[i]) for i, k in enumerate(names)}
        pms, pn = ctypes.c_double(), ctypes.c_longlong()
        self.lib.pmpc_profile_read_partial(self.h, ctypes.byref(pms), ctypes.byref(pn))
        out["bwd_factor_partial"] = (pms.value, pn.value)  # active-set rounds that skip the settled particles (level 2 only)
        ms8, n8 = (ctypes.c_double * 8)(), (ctypes.c_longlong * 8)()
        self.lib.pmpc_profile_read_all(self.h, ms8, n8, 8)
        for k, name in ((5, "as_bookkeeping"), (6, "linearize"), (7, "scp_residual")):
            out[name] = (ms8[k], n8[k])
        return out

    def restart_stats(self, reset: bool = True):
        """Factor sweeps of unsettled particles in the later rounds since the last reset (option ``as_ckpt``):
        ``dict(restarted=, restarted_stages=, full=, full_stages=)`` — sweeps that started from a checkpoint / from the
        terminal cost, and the stages they ran."""
        out = (ctypes.c_ulonglong * 4)()
        self.lib.pmpc_restart_stats(self.h, out, 1 if reset else 0)
        return dict(restarted=int(out[0]), restarted_stages=int(out[1]), full=int(out[2]), full_stages=int(out[3]))


def to_device_problem(prob: dict, device="cuda"):
    """py-layout numpy problem (pmpc_amd.dynamics.make_*_problem) -> ABI-layout CUDA tensors."""
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=device)
    tm = lambda a: torch.as_tensor(np.ascontiguousarray(np.swapaxes(a, -1, -2)), dtype=torch.float64, device=device)
    out = dict(x0=t(prob["x0"]), Q=tm(prob["Q"]), R=tm(prob["R"]), X_ref=t(prob["X_ref"]), U_ref=t(prob["U_ref"]),
               X_prev=t(prob["X_prev"]), U_prev=t(prob["U_prev"]), params=t(prob["params"]))
    for k_src, k_dst in (("u_l", "lu"), ("u_u", "uu"), ("x_l", "lx"), ("x_u", "ux")):
        if prob.get(k_src) is not None:
            out[k_dst] = t(prob[k_src])
    return out
