"""ctypes binding of libpmpc_hip.so (include/pmpc_abi.h).

The library is built in-tree by `__graft_entry__.build()` / `make -C pmpc_amd/csrc`.  There is no
CPU path: `load()` raises if the shared object is missing, and every solve raises if no HIP device
is present (the product path must fail loudly rather than fall back)."""
from __future__ import annotations

import ctypes
import os
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ["PMPC_HIP_LIB"]) if os.environ.get("PMPC_HIP_LIB") else _HERE / "libpmpc_hip.so"  # (env: diagnostic / A-B builds)
_lib = None

c_dp = ctypes.POINTER(ctypes.c_double)

# flags of pmpc_problem.flags (include/pmpc_abi.h)
HAS_XBOUNDS, HAS_UBOUNDS, HAS_SLEW, HAS_SLEW0, FORCE_GENERIC, SYMMETRIC_COST, COLD_START, STATIC_CONS_BOUNDS, PREV_IS_LAST_SOLUTION, F32_MATRICES, CONE_OBJECTIVE = 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024

ABI_SYMBOLS = [
    "c_lqp_solve", "c_lcone_solve", "pmpc_lqp_solve_host", "pmpc_lcone_solve_host", "pmpc_create", "pmpc_destroy", "pmpc_stream", "pmpc_sync", "pmpc_lqp_solve_device",
    "pmpc_comm_unique_id", "pmpc_comm_init", "pmpc_comm_rank", "pmpc_comm_world", "pmpc_linearize_device",
    "pmpc_profile_enable", "pmpc_profile_read", "pmpc_version", "pmpc_lcone_solve_device", "pmpc_particle_costs_device", "pmpc_lsoc_solve_device", "pmpc_comm_init_mock",
    "pmpc_scp_residual_device", "pmpc_profile_read_partial", "pmpc_profile_read_all", "pmpc_scp_loop_device", "pmpc_linearize_device_f32",
    "pmpc_set_option", "pmpc_get_option", "pmpc_abi_struct_sizes", "pmpc_lcone_solve_host_ex", "pmpc_restart_stats",
]


class PmpcProblem(ctypes.Structure):
    _fields_ = (
        [(k, ctypes.c_size_t) for k in ("xdim", "udim", "N", "M")]
        + [("Nc", ctypes.c_longlong), ("flags", ctypes.c_uint), ("reg_x", ctypes.c_double), ("reg_u", ctypes.c_double)]
        + [(k, ctypes.c_void_p) for k in ("x0", "f", "fx", "fu", "X_prev", "U_prev", "Q", "R", "X_ref", "U_ref",
                                          "lx", "ux", "lu", "uu", "slew_reg", "slew_reg0", "slew_um1", "X_out", "U_out", "weights")]
        + [("barrier_mu", ctypes.c_double), ("soc_q", ctypes.c_size_t), ("soc_W", ctypes.c_void_p), ("soc_w0", ctypes.c_void_p),
           ("soc_v", ctypes.c_void_p), ("soc_v0", ctypes.c_double), ("soc_u_interior", ctypes.c_void_p),
           ("cone_count", ctypes.c_size_t), ("cone_sizes", ctypes.POINTER(ctypes.c_int)), ("cone_A", ctypes.c_void_p), ("cone_c", ctypes.c_void_p),
           ("cone_per_stage", ctypes.c_int), ("cone_k", ctypes.c_longlong), ("smooth_cstr", ctypes.c_int), ("smooth_beta", ctypes.c_double)]
    )


class PmpcInfo(ctypes.Structure):
    _fields_ = [("status", ctypes.c_int), ("ipm_iters", ctypes.c_int), ("structured_solves", ctypes.c_int),
                ("fast_path", ctypes.c_int), ("mu", ctypes.c_double), ("slack_res", ctypes.c_double),
                ("max_violation", ctypes.c_double), ("outer_solves", ctypes.c_int), ("active_set_rounds", ctypes.c_int)]


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C pmpc_amd/csrc). pmpc_amd has no CPU fallback.")
    lib = ctypes.CDLL(str(LIB_PATH), mode=ctypes.RTLD_GLOBAL)
    sz, ll, dbl, vp = ctypes.c_size_t, ctypes.c_longlong, ctypes.c_double, ctypes.c_void_p
    common = [c_dp, c_dp, sz, sz, sz, sz, ll] + [c_dp] * 14 + [dbl, dbl] + [c_dp] * 3 + [ll]
    lib.c_lqp_solve.argtypes = common
    lib.c_lqp_solve.restype = None
    lib.c_lcone_solve.argtypes = common + [dbl, ctypes.c_char_p]
    lib.c_lcone_solve.restype = None
    lib.pmpc_lqp_solve_host.argtypes = common + [ctypes.c_uint]
    lib.pmpc_lqp_solve_host.restype = None
    lib.pmpc_lcone_solve_host.argtypes = common + [dbl, ctypes.c_uint, ll]
    lib.pmpc_lcone_solve_host.restype = None
    lib.pmpc_lcone_solve_host_ex.argtypes = common + [dbl, ctypes.c_uint, ll, ctypes.c_int, dbl]
    lib.pmpc_lcone_solve_host_ex.restype = None
    lib.pmpc_create.argtypes = [ctypes.POINTER(vp), ctypes.c_int]
    lib.pmpc_create.restype = ctypes.c_int
    lib.pmpc_destroy.argtypes = [vp]
    lib.pmpc_destroy.restype = None
    lib.pmpc_set_option.argtypes = [vp, ctypes.c_char_p, ctypes.c_double]
    lib.pmpc_set_option.restype = ctypes.c_int
    lib.pmpc_get_option.argtypes = [vp, ctypes.c_char_p, c_dp]
    lib.pmpc_get_option.restype = ctypes.c_int
    lib.pmpc_stream.argtypes = [vp]
    lib.pmpc_stream.restype = vp
    lib.pmpc_sync.argtypes = [vp]
    lib.pmpc_sync.restype = None
    lib.pmpc_lqp_solve_device.argtypes = [vp, ctypes.POINTER(PmpcProblem), ctypes.POINTER(PmpcInfo), ctypes.c_int]
    lib.pmpc_lqp_solve_device.restype = ctypes.c_int
    lib.pmpc_lcone_solve_device.argtypes = [vp, ctypes.POINTER(PmpcProblem), dbl, ctypes.POINTER(PmpcInfo), ctypes.c_int]
    lib.pmpc_lcone_solve_device.restype = ctypes.c_int
    lib.pmpc_lsoc_solve_device.argtypes = [vp, ctypes.POINTER(PmpcProblem), ctypes.POINTER(PmpcInfo), ctypes.c_int]
    lib.pmpc_lsoc_solve_device.restype = ctypes.c_int
    lib.pmpc_particle_costs_device.argtypes = [vp, ctypes.POINTER(PmpcProblem), vp, vp, vp]
    lib.pmpc_particle_costs_device.restype = ctypes.c_int
    lib.pmpc_comm_unique_id.argtypes = [vp]
    lib.pmpc_comm_unique_id.restype = ctypes.c_int
    lib.pmpc_comm_init.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp]
    lib.pmpc_comm_init.restype = ctypes.c_int
    lib.pmpc_comm_init_mock.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.pmpc_comm_init_mock.restype = ctypes.c_int
    lib.pmpc_comm_rank.argtypes = [vp]
    lib.pmpc_comm_rank.restype = ctypes.c_int
    lib.pmpc_comm_world.argtypes = [vp]
    lib.pmpc_comm_world.restype = ctypes.c_int
    lib.pmpc_linearize_device.argtypes = [vp, ctypes.c_int, sz, sz] + [vp] * 7
    lib.pmpc_linearize_device.restype = ctypes.c_int
    lib.pmpc_linearize_device_f32.argtypes = [vp, ctypes.c_int, sz, sz] + [vp] * 7
    lib.pmpc_linearize_device_f32.restype = ctypes.c_int
    lib.pmpc_scp_residual_device.argtypes = [vp, sz, sz, sz, sz] + [vp] * 5
    lib.pmpc_scp_residual_device.restype = ctypes.c_int
    lib.pmpc_profile_enable.argtypes = [vp, ctypes.c_int]
    lib.pmpc_profile_enable.restype = None
    lib.pmpc_profile_read.argtypes = [vp, c_dp, ctypes.POINTER(ctypes.c_longlong)]
    lib.pmpc_profile_read.restype = None
    lib.pmpc_profile_read_partial.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_longlong)]
    lib.pmpc_profile_read_partial.restype = None
    lib.pmpc_restart_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    lib.pmpc_restart_stats.restype = None
    lib.pmpc_profile_read_all.argtypes = [vp, c_dp, ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
    lib.pmpc_profile_read_all.restype = None
    lib.pmpc_scp_loop_device.argtypes = [vp, ctypes.c_int, vp, ctypes.POINTER(PmpcProblem), vp, vp, vp, ctypes.c_int, ctypes.c_int, vp,
                                         ctypes.POINTER(PmpcInfo), ctypes.POINTER(ctypes.c_int)]
    lib.pmpc_scp_loop_device.restype = ctypes.c_int
    lib.pmpc_version.argtypes = []
    lib.pmpc_version.restype = ctypes.c_char_p
    # layout check: the library's structs against this binding's mirrors (include/pmpc_abi.h: pmpc_abi_struct_sizes)
    if not hasattr(lib, "pmpc_abi_struct_sizes"):
        raise ImportError(f"{LIB_PATH} predates this binding (no pmpc_abi_struct_sizes): rebuild it (make -C pmpc_amd/csrc)")
    lib.pmpc_abi_struct_sizes.argtypes = [ctypes.POINTER(sz), ctypes.POINTER(sz)]
    lib.pmpc_abi_struct_sizes.restype = None
    sp, si = sz(0), sz(0)
    lib.pmpc_abi_struct_sizes(ctypes.byref(sp), ctypes.byref(si))
    if (sp.value, si.value) != (ctypes.sizeof(PmpcProblem), ctypes.sizeof(PmpcInfo)):
        raise ImportError(f"{LIB_PATH} ({lib.pmpc_version().decode()}) and pmpc_amd/_lib.py disagree on the ABI structs: library "
                          f"{sp.value}/{si.value} bytes, binding {ctypes.sizeof(PmpcProblem)}/{ctypes.sizeof(PmpcInfo)}: rebuild the library")
    _lib = lib
    return lib


def dptr(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_dp)
