"""`Problem` — keyword-bag builder for `solve(**problem)` with the reference's defaults
(pmpc/problem_struct.py:10-155): dimensions are inferred from any array argument, missing pieces get
defaults (Q = I, R = 0.1 I, zero references, X_prev = x0 tiled, reg_x = reg_u = 1, max_it = 30,
res_tol = 1e-6, verbose), per-stage arrays given without the particle axis are tiled over `M`, and
`Nc` is injected into `solver_settings` for multi-particle problems (`to_dict`, :119-142)."""
from __future__ import annotations

from collections.abc import Mapping
from copy import copy
from typing import Dict, Tuple
from warnings import warn

import numpy as np

_DIM_MAP: Dict[str, Tuple[str, ...]] = {
    "Q": ("N", "xdim", "xdim"), "R": ("N", "udim", "udim"), "X_ref": ("N", "xdim"), "U_ref": ("N", "udim"),
    "X_prev": ("N", "xdim"), "U_prev": ("N", "udim"), "u_l": ("N", "udim"), "u_u": ("N", "udim"),
    "x_l": ("N", "xdim"), "x_u": ("N", "xdim"), "x0": ("xdim",),
}
_PLAIN_KEYS = ["solver_settings", "reg_x", "reg_u", "max_it", "res_tol", "verbose", "slew_rate", "P"]
_OPTIONAL_KEYS = ["lin_cost_fn", "extra_cstrs_fns"]


class Problem(Mapping):
    dim_map = _DIM_MAP

    def __init__(self, **kw):
        object.__setattr__(self, "_arrays", {})
        self._dims = self._infer_dims(kw)
        self.M = kw.get("M", None)
        N, x, u = self._dims["N"], self._dims["xdim"], self._dims["udim"]
        # defaults (problem_struct.py:88-102)
        self._arrays.update(
            Q=np.tile(np.eye(x), (N, 1, 1)), R=np.tile(0.1 * np.eye(u), (N, 1, 1)), x0=np.zeros(x),
            X_ref=np.zeros((N, x)), U_ref=np.zeros((N, u)), X_prev=np.zeros((N, x)), U_prev=np.zeros((N, u)),
            u_l=None, u_u=None, x_l=None, x_u=None)
        self.solver_settings = dict()
        self.reg_x, self.reg_u, self.max_it, self.res_tol, self.verbose = 1e0, 1e0, 30, 1e-6, True
        self.slew_rate = None
        self.P = None
        for k, v in kw.items():
            if k.startswith("_"):
                warn(f"Cannot set private attribute {k}")
            elif k not in ("N", "xdim", "udim", "M"):
                setattr(self, k, v)
        for k in _DIM_MAP:  # tile the defaults over the particle axis
            setattr(self, k, self._arrays[k])
        if not hasattr(self, "Nc"):
            self.Nc = 0

    # ---- dimensions ------------------------------------------------------------------------------------
    @staticmethod
    def _infer_dims(kw):
        dims = {k: int(kw[k]) for k in ("N", "xdim", "udim") if k in kw}
        for k, names in _DIM_MAP.items():
            if k in kw and kw[k] is not None:
                shp = np.shape(kw[k])
                for i in range(1, min(len(names), len(shp)) + 1):  # trailing axes only (lower-rank arrays get tiled)
                    dims[names[-i]] = shp[-i]
        for k in ("N", "xdim", "udim"):
            if k not in dims:
                raise ValueError(f"Missing dimension {k}")
        return dims

    @property
    def dims(self):
        return copy(self._dims)

    N = property(lambda self: self._dims["N"])
    xdim = property(lambda self: self._dims["xdim"])
    udim = property(lambda self: self._dims["udim"])

    def __repr__(self):
        return f"Problem({self._dims}, id={abs(hash(str(id(self))))})"

    # ---- shape-checked, M-tiled array attributes ------------------------------------------------------
    def __setattr__(self, k, v):
        if k in _DIM_MAP:
            if v is not None:
                v = np.array(v)
                full = tuple(self._dims[d] for d in _DIM_MAP[k])
                if self.M is not None:
                    full = (self.M,) + full
                assert v.shape == full[-v.ndim:] if v.ndim else True, (
                    f"v does not have the correct shape, v.shape = {v.shape}, correct_shape = {full[-v.ndim:]}")
                v = np.tile(v, full[: len(full) - v.ndim] + (1,) * v.ndim)
            self._arrays[k] = v
        else:
            object.__setattr__(self, k, v)

    def __getattr__(self, k):
        arrays = object.__getattribute__(self, "_arrays")
        if k in arrays:
            return arrays[k]
        raise AttributeError(k)

    # ---- mapping interface: solve(**problem) -------------------------------------------------------------
    def to_dict(self):
        problem = {k: self._arrays[k] for k in _DIM_MAP}
        problem.update({k: getattr(self, k, None) for k in _PLAIN_KEYS})
        if self.M is not None:
            ss = problem["solver_settings"]
            if "Nc" in ss and ss["Nc"] != self.Nc:
                warn("Nc specified in solver_settings, but Problem specifies Nc via a property. "
                     f"We will use Nc = {self.Nc} from the Problem.")
            ss["Nc"] = self.Nc
        if "f_fx_fu_fn" in self.__dict__:
            problem["f_fx_fu_fn"] = self.f_fx_fu_fn
        else:
            warn("No dynamics function specified, please set `prob.f_fx_fu_fn`")
        problem.update({k: getattr(self, k) for k in _OPTIONAL_KEYS if k in self.__dict__})
        return problem

    def __iter__(self):
        return iter(self.to_dict().keys())

    def __getitem__(self, k):
        return self.to_dict()[k]

    def __len__(self):
        return len(self.to_dict())
