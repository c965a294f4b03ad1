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


def _array_defaults(N: int, x: int, u: int) -> Dict[str, object]:
    """Per-particle defaults of the array fields (what pmpc/problem_struct.py:88-102 fills in): identity state cost,
    0.1 I control cost, everything else zero, no boxes."""
    zeros = {"x0": (x,), "X_ref": (N, x), "U_ref": (N, u), "X_prev": (N, x), "U_prev": (N, u)}
    out: Dict[str, object] = {k: np.zeros(shape) for k, shape in zeros.items()}
    out["Q"] = np.broadcast_to(np.eye(x), (N, x, x)).copy()
    out["R"] = np.broadcast_to(0.1 * np.eye(u), (N, u, u)).copy()
    out.update(dict.fromkeys(("u_l", "u_u", "x_l", "x_u")))
    return out


# scalar / option fields and their defaults
_OPTION_DEFAULTS = dict(reg_x=1.0, reg_u=1.0, max_it=30, res_tol=1e-6, verbose=True, slew_rate=None, P=None)
_DIM_KEYS = ("N", "xdim", "udim", "M")


class Problem(Mapping):
    dim_map = _DIM_MAP

    def __init__(self, **kw):
        object.__setattr__(self, "_arrays", {})
        self._dims = self._infer_dims(kw)
        self.M = kw.get("M")
        self.solver_settings = {}
        for name, value in _OPTION_DEFAULTS.items():
            setattr(self, name, value)
        # array fields: the caller's value where given, the default otherwise — both go through __setattr__, which checks the
        # shape against the inferred dimensions and tiles over the particle axis
        fields = _array_defaults(self._dims["N"], self._dims["xdim"], self._dims["udim"])
        for name, value in kw.items():
            if name.startswith("_"):
                warn(f"Cannot set private attribute {name}")
            elif name in fields:
                fields[name] = value
            elif name not in _DIM_KEYS:
                setattr(self, name, value)
        for name in _DIM_MAP:
            setattr(self, name, fields[name])
        if not hasattr(self, "Nc"):  # (a subclass may define Nc as a class attribute: pmpc/problem_struct.py:57-58)
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
        d = self._dims
        batch = "" if self.M is None else f"M={self.M}, "
        return f"Problem({batch}N={d['N']}, xdim={d['xdim']}, udim={d['udim']}) at {id(self):#x}"

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
