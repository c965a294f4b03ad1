"""Small host helpers with the reference's names (pmpc/utils.py:6-80)."""
from __future__ import annotations

from typing import Optional

import numpy as np


class TablePrinter:
    """Fixed-width table rows `it, elaps, obj, resid, reg_x, reg_u` like the reference prints
    (pmpc/utils.py:6-62; used at pmpc/scp_mpc.py:325-327, :409-414)."""

    def __init__(self, names, fmts=None, prefix=""):
        self.names = list(names)
        self.fmts = list(fmts) if fmts is not None else ["%9.4e"] * len(self.names)
        self.prefix = prefix
        self.widths = [max(self._width(f), len(n)) + 2 for f, n in zip(self.fmts, self.names)]

    @staticmethod
    def _width(fmt):
        kind = fmt[-1]
        if kind in "fedi":
            return max(len(fmt % 1), len(fmt % -1))
        if kind == "s":
            return len(fmt % "")
        raise ValueError("I can't recognized the [%s] print format" % fmt)

    @staticmethod
    def _pad(s, width, left):
        rem = width - len(s)
        assert rem >= 0
        a, b = rem // 2, rem // 2 + rem % 2
        return " " * a + s + " " * b if left else " " * b + s + " " * a

    def make_row_sep(self):
        return "+" + "".join("-" * w + "+" for w in self.widths)

    def make_header(self):
        row = "".join("|" + self._pad(str(n), w, True) for n, w in zip(self.names, self.widths)) + "|"
        return "\n".join([self.prefix + self.make_row_sep(), self.prefix + row, self.prefix + self.make_row_sep()])

    def make_footer(self):
        return self.prefix + self.make_row_sep()

    def make_values(self, vals):
        assert len(vals) == len(self.fmts)
        return self.prefix + "".join("|" + self._pad(f % v, w, False) for v, f, w in zip(vals, self.fmts, self.widths)) + "|"

    def print_header(self):
        print(self.make_header())

    def print_footer(self):
        print(self.make_footer())

    def print_values(self, vals):
        print(self.make_values(vals))


def atleast_nd(x: Optional[np.ndarray], n: int):
    """Left-pad the shape with ones up to n dims (pmpc/utils.py:65-69)."""
    if x is None:
        return None
    x = np.asarray(x)
    return x.reshape((1,) * max(n - x.ndim, 0) + x.shape)


def to_numpy_f64(x):
    """pmpc/utils.py:72-80; also accepts torch tensors (host or ROCm) and anything array-like."""
    if isinstance(x, np.ndarray):
        return x if x.dtype == np.float64 else x.astype(np.float64)
    if isinstance(x, (float, int)):
        return x
    if hasattr(x, "detach"):  # torch tensor
        return x.detach().cpu().numpy().astype(np.float64, copy=False)
    return np.array(x, dtype=np.float64)
