"""Small host helpers under the reference's names (pmpc/utils.py): the fixed-width SCP progress table, `atleast_nd`,
`to_numpy_f64`."""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence

import numpy as np

_NUMERIC_KINDS = frozenset("fedi")


def _cell_width(fmt: str, title: str) -> int:
    """Room for the widest rendering of `fmt` (sign included) and for the column title, plus one blank either side."""
    kind = fmt[-1:]
    if kind in _NUMERIC_KINDS:
        body = max(len(fmt % probe) for probe in (1, -1))
    elif kind == "s":
        body = len(fmt % "")
    else:
        raise ValueError("I can't recognized the [%s] print format" % fmt)
    return max(body, len(title)) + 2


def _centre(text: str, width: int, spare_right: bool) -> str:
    """Centre `text`; an odd spare blank goes to the right (titles) or to the left (values), as the reference's table does."""
    half = (width - len(text)) // 2
    if half < 0 or len(text) > width:
        raise AssertionError("cell wider than its column")
    return text.rjust(len(text) + half).ljust(width) if spare_right else text.ljust(len(text) + half).rjust(width)


class TablePrinter:
    """`+----+…+` framed rows `it | elaps | obj | resid | reg_x | reg_u` of the SCP loop (layout of pmpc/utils.py:6-62;
    used by scp_solve at pmpc/scp_mpc.py:325-327, :409-414)."""

    def __init__(self, names: Sequence[str], fmts: Optional[Sequence[str]] = None, prefix: str = ""):
        self.names: List[str] = [str(n) for n in names]
        self.fmts: List[str] = list(fmts) if fmts is not None else ["%9.4e"] * len(self.names)
        self.prefix = prefix
        self.widths = [_cell_width(f, n) for f, n in zip(self.fmts, self.names)]

    def _frame(self, cells: Iterable[str]) -> str:
        return self.prefix + "|" + "|".join(cells) + "|"

    def make_row_sep(self) -> str:
        return "+" + "+".join("-" * w for w in self.widths) + "+"

    def make_footer(self) -> str:
        return self.prefix + self.make_row_sep()

    def make_header(self) -> str:
        rule = self.make_footer()
        titles = self._frame(_centre(n, w, True) for n, w in zip(self.names, self.widths))
        return "\n".join((rule, titles, rule))

    def make_values(self, vals) -> str:
        if len(vals) != len(self.fmts):
            raise AssertionError("one value per column")
        return self._frame(_centre(f % v, w, False) for v, f, w in zip(vals, self.fmts, self.widths))

    def print_header(self):
        print(self.make_header())

    def print_footer(self):
        print(self.make_footer())

    def print_values(self, vals):
        print(self.make_values(vals))


def atleast_nd(x: Optional[np.ndarray], n: int):
    """`x` with leading singleton axes up to `n` dimensions; `None` passes through (pmpc/utils.py:65-69)."""
    if x is None:
        return None
    arr = np.asarray(x)
    missing = n - arr.ndim
    return arr if missing <= 0 else arr[(None,) * missing]


def to_numpy_f64(x):
    """float64 ndarray view / copy of arrays, torch tensors (host or ROCm) and array-likes; Python scalars pass through
    (pmpc/utils.py:72-80)."""
    if isinstance(x, (float, int)):
        return x
    if hasattr(x, "detach"):  # torch tensor, possibly on the GPU
        x = x.detach().cpu().numpy()
    return np.asarray(x, dtype=np.float64)
