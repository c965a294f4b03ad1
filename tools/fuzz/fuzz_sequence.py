"""ONE context, a random SEQUENCE of unrelated problems through the host entry points (c_lqp_solve / c_lcone_solve semantics): a few
shapes that recur, QP and cone objective (hard boxes, log-barrier smoothing), boxes that bind or not, state boxes — every answer against
the oracle.  What one solve remembers (active sets, multipliers, weight assignments, smoothed iterates, which path worked for a shape)
must never leak into the next problem's ANSWER.   usage: fuzz_sequence.py SEED CALLS"""
import signal
import sys

import numpy as np

sys.path.insert(0, ".")
from oracle import lqp_oracle as orc
from pmpc_amd import backend
from tests.support.problems import abi_args, rand_problem

seed, calls = int(sys.argv[1]), int(sys.argv[2])
verbose_at = int(sys.argv[3]) if len(sys.argv) > 3 else -1  # (debugging: this call runs verbose)


class OracleTimeout(Exception):
    pass


def _alarm(*_):
    raise OracleTimeout()


signal.signal(signal.SIGALRM, _alarm)
rng = np.random.default_rng(seed)
shapes = [(6, 6, 4, 2), (6, 6, 4, 2), (12, 5, 6, 3), (3, 8, 12, 4), (9, 4, 3, 2)]
worst, fails, skipped = 0.0, 0, 0
for call in range(calls):
    M, N, x, u = shapes[int(rng.integers(0, len(shapes)))]
    Nc = int(rng.choice([1, 1, 2, -1, 0]))
    bu = float(rng.choice([0.3, 1.0, 3.0]))
    bx = 4.0 if rng.random() < 0.2 else None
    kind = str(rng.choice(["qp", "qp", "cone", "cone", "smooth"]))
    alpha = float(rng.choice([1.0, 10.0, 100.0])) if kind == "smooth" else float("nan")
    args, kw = rand_problem(rng, M, N, x, u, bu, bx)
    try:
        signal.alarm(30)
        if kind == "qp":
            Xo, Uo = orc.lqp_solve_py(*args, Nc=Nc, **kw)
        else:
            Xo, Uo = orc.lcone_solve_py(*args, Nc=Nc, smooth_alpha=alpha, **kw)
        signal.alarm(0)
    except BaseException:
        signal.alarm(0)
        skipped += 1
        continue
    if kind == "qp":
        X, U = backend.lqp_solve(*abi_args(args, kw, Nc))
    else:
        X, U = backend.lcone_solve(*abi_args(args, kw, Nc), smooth_alpha=alpha, solver="ecos", verbose=call == verbose_at)
    tag = f"call {call}: {kind} M{M} N{N} x{x} u{u} Nc{Nc} bu{bu} bx{bx} alpha {alpha}"
    if np.isnan(U).any():
        fails += 1
        print(tag + ": solver failed", flush=True)
        continue
    rel = lambda a_, b_: np.linalg.norm(a_ - b_) / max(np.linalg.norm(b_), 1.0)
    e = max(rel(X, Xo), rel(U, Uo))
    worst = max(worst, e)
    if e > 1e-6:
        fails += 1
        print(tag + f": rel err {e:.3e}", flush=True)
print(f"{calls} calls ({skipped} skipped), {fails} failures, worst rel err {worst:.3e}")
