"""ONE context, a random SEQUENCE of unrelated problems through the host entry points (c_lqp_solve / c_lcone_solve semantics, aff_solve
with extra_cstrs rows on states): a few shapes that recur, QP (with and without slew penalties) and cone objective (hard boxes, log-barrier
smoothing), boxes that bind or not, state boxes — every answer against the oracle.  What one solve remembers (active sets, multipliers, weight assignments, smoothed iterates, which path worked for a shape)
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
    bx = 6.0 if rng.random() < 0.2 else None  # (tighter state boxes make some random problems infeasible: the oracle's sparse LU then meets a singular KKT matrix and scipy segfaults)
    kind = str(rng.choice(["qp", "qp", "cone", "cone", "smooth", "slew", "rows"]))
    alpha = float(rng.choice([1.0, 10.0, 100.0])) if kind == "smooth" else float("nan")
    args, kw = rand_problem(rng, M, N, x, u, bu, bx, *((0.5, 0.3) if kind == "slew" else ()))
    rows = None
    if kind == "rows":  # a few extra_cstrs rows on states (obstacle half-spaces) that cut into the optimum without them
        import scipy.sparse as sp

        Ncc = N if Nc < 0 else Nc
        ncu = Ncc * u + M * (N - Ncc) * u
        n = ncu + M * N * x
        try:
            X0, U0 = orc.lqp_solve_py(*args, Nc=Nc, **kw)
        except BaseException:
            skipped += 1
            continue
        pairs = [(i, t_) for i in range(M) for t_ in range(N)]
        rng.shuffle(pairs)
        pairs = pairs[:int(rng.integers(1, M + 1))]
        G, h = np.zeros((len(pairs), n)), np.zeros(len(pairs))
        for k_, (i, t_) in enumerate(pairs):
            a_ = rng.standard_normal(x)
            G[k_, ncu + (i * N + t_) * x: ncu + (i * N + t_ + 1) * x] = a_
            h[k_] = a_ @ X0[i, t_] - 0.1 * rng.random()
        rows = (sp.csr_matrix(G), h)
    try:
        signal.alarm(30)
        if kind in ("qp", "slew", "rows"):
            Xo, Uo = orc.lqp_solve_py(*args, Nc=Nc, rows=rows, **kw)
        else:
            Xo, Uo = orc.lcone_solve_py(*args, Nc=Nc, smooth_alpha=alpha, **kw)
        signal.alarm(0)
    except BaseException:
        signal.alarm(0)
        skipped += 1
        continue
    if kind in ("qp", "slew"):
        X, U = backend.lqp_solve(*abi_args(args, kw, Nc))
    elif kind == "rows":
        x0_, f_, fx_, fu_, X_prev_, U_prev_, Q_, R_, X_ref_, U_ref_ = args
        tup = (len(rows[1]), [], 0, rows[0], sp.csr_matrix((len(rows[1]), 0)), rows[1], np.zeros(n), np.zeros(0))
        X, U, _ = backend.aff_solve(f_, fx_, fu_, x0_, X_prev_, U_prev_, Q_, R_, X_ref_, U_ref_, kw["reg_x"], kw["reg_u"], None, None, kw.get("x_l"), kw.get("x_u"),
                                    kw.get("u_l"), kw.get("u_u"), solver_settings=dict(solver="osqp", Nc=Nc, extra_cstrs=[tup]))
        X = X[:, 1:]
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
