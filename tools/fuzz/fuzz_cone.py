"""Random problems through the reference's default solver path (`c_lcone_solve` semantics: eps-anchored epigraph objective,
PMPC.jl/src/main.jl:204-316): hard boxes, log-barrier and squareplus smoothing at several alpha, worst-k, several consensus horizons,
control and state boxes.  Oracle: `lqp_oracle.lcone_solve_py` (fixed point over the cost ranking with exact weighted QPs — random costs
have no ties) for hard boxes and the log barrier; the line-cited restatement of the cone program (`cone_oracle.lcone_direct_py`, slow) for
squareplus and for the worst-k objective (k < M puts many costs on the threshold; the ranking oracle knows two-way ties only) at small M.
With k < M the particles below the threshold carry no multiplier and the reference's minimiser is not unique in their free variables: there
the comparison is the shared controls, the particles on or above the threshold and the value of the objective.
usage: fuzz_cone.py SEED CASES [MAX_M = 24] [MAX_M_DIRECT = 8]"""
import signal
import sys

import numpy as np

sys.path.insert(0, ".")
from oracle import cone_oracle as co
from oracle import lqp_oracle as orc
from pmpc_amd import backend
from tests.support.problems import abi_args, rand_problem

seed, cases = int(sys.argv[1]), int(sys.argv[2])
max_M = int(sys.argv[3]) if len(sys.argv) > 3 else 24
max_M_direct = int(sys.argv[4]) if len(sys.argv) > 4 else 8  # particle count of the cases that need the (slow) direct cone program


class OracleTimeout(Exception):
    pass


def _alarm(*_):
    raise OracleTimeout()


signal.signal(signal.SIGALRM, _alarm)
rng = np.random.default_rng(seed)
worst, fails, skipped, by_kind = 0.0, 0, 0, {}
for case in range(cases):
    M, N = int(rng.integers(2, max_M + 1)), int(rng.integers(3, 9))
    x, u = [(4, 2), (3, 2), (6, 3), (5, 2), (4, 3), (6, 2)][int(rng.integers(0, 6))]
    Nc = int(rng.choice([0, 1, 1, 2, -1]))
    bu = float(rng.choice([0.4, 1.0, 2.5]))
    bx = 5.0 if rng.random() < 0.25 else None
    kind = str(rng.choice(["hard", "hard", "logbarrier", "logbarrier", "squareplus"]))
    alpha = float("nan") if kind == "hard" else float(rng.choice([1.0, 10.0, 100.0]))
    k = None if rng.random() < 0.7 else int(rng.integers(1, M + 1))
    if kind == "squareplus":
        Nc, M = (Nc if Nc in (0, 1) else 1), min(M, 6, max_M_direct)
    if k is not None:
        M = min(M, max_M_direct)
        k = min(k, M)
        k = None if k == M else k
    args, kw = rand_problem(rng, M, N, x, u, bu, bx)
    okw = dict(kw)
    if k is not None:
        okw["k"] = k
    try:
        signal.alarm(20)
        if kind == "squareplus" or k is not None:
            Xo, Uo = co.lcone_direct_py(*args, Nc=Nc, **(dict(smooth_alpha=alpha, smooth_cstr=kind) if kind != "hard" else {}), **okw)
        else:
            Xo, Uo = orc.lcone_solve_py(*args, Nc=Nc, smooth_alpha=alpha, **okw)
        signal.alarm(0)
    except BaseException as e:  # (the oracle's own failure — infeasible state boxes, a stalled path-following run — is not a case)
        signal.alarm(0)
        skipped += 1
        continue
    skw = {} if kind == "hard" else dict(smooth_cstr=kind, smooth_beta=1.0)
    X, U = backend.lcone_solve(*abi_args(args, kw, Nc), smooth_alpha=alpha, solver="ecos", k=k, **skw)
    tag = f"case {case}: M{M} N{N} x{x} u{u} Nc{Nc} bu{bu} bx{bx} {kind} alpha {alpha} k {k}"
    if np.isnan(U).any():
        fails += 1
        print(tag + ": solver failed", flush=True)
        continue
    rel = lambda a_, b_: np.linalg.norm(a_ - b_) / max(np.linalg.norm(b_), 1.0)
    if k is None:
        e = max(rel(X, Xo), rel(U, Uo))
    else:
        x0_, f_, fx_, fu_, X_prev_, U_prev_, Q_, R_, X_ref_, U_ref_ = args
        Jg, Jo = (orc.particle_costs_py(X_, U_, X_prev_, U_prev_, Q_, R_, X_ref_, U_ref_, reg_x=kw["reg_x"], reg_u=kw["reg_u"]) for X_, U_ in ((X, U), (Xo, Uo)))
        cone_obj = lambda Jv: min((1 + 1e-3) * np.sum(np.maximum(Jv - t_, 0.0)) + (1 - 1e-3) * k * t_ for t_ in Jv)
        top = Jo >= np.sort(Jo)[::-1][k - 1] - 1e-9 * max(1.0, np.abs(Jo).max())  # on or above the threshold
        Ncc = U.shape[1] if Nc < 0 else Nc
        e = max(rel(U[:, :Ncc], Uo[:, :Ncc]), rel(X[top], Xo[top]), rel(U[top], Uo[top]))
        if kind == "hard":
            e = max(e, abs(cone_obj(Jg) - cone_obj(Jo)) / max(1.0, abs(cone_obj(Jo))))
    worst = max(worst, e)
    by_kind[kind] = max(by_kind.get(kind, 0.0), e)
    if e > 1e-6:
        fails += 1
        print(tag + f": rel err {e:.3e}", flush=True)
print("worst per kind:", {k_: f"{v:.2e}" for k_, v in by_kind.items()})
print(f"{cases} cases ({skipped} skipped), {fails} failures, worst rel err {worst:.3e}")
