"""Random problems solved on ONE rank and sharded over 2 .. 4 mock ranks on one GPU (threads of this process, the library's mock
communicator standing in for RCCL): QP path and cone objective (hard boxes; log-barrier / squareplus smoothing with the optional third argument
"smooth": every cone case smoothed), control / state boxes, slew penalties, several consensus horizons, cold and warm.  Sharding must not change the answer (same arithmetic, sums in another order: 1e-9) nor the shared controls' equality
across ranks (bitwise).   usage: fuzz_sharded.py SEED CASES [smooth]"""
import sys

import numpy as np

sys.path.insert(0, ".")
from tests.support.problems import rand_problem
from tests.test_multirank_gpu import _solve_sharded

seed, cases = int(sys.argv[1]), int(sys.argv[2])
smooth_mode = len(sys.argv) > 3 and sys.argv[3] == "smooth"
rng = np.random.default_rng(seed)
worst, fails, skipped = 0.0, 0, 0
for case in range(cases):
    world = int(rng.choice([2, 3, 4]))
    Ml = int(rng.integers(1, 7))
    M, N = world * Ml, int(rng.integers(3, 10))
    x, u = [(4, 2), (3, 2), (6, 3), (5, 2), (4, 3), (12, 4)][int(rng.integers(0, 6))]
    Nc = int(rng.choice([0, 1, 1, 2, -1]))
    bu = float(rng.choice([0.3, 0.8, 2.0])) if rng.random() < 0.85 else None
    bx = 3.0 if rng.random() < 0.25 else None
    slew = 0.5 if rng.random() < 0.2 else None
    cone = bool(rng.random() < (0.9 if smooth_mode else 0.35)) and M > 1 and slew is None
    smooth = None
    if smooth_mode and cone and bu is not None:
        smooth = dict(smooth_alpha=float(rng.choice([1.0, 10.0, 100.0])))
        if rng.random() < 0.3 and Nc in (0, 1):
            smooth.update(smooth_cstr="squareplus", smooth_beta=float(rng.choice([1.0, 5.0])))
    repeats = int(rng.choice([1, 2]))
    args, kw = rand_problem(rng, M, N, x, u, bu, bx, slew)
    tag = f"case {case}: world {world} M{M} N{N} x{x} u{u} Nc{Nc} bu{bu} bx{bx} slew{slew} cone{cone} smooth{smooth} repeats{repeats}"
    try:
        X1, U1, i1 = _solve_sharded(args, kw, Nc, 1, repeats=repeats, cone=cone, smooth=smooth)
    except AssertionError:  # (the helper asserts status 0: a problem one rank cannot solve — infeasible state boxes — is not a case)
        skipped += 1
        continue
    try:
        Xw, Uw, iw = _solve_sharded(args, kw, Nc, world, repeats=repeats, cone=cone, smooth=smooth)
    except AssertionError as e:
        fails += 1
        print(tag + f": solved on one rank, not sharded: {str(e)[:120]}", flush=True)
        continue
    rel = lambda a_, b_: np.linalg.norm(a_ - b_) / max(np.linalg.norm(b_), 1.0)
    e = max(rel(Xw, X1), rel(Uw, U1))
    Ncc = N if Nc < 0 else Nc
    shared_ok = bool(np.all(Uw[:, :Ncc] == Uw[0:1, :Ncc]))
    worst = max(worst, e)
    if e > 1e-8 or not shared_ok:
        fails += 1
        print(tag + f": rel err {e:.3e} shared controls equal {shared_ok}", flush=True)
print(f"{cases} cases ({skipped} skipped), {fails} failures, worst rel err {worst:.3e}")
