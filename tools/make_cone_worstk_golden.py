"""Golden cases for the worst-k cone objective with several costs on the threshold: the three problems of tools/fuzz/fuzz_cone.py
(seed 22, cases 37, 41, 74 — random problems, loose boxes, k < M) on which the ranking iteration failed or, before its acceptance
test was tightened, returned a non-optimal point.  Expected outputs: the line-cited restatement of the reference's cone program
(oracle/cone_oracle.py lcone_direct_py), computed here on the CPU.  Writes tests/golden/worstk_ties.npz.
usage: python tools/make_cone_worstk_golden.py   (regenerates the problems by replaying the fuzzer's random stream)"""
import sys

import numpy as np

sys.path.insert(0, ".")
from oracle import cone_oracle as co
from tests.support.problems import rand_problem

want = {37, 41, 74}
rng = np.random.default_rng(22)
out = {}
for case in range(80):  # the parameter draws of tools/fuzz/fuzz_cone.py, in its order
    M, N = int(rng.integers(2, 25)), int(rng.integers(3, 9))
    x, u = [(4, 2), (3, 2), (6, 3), (5, 2), (4, 3), (6, 2)][int(rng.integers(0, 6))]
    Nc = int(rng.choice([0, 1, 1, 2, -1]))
    bu = float(rng.choice([0.4, 1.0, 2.5]))
    bx = 5.0 if rng.random() < 0.25 else None
    kind = str(rng.choice(["hard", "hard", "logbarrier", "logbarrier", "squareplus"]))
    alpha = float("nan") if kind == "hard" else float(rng.choice([1.0, 10.0, 100.0]))
    k = None if rng.random() < 0.7 else int(rng.integers(1, M + 1))
    if kind == "squareplus":
        Nc, M = (Nc if Nc in (0, 1) else 1), min(M, 6)
    if k is not None:
        M = min(M, 8)
        k = min(k, M)
        k = None if k == M else k
    args, kw = rand_problem(rng, M, N, x, u, bu, bx)
    if case not in want:
        continue
    assert kind == "hard" and k is not None
    Xo, Uo = co.lcone_direct_py(*args, Nc=Nc, k=k, **kw)
    names = ["x0", "f", "fx", "fu", "X_prev", "U_prev", "Q", "R", "X_ref", "U_ref"]
    for n_, a_ in zip(names, args):
        out[f"c{case}_{n_}"] = a_
    for k_, v in kw.items():
        out[f"c{case}_kw_{k_}"] = np.asarray(v)
    out[f"c{case}_meta"] = np.array([Nc, k])
    out[f"c{case}_X"], out[f"c{case}_U"] = Xo, Uo
    print("case", case, "M", M, "N", N, "x", x, "u", u, "Nc", Nc, "k", k, flush=True)
np.savez_compressed("tests/golden/worstk_ties.npz", **out)
