"""The reference's canonical ABI example (tests/pmpcjl_test.py:164-219) through the device API, verbose."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from pmpc_amd.device import DeviceSolver
from tests.test_oracle_golden import load_qp
from tests.test_slew_gpu import _solve, _rel
args, kw, Nc, Xg, Ug, _ = load_qp("qp_double_integrator_u04.npz")
s = DeviceSolver(0)
for rep in range(2):
    X, U, status, info = _solve(s, args, kw, Nc, verbose=1)
    print(status, info, _rel(X, Xg), _rel(U, Ug))
import time, torch
for name, opt in (("increment form + state rows", 1), ("generic kernels", 0)):
    s = DeviceSolver(0)
    s.set_option("slew_increment_boxes", opt)
    s.set_option("warn_slow_path", 0)
    for cold in (True, False):
        ts = []
        for rep in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            X, U, status, info = _solve(s, args, kw, Nc, cold_start=cold)
            ts.append(time.perf_counter() - t0)
        print(f"{name:28s} cold={cold}: {1e3 * min(ts[1:]):.3f} ms  ipm {info['ipm_iters']} rounds {info['active_set_rounds']} factorisations {info['structured_solves']} err {_rel(X, Xg):.1e}")
    s.close()
