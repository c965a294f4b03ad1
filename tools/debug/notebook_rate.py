import sys, time, numpy as np
sys.path.insert(0, ".")
import pmpc_amd
from tests.support import notebook_problem as nbp
args, kw, table = nbp.load()
for solver in ("ecos", "osqp"):
    pmpc_amd.solve(*args, solver_settings=dict(solver=solver), **dict(kw, max_it=3))
    t = time.time()
    X, U, data = pmpc_amd.solve(*args, solver_settings=dict(solver=solver), **kw)
    dt = time.time() - t
    el = [h["elaps"] for h in data["hist"]]
    print(solver, "50 SCP iterations in %.3f s -> %.1f it/s; steady per-iteration %.3f ms; aff_solve mean %.3f ms" % (dt, 50 / dt, 1e3 * (el[-1] - el[9]) / 40, 1e3 * np.mean(data["t_aff_solve"][10:])))
