"""Thread sweep of the structured CPU baseline (oracle/structured_cpu.c) on this host."""
import os, sys, time
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
sys.path.insert(0, ".")
import numpy as np
from oracle import lqp_oracle as orc
from pmpc_amd import dynamics as dyn
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    if os.path.exists(p):
        print(p, open(p).read().strip())
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
Ms = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
prob = dyn.make_quadrotor_problem(M=Ms, N=50, Nc=1)
X_ = np.concatenate([prob["x0"][:, None, :], prob["X_prev"][:, :-1]], 1)
f, fx, fu = prob["f_fx_fu_fn"](X_, prob["U_prev"])
for th in (1, 4, 8, 16, 32, 64, 128, 256):
    if th > len(os.sched_getaffinity(0)):
        break
    best = 1e9
    for rep in range(2):
        _, _, info = orc.structured_cpu_solve_py(prob["x0"], f, fx, fu, prob["X_prev"], prob["U_prev"], prob["Q"], prob["R"],
                                                 prob["X_ref"], prob["U_ref"], prob["reg_x"], prob["reg_u"], Nc=1,
                                                 u_l=prob["u_l"], u_u=prob["u_u"], threads=th)
        best = min(best, info["solve_s"])
    print(f"threads {th:4d}: {best:.3f}s for M={Ms} ({info['iters']} its) -> {best * 4096 / Ms:.2f}s at 4096")
