"""PCIe-inclusive cost of the host-pointer ABI (c_lqp_solve) at config D's size."""
import sys, time, numpy as np
sys.path.insert(0, ".")
from pmpc_amd import backend, dynamics as dyn
from tests.support.problems import abi_args
M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
prob = dyn.make_quadrotor_problem(M=M, N=50)
X_lin = np.concatenate([prob["x0"][:, None, :], prob["X_prev"][:, :-1, :]], 1)
f, fx, fu = prob["f_fx_fu_fn"](X_lin, prob["U_prev"])
args = (prob["x0"], f, fx, fu, prob["X_prev"], prob["U_prev"], prob["Q"], prob["R"], prob["X_ref"], prob["U_ref"])
kw = dict(reg_x=prob["reg_x"], reg_u=prob["reg_u"], u_l=prob["u_l"], u_u=prob["u_u"])
a = abi_args(args, kw, 1)
nbytes = sum(v.nbytes for v in a if isinstance(v, np.ndarray))
for k in range(4):
    t = time.time()
    X, U = backend.lqp_solve(*a)
    dt = time.time() - t
    print(f"call {k}: {dt*1e3:.1f} ms  ({nbytes/1e6:.0f} MB of inputs -> {nbytes/dt/1e9:.1f} GB/s if it were all copy)")

# the C entry point alone: arrays already Fortran-contiguous (what pybind11's f_style cast / asfortranarray hands over)
pre = tuple(np.asfortranarray(v) if isinstance(v, np.ndarray) else v for v in a)
for k in range(4):
    t = time.time()
    X2, U2 = backend.lqp_solve(*pre)
    dt = time.time() - t
    print(f"C ABI only, call {k}: {dt*1e3:.1f} ms ({nbytes/dt/1e9:.1f} GB/s input rate)")
assert np.allclose(U, U2, rtol=0, atol=1e-8), np.abs(U - U2).max()

# an SCP loop through the host ABI: the linearisation (f, fx, fu) and X_prev / U_prev change every call, Q, R, the references and
# the boxes do not — their chunks are recognised as unchanged (memcmp against the bounce buffer) and not sent again
rng = np.random.default_rng(0)
names = ("x0", "f", "fx", "fu", "X_prev", "U_prev")
cur = list(pre)
for k in range(5):
    for idx in (2, 3, 4, 5, 6):  # abi_args order: (Nc, x0, f, fx, fu, X_prev, U_prev, Q, R, ...): the per-iteration arrays
        if isinstance(cur[idx], np.ndarray) and cur[idx].dtype == np.float64 and cur[idx].size > 1000:
            cur[idx] = np.asfortranarray(cur[idx] * (1.0 + 1e-6 * rng.standard_normal()))
    t = time.time()
    X3, U3 = backend.lqp_solve(*cur)
    dt = time.time() - t
    print(f"SCP-like call {k}: {dt*1e3:.1f} ms")
