"""ORACLE (test infrastructure, not product code) for the `c_lqp_solve` hot path.

A CPU restatement of what the reference does for one convex sub-problem
(PMPC.jl/src/main.jl:115-171 `lqp_solve`):

  1. assemble the joint sparse QP  (P,q), (A,b), (G,l,u)  exactly as
     PMPC.jl/src/lqp_utils.jl:2-393 does            -> oracle/lqp_assemble.c (ctypes)
  2. hand it to OSQP: `[A;G] z in [[b;l],[b;u]]`    (PMPC.jl/src/osqp_solver.jl:20-72)
  3. scatter z back into X,U (lqp_utils.jl:395-423) -> split_lqp_vars

The numeric solver of step 2 is a third-party dependency that is NOT in
/root/reference: OSQP.jl 0.8.0 -> OSQP_jll 0.600.200 (OSQP C 0.6.2, ADMM + QDLDL;
PMPC.jl/Manifest.toml:294-304).  Its published algorithm (Stellato et al., "OSQP: an
operator splitting solver for quadratic programs", Alg. 1 + the rho_eq = 1e3*rho rule
+ adaptive rho + polish) is restated in `osqp_admm` below with scipy's SuperLU in
place of QDLDL.  Because OSQP at its default eps=1e-3 only reaches z* to ~1e-3, the
oracle's answer is the UNIQUE optimum z* of the strictly convex QP: `solve_qp_exact`
refines the ADMM guess by an active-set KKT solve and returns a KKT certificate
(stationarity, primal feasibility, multiplier signs, complementarity).

OSQP.jl keeps only the upper triangle of P (`P = triu(P)` in its `setup!`), so the
effective Hessian is triu(P) + triu(P,1)'.  `effective_P` applies that; for the
symmetric Q,R every reference problem uses it is the identity.

PARITY PINNED ONLY WEAKLY: the reference's own tests hold no numeric golden vector for
this path (PMPC.jl/test/runtests.jl:33-41 only asserts !isnan; tests/pmpcjl_test.py has no
asserts) and neither Julia nor OSQP can run in the build container.  The oracle is
pinned by (a) its KKT certificate on every solve, (b) the reference's own Python SCP
loop (pmpc/scp_mpc.py:205-442) run over it to make tests/golden/*.npz, and (c) the one
output of the reference's own Julia + ECOS stack that exists for this path: the 50-row
(obj, resid) table stored in examples/gpu_solver.ipynb (M = 1, N = 20, |u| <= 1), which
this oracle + the host loop reproduce to its 4 printed digits
(tests/golden/ref_notebook_cpu_table.npz, tests/test_host_logic.py).  The branches that
table does not exercise — consensus (M > 1), slew, state bounds, the cone weights for
M > 1, smoothing — remain UNPINNED against reference output.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Layouts.  "py" layout = the reference's Python-side arrays (pmpc/scp_mpc.py:297-321):
x0 (M,x), f (M,N,x), fx (M,N,x,x) with fx[i,j,r,t] = dF_r/dx_t, fu (M,N,x,u),
Q (M,N,x,x), R (M,N,u,u), vectors (M,N,d).  "abi" layout = the C-ABI buffers
(PMPC.jl/src/c_interface.jl:28-46): the same vectors, and matrices with their last two
axes swapped (column-major blocks) — pmpc/julia_utils.py:69-77 + asfortranarray.
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess
import time
from pathlib import Path
from typing import Optional

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

_HERE = Path(__file__).resolve().parent
_SO = _HERE / "liblqp_oracle.so"
_lib = None


def build(force: bool = False) -> Path:
    """gcc-compile the C restatement (recipe also in oracle/Makefile)."""
    src = _HERE / "lqp_assemble.c"
    if force or not _SO.exists() or _SO.stat().st_mtime < src.stat().st_mtime:
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-std=c99", "-o", str(_SO), str(src)])
    return _SO


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(str(_SO))
        _lib.lqp_repr_Pq.restype = ctypes.c_longlong
        _lib.lqp_repr_Ab.restype = ctypes.c_longlong
        _lib.lqp_repr_Gla.restype = ctypes.c_longlong
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


_SO_CPU = _HERE / "libstructured_cpu.so"
_lib_cpu = None


def build_structured_cpu(force: bool = False) -> Path:
    """gcc -fopenmp build of oracle/structured_cpu.c (the multi-core structured CPU baseline; recipe also in oracle/Makefile).
    AVX2 + FMA only (no -march=native): the .so is built in the dev container and travels to the GPU box."""
    src = _HERE / "structured_cpu.c"
    if force or not _SO_CPU.exists() or _SO_CPU.stat().st_mtime < src.stat().st_mtime:
        subprocess.check_call(["gcc", "-O3", "-mavx2", "-mfma", "-fopenmp", "-fPIC", "-shared", "-std=gnu99", "-o", str(_SO_CPU),
                               str(src), "-lm"])
    return _SO_CPU


def structured_cpu_solve_py(x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref, reg_x, reg_u, Nc=-1, x_l=None, x_u=None, u_l=None,
                            u_u=None, threads=0):
    """The same QP as `lqp_solve_py` (py layout in, py layout out) through oracle/structured_cpu.c: Riccati + condensing +
    Mehrotra on the boxes, OpenMP over particles, and — control boxes only — one exact Newton step on the active set the
    interior-point iterate names (`polished`: the answer then agrees with the exact oracle to ~1e-15 instead of the ~1e-6 an
    interior-point iterate leaves at weakly active boxes).  No slew terms.  Returns X (M,N,x), U (M,N,u), info."""
    global _lib_cpu
    if _lib_cpu is None:
        build_structured_cpu()
        _lib_cpu = ctypes.CDLL(str(_SO_CPU))
        _lib_cpu.structured_cpu_solve.restype = ctypes.c_int
    f = _f64(f)
    M, N, x = f.shape
    u = np.asarray(fu).shape[-1]
    bc = lambda a, d: None if a is None else _f64(np.broadcast_to(np.asarray(a, float), (M, N, d)))
    arrs = [_f64(x0), f, to_abi_mat(fx), to_abi_mat(fu), _f64(X_prev), _f64(U_prev), to_abi_mat(Q), to_abi_mat(R), _f64(X_ref),
            _f64(U_ref)]
    bnd = [bc(x_l, x), bc(x_u, x), bc(u_l, u), bc(u_u, u)]
    X, U = np.empty((M, N, x)), np.empty((M, N, u))
    iters = ctypes.c_int(0)
    t0 = time.perf_counter()
    st = _lib_cpu.structured_cpu_solve(
        ctypes.c_int(x), ctypes.c_int(u), ctypes.c_int(N), ctypes.c_int(M), ctypes.c_longlong(int(Nc)), *[_p(a) for a in arrs],
        *[None if b is None else _p(b) for b in bnd], ctypes.c_double(reg_x), ctypes.c_double(reg_u), ctypes.c_int(int(threads)),
        _p(X), _p(U), ctypes.byref(iters))
    it = int(iters.value)
    return X, U, dict(status=int(st), iters=it % 1000, polished=it >= 1000, solve_s=time.perf_counter() - t0)


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def to_abi_mat(A):
    """py (..., row, col) -> abi memory (..., col, row)."""
    return np.ascontiguousarray(np.swapaxes(np.asarray(A, dtype=np.float64), -1, -2))


# -------------------------------------------------------------------------------------------------
# sentinels: PMPC.jl/src/c_interface.jl:56-70
# -------------------------------------------------------------------------------------------------
def unwrap_sentinels(M, udim, lx, ux, lu, uu, slew_reg, slew_reg0, slew_um1):
    has_xb = not (np.any(np.isnan(lx)) or np.any(np.isnan(ux)))
    has_ub = not (np.any(np.isnan(lu)) or np.any(np.isnan(uu)))
    # absent slew_reg -> make_probs default 0.0 (main.jl:23-25)
    sr = np.zeros(M) if np.any(np.isnan(slew_reg)) else _f64(slew_reg).reshape(M)
    if np.any(np.isnan(slew_reg0)) or np.any(np.isnan(slew_um1)):
        sr0, um1 = np.zeros(M), np.zeros((M, udim))  # main.jl:23-27 defaults
    else:
        sr0, um1 = _f64(slew_reg0).reshape(M), _f64(slew_um1).reshape(M, udim)
    return has_xb, has_ub, sr, sr0, um1


# -------------------------------------------------------------------------------------------------
# assembly (ctypes into lqp_assemble.c)
# -------------------------------------------------------------------------------------------------
class JointQP:
    __slots__ = ("P", "q", "A", "b", "G", "l", "u", "dims", "Nc", "t_assemble")


def assemble_abi(
    xdim, udim, N, M, Nc, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref, lx, ux, lu, uu, reg_x, reg_u,
    slew_reg, slew_reg0, slew_um1, weights=None,
) -> JointQP:
    """All array arguments are float64 buffers in ABI layout (any shape, C-contiguous memory)."""
    lib = _load()
    if Nc > N:  # the reference indexes out of bounds here (lqp_utils.jl loops 1:Nc over N-long arrays); the C restatement would too
        raise ValueError(f"Nc = {Nc} exceeds the horizon N = {N}")
    t0 = time.perf_counter()
    f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = map(_f64, (f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref))
    lx, ux, lu, uu = map(_f64, (lx, ux, lu, uu))
    has_xb, has_ub, sr, sr0, um1 = unwrap_sentinels(M, udim, lx, ux, lu, uu, slew_reg, slew_reg0, slew_um1)
    sr, sr0, um1 = map(_f64, (sr, sr0, um1))
    ll = ctypes.c_longlong
    n, m_eq, m_in, nnzP, nnzA = ll(), ll(), ll(), ll(), ll()
    sz = ctypes.c_size_t
    dims = (sz(xdim), sz(udim), sz(N), sz(M), ll(Nc))
    lib.lqp_sizes(*dims, ctypes.c_int(has_ub), ctypes.c_int(has_xb), *(ctypes.byref(v) for v in (n, m_eq, m_in, nnzP, nnzA)))
    n, m_eq, m_in, nnzP, nnzA = (v.value for v in (n, m_eq, m_in, nnzP, nnzA))

    Pp, Pi, Px, q = np.zeros(n + 1, np.int64), np.zeros(nnzP, np.int64), np.zeros(nnzP), np.zeros(n)
    wts = None if weights is None else _f64(weights).reshape(M)
    k = lib.lqp_repr_Pq(*dims, _p(X_prev), _p(U_prev), _p(Q), _p(R), _p(X_ref), _p(U_ref),
                        ctypes.c_double(reg_x), ctypes.c_double(reg_u), _p(sr), _p(sr0), _p(um1),
                        _p(Pp), _p(Pi), _p(Px), _p(q), _p(wts) if wts is not None else None)
    assert 0 <= k <= nnzP
    P = sp.csc_matrix((Px[:k], Pi[:k], Pp), shape=(n, n))

    Ap, Ai, Ax, b = np.zeros(n + 1, np.int64), np.zeros(nnzA, np.int64), np.zeros(nnzA), np.zeros(m_eq)
    k = lib.lqp_repr_Ab(*dims, _p(f), _p(fx), _p(fu), _p(X_prev), _p(U_prev), _p(Ap), _p(Ai), _p(Ax), _p(b))
    assert k == nnzA
    A = sp.csc_matrix((Ax, Ai, Ap), shape=(m_eq, n))

    Gp, Gi, Gx = np.zeros(n + 1, np.int64), np.zeros(m_in, np.int64), np.zeros(m_in)
    l, u = np.zeros(m_in), np.zeros(m_in)
    k = lib.lqp_repr_Gla(*dims, ctypes.c_int(has_ub), ctypes.c_int(has_xb), _p(lx), _p(ux), _p(lu), _p(uu),
                         _p(Gp), _p(Gi), _p(Gx), _p(l), _p(u))
    assert k == m_in
    G = sp.csc_matrix((Gx, Gi, Gp), shape=(m_in, n))

    qp = JointQP()
    qp.P, qp.q, qp.A, qp.b, qp.G, qp.l, qp.u = P, q, A, b, G, l, u
    qp.dims, qp.Nc = (xdim, udim, N, M), (Nc if Nc >= 0 else N)
    qp.t_assemble = time.perf_counter() - t0
    return qp


def split_vars(qp: JointQP, z):
    """split_lqp_vars (lqp_utils.jl:395-423) -> X (M,N,x), U (M,N,u) (== ABI (x,N,M), (u,N,M) memory)."""
    lib = _load()
    xdim, udim, N, M = qp.dims
    z = _f64(z)
    X, U = np.zeros((M, N, xdim)), np.zeros((M, N, udim))
    sz, ll = ctypes.c_size_t, ctypes.c_longlong
    lib.split_lqp_vars(sz(xdim), sz(udim), sz(N), sz(M), ll(qp.Nc), _p(z), _p(X), _p(U))
    return X, U


def effective_P(P):
    """OSQP.jl `setup!` keeps triu(P); the Hessian OSQP minimises is triu(P) + triu(P,1)'."""
    Pu = sp.triu(P, format="csc")
    return (Pu + sp.triu(P, k=1, format="csc").T).tocsc()


# -------------------------------------------------------------------------------------------------
# restated OSQP (ADMM) — the reference's numeric solver (osqp_solver.jl:34-72 -> OSQP C 0.6.2)
# -------------------------------------------------------------------------------------------------
def osqp_admm(P, q, Aa, la, ua, eps_abs=1e-3, eps_rel=1e-3, max_iter=4000, rho=0.1, sigma=1e-6, alpha=1.6,
              adaptive_rho=True, check_every=25, x0=None, y0=None):
    """OSQP Algorithm 1 with the published defaults (rho=0.1, sigma=1e-6, alpha=1.6, rho_eq=1e3*rho,
    eps_abs=eps_rel=1e-3, max_iter=4000, adaptive rho).  Problem scaling (Ruiz) is not restated: it
    changes the iteration path, not the fixed point.  Returns x, y, info."""
    n, m = P.shape[0], Aa.shape[0]
    eq = (ua - la) <= 1e-12 * np.maximum(1.0, np.abs(la))
    x = np.zeros(n) if x0 is None else x0.copy()
    y = np.zeros(m) if y0 is None else y0.copy()
    z = np.clip(Aa @ x, la, ua)
    AaT = Aa.T.tocsc()
    nfact = 0

    def factor(rho_):
        rho_vec = np.where(eq, 1e3 * rho_, rho_)
        K = sp.bmat([[P + sigma * sp.identity(n), AaT], [Aa, -sp.diags(1.0 / rho_vec)]], format="csc")
        return spla.splu(K), rho_vec

    lu, rho_vec = factor(rho)
    nfact += 1
    it, status = 0, "max_iter"
    for it in range(1, max_iter + 1):
        rhs = np.concatenate([sigma * x - q, z - y / rho_vec])
        if not np.isfinite(rhs).all() or np.max(np.abs(rhs)) > 1e30:
            raise FloatingPointError("restated OSQP diverged: the QP is primal or dual infeasible")
        sol = lu.solve(rhs)
        xt, nu = sol[:n], sol[n:]
        zt = z + (nu - y) / rho_vec
        x = alpha * xt + (1 - alpha) * x
        zr = alpha * zt + (1 - alpha) * z
        z_new = np.clip(zr + y / rho_vec, la, ua)
        y = y + rho_vec * (zr - z_new)
        z = z_new
        if it % check_every == 0 or it == max_iter:
            if not (np.all(np.isfinite(x)) and np.max(np.abs(x)) < 1e15):
                raise FloatingPointError("restated OSQP diverged: the QP is primal or dual infeasible")
            Ax, Px, Aty = Aa @ x, P @ x, AaT @ y
            rp, rd = np.linalg.norm(Ax - z, np.inf), np.linalg.norm(Px + q + Aty, np.inf)
            ep = eps_abs + eps_rel * max(np.linalg.norm(Ax, np.inf), np.linalg.norm(z, np.inf))
            ed = eps_abs + eps_rel * max(np.linalg.norm(Px, np.inf), np.linalg.norm(Aty, np.inf), np.linalg.norm(q, np.inf))
            if rp <= ep and rd <= ed:
                status = "solved"
                break
            if adaptive_rho:
                num = rp / max(np.linalg.norm(Ax, np.inf), np.linalg.norm(z, np.inf), 1e-30)
                den = rd / max(np.linalg.norm(Px, np.inf), np.linalg.norm(Aty, np.inf), np.linalg.norm(q, np.inf), 1e-30)
                rho_new = float(np.clip(rho * math.sqrt(num / max(den, 1e-30)), 1e-6, 1e6))
                if rho_new > 5 * rho or rho_new < rho / 5:
                    rho = rho_new
                    lu, rho_vec = factor(rho)
                    nfact += 1
    return x, y, dict(iters=it, status=status, rho=rho, factorizations=nfact)


# -------------------------------------------------------------------------------------------------
# exact optimum + KKT certificate
# -------------------------------------------------------------------------------------------------
def _kkt_solve(P, A, GW, rhs_z, rhs_eq, rhs_w, refine=3):
    n, me, mw = P.shape[0], A.shape[0], GW.shape[0]
    blocks = [[P, A.T, GW.T if mw else None], [A, None, None]]
    if mw:
        blocks.append([GW, None, None])
    else:
        blocks = [[P, A.T], [A, None]]
    K = sp.bmat(blocks, format="csc")
    rhs = np.concatenate([rhs_z, rhs_eq, rhs_w])
    try:
        lu = spla.splu(K)
    except RuntimeError:
        # degenerate active set (redundant active rows): regularise like OSQP's polish
        # (delta = 1e-6) and recover the un-regularised solution by iterative refinement
        delta = 1e-7
        reg = sp.diags(np.concatenate([np.full(n, delta), np.full(me + mw, -delta)]))
        lu = spla.splu((K + reg).tocsc())
        refine = 30
    sol = lu.solve(rhs)
    for _ in range(refine):
        sol += lu.solve(rhs - K @ sol)
    return sol[:n], sol[n:n + me], sol[n + me:]


def kkt_certificate(P, q, A, b, G, l, u, z, y, mu):
    """mu > 0 pushes against the upper bound, mu < 0 against the lower one."""
    Gz = G @ z if G.shape[0] else np.zeros(0)
    r_stat = P @ z + q + A.T @ y + (G.T @ mu if G.shape[0] else 0.0)
    r_eq = A @ z - b
    viol = np.maximum(np.maximum(l - Gz, Gz - u), 0.0) if G.shape[0] else np.zeros(0)
    comp = np.minimum(np.maximum(mu, 0.0), np.abs(u - Gz)) + np.minimum(np.maximum(-mu, 0.0), np.abs(Gz - l)) if G.shape[0] else np.zeros(0)
    comp = np.where(np.isfinite(comp), comp, 0.0)
    mx = lambda v: float(np.max(np.abs(v))) if np.size(v) else 0.0
    return dict(stationarity=mx(r_stat), equality=mx(r_eq), bound_violation=mx(viol), complementarity=mx(comp))


def solve_qp_exact(qp: JointQP, tol=1e-9, verbose=False):
    """Unique optimum z* of the strictly convex joint QP + its KKT certificate."""
    P, q, A, b, G, l, u = effective_P(qp.P), qp.q, qp.A, qp.b, qp.G, qp.l, qp.u
    n, m_in = P.shape[0], G.shape[0]
    if m_in == 0:
        z, y, _ = _kkt_solve(P, A, sp.csc_matrix((0, n)), -q, b, np.zeros(0))
        cert = kkt_certificate(P, q, A, b, G, l, u, z, y, np.zeros(0))
        assert max(cert.values()) <= tol * max(1.0, float(np.max(np.abs(q)))), cert
        return z, dict(cert=cert, admm=None, active_set_iters=0)

    # 1) the reference's algorithm (ADMM) to a moderate tolerance -> active-set guess
    Aa = sp.vstack([A, G], format="csc")  # augmented_A, osqp_solver.jl:20-32
    la, ua = np.concatenate([b, l]), np.concatenate([b, u])
    x_admm, y_admm, info = osqp_admm(P, q, Aa, la, ua, eps_abs=1e-6, eps_rel=1e-6, max_iter=20000)
    mu = y_admm[A.shape[0]:]
    Gz = G @ x_admm
    scale = np.maximum(1.0, np.abs(Gz))
    lower = (Gz - l < -mu) | (Gz < l + 1e-9 * scale)  # OSQP polish active-set rule
    upper = (u - Gz < mu) | (Gz > u - 1e-9 * scale)
    Gr = G.tocsr()

    # 2) primal-dual active-set refinement on the sparse KKT system
    z = y = None
    mu_full = np.zeros(m_in)
    seen = set()
    for it in range(200):
        both = lower & upper
        W = np.flatnonzero(lower | upper)
        v = np.where(upper[W] & ~lower[W], u[W], l[W])
        z, y, muW = _kkt_solve(P, A, Gr[W].tocsc(), -q, b, v)
        mu_full = np.zeros(m_in)
        mu_full[W] = muW
        Gz = G @ z
        bad_lower = lower & ~both & (mu_full > 1e-12)   # lower-active needs mu <= 0
        bad_upper = upper & ~both & (mu_full < -1e-12)  # upper-active needs mu >= 0
        add_lower = ~lower & (Gz < l - 1e-11 * np.maximum(1.0, np.abs(l)))
        add_upper = ~upper & (Gz > u + 1e-11 * np.maximum(1.0, np.abs(u)))
        if not (bad_lower.any() or bad_upper.any() or add_lower.any() or add_upper.any()):
            break
        lower = (lower & ~bad_lower) | add_lower
        upper = (upper & ~bad_upper) | add_upper
        key = (lower.tobytes(), upper.tobytes())
        if key in seen:  # cycling: restart the guess from a tighter ADMM run
            x_admm, y_admm, info = osqp_admm(P, q, Aa, la, ua, eps_abs=1e-9, eps_rel=1e-9, max_iter=100000,
                                             x0=x_admm, y0=y_admm)
            mu = y_admm[A.shape[0]:]
            Gz = G @ x_admm
            lower, upper = (Gz - l < -mu), (u - Gz < mu)
            seen.clear()
        seen.add(key)
    cert = kkt_certificate(P, q, A, b, G, l, u, z, y, mu_full)
    if verbose:
        print("oracle: admm", info, "active-set iters", it, cert)
    assert max(cert.values()) <= tol * max(1.0, float(np.max(np.abs(q)))), cert
    return z, dict(cert=cert, admm=info, active_set_iters=it)



def solve_barrier_exact(qp: JointQP, mu, tol=1e-11, max_iter=200, verbose=False):
    """Minimiser of  1/2 z'Pz + q'z - mu sum log(Gz - l) - mu sum log(u - Gz)  s.t.  Az = b  — what the cone path
    solves when `smooth_cstr = "logbarrier"` replaces every bound row g'z <= h by an exponential-cone epigraph
    t >= -log(alpha (h - g'z)) / alpha with unit cost on t (make_logbarrier_constraint, cone_utils.jl:173-204;
    main.jl:246-262), mu = 1/alpha.  Infeasible-start primal-dual Newton at FIXED mu with a fraction-to-boundary
    step; certificate = residuals of the perturbed KKT system (stationarity, equality, slack, t*lambda = mu)."""
    P, q, A, b, G, l, u = effective_P(qp.P), qp.q, qp.A, qp.b, qp.G, qp.l, qp.u
    n, me = P.shape[0], A.shape[0]
    ml, mh = np.isfinite(l), np.isfinite(u)
    z, y, _ = _kkt_solve(P, A, sp.csc_matrix((0, n)), -q, b, np.zeros(0))
    Gz = G @ z
    tl = np.where(ml, np.maximum(Gz - np.where(ml, l, 0.0), 1.0), 1.0)
    tu = np.where(mh, np.maximum(np.where(mh, u, 0.0) - Gz, 1.0), 1.0)
    ll, lu_ = np.where(ml, mu / tl, 0.0), np.where(mh, mu / tu, 0.0)
    Gt = G.T.tocsc()
    for it in range(max_iter):
        Gz = G @ z
        r_d = P @ z + q + A.T @ y - Gt @ ll + Gt @ lu_
        r_p = A @ z - b
        r_tl = np.where(ml, Gz - np.where(ml, l, 0.0) - tl, 0.0)
        r_tu = np.where(mh, np.where(mh, u, 0.0) - Gz - tu, 0.0)
        r_cl, r_cu = np.where(ml, tl * ll - mu, 0.0), np.where(mh, tu * lu_ - mu, 0.0)
        res = max(np.max(np.abs(r_d)) / max(1.0, np.max(np.abs(q))), np.max(np.abs(r_p), initial=0.0), np.max(np.abs(r_tl), initial=0.0),
                  np.max(np.abs(r_tu), initial=0.0), np.max(np.abs(r_cl), initial=0.0) / mu, np.max(np.abs(r_cu), initial=0.0) / mu)
        if verbose:
            print(f"oracle barrier it {it:3d} res {res:9.3e}")
        if res <= tol:
            break
        D = np.where(ml, ll / tl, 0.0) + np.where(mh, lu_ / tu, 0.0)
        el = np.where(ml, (-r_cl - ll * r_tl) / tl, 0.0)
        eu = np.where(mh, (-r_cu - lu_ * r_tu) / tu, 0.0)
        H = (P + Gt @ sp.diags(D) @ G).tocsc()
        dz, dy, _ = _kkt_solve(H, A, sp.csc_matrix((0, n)), -r_d + Gt @ el - Gt @ eu, -r_p, np.zeros(0))
        Gdz = G @ dz
        dtl, dtu = np.where(ml, Gdz + r_tl, 0.0), np.where(mh, -Gdz + r_tu, 0.0)
        dll = np.where(ml, -(ll / tl) * Gdz + el, 0.0)
        dlu = np.where(mh, (lu_ / tu) * Gdz + eu, 0.0)
        a = 1.0
        for v, dv, m in ((tl, dtl, ml), (tu, dtu, mh), (ll, dll, ml), (lu_, dlu, mh)):
            neg = m & (dv < 0)
            if neg.any():
                a = min(a, 0.995 * float(np.min(-v[neg] / dv[neg])))
        z, y = z + a * dz, y + a * dy
        tl, tu, ll, lu_ = tl + a * dtl, tu + a * dtu, ll + a * dll, lu_ + a * dlu
    else:
        raise RuntimeError("oracle barrier Newton did not converge")
    cert = dict(stationarity=float(np.max(np.abs(r_d))), equality=float(np.max(np.abs(r_p), initial=0.0)),
                bound_violation=float(max(np.max(np.abs(r_tl), initial=0.0), np.max(np.abs(r_tu), initial=0.0))),
                complementarity=float(max(np.max(np.abs(r_cl), initial=0.0), np.max(np.abs(r_cu), initial=0.0))))
    return z, dict(cert=cert, newton_iters=it)


# -------------------------------------------------------------------------------------------------
# entry points
# -------------------------------------------------------------------------------------------------
def lqp_solve_abi(xdim, udim, N, M, Nc, x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref, lx, ux, lu, uu,
                  reg_x, reg_u, slew_reg, slew_reg0, slew_um1, verbose=False, return_info=False, weights=None,
                  barrier_mu=0.0, rows=None, c_left=None):
    """Same argument list as `c_lqp_solve` (PMPC.jl/src/c_interface.jl:77-141) minus the output
    pointers; ABI-layout buffers in, X (M,N,x) / U (M,N,u) out (== the (x,N,M) / (u,N,M) the
    reference copies into X_out / U_out, c_interface.jl:138-139).  x0 is accepted and ignored, as
    in the reference (lqp_utils.jl:293-296)."""
    qp = assemble_abi(xdim, udim, N, M, Nc, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref, lx, ux, lu, uu,
                      reg_x, reg_u, slew_reg, slew_reg0, slew_um1, weights=weights)
    if rows is not None:  # ad hoc linear rows  G z <= h  over z = [U_cons; U_free; X] (the `l` part of an extra_cstrs tuple, main.jl:293-316)
        Gr, hr = sp.csc_matrix(rows[0]), _f64(rows[1]).reshape(-1)
        assert Gr.shape == (hr.size, qp.P.shape[0])
        qp.G = sp.vstack([qp.G, Gr], format="csc")
        qp.l, qp.u = np.concatenate([qp.l, np.full(hr.size, -np.inf)]), np.concatenate([qp.u, hr])
    if c_left is not None:  # linear cost on z (augment_cone_problem!'s c_left, cone_utils.jl:147-148)
        qp.q = qp.q + _f64(c_left).reshape(-1)
    if barrier_mu > 0.0 and qp.G.shape[0] > 0:
        z, info = solve_barrier_exact(qp, barrier_mu, verbose=verbose)
    else:
        z, info = solve_qp_exact(qp, verbose=verbose)
    X, U = split_vars(qp, z)
    return (X, U, info) if return_info else (X, U)


def lqp_solve_py(x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref, *, reg_x, reg_u, Nc=-1, x_l=None, x_u=None,
                 u_l=None, u_u=None, slew_reg=None, slew_reg0=None, slew_um1=None, return_info=False, weights=None,
                 barrier_mu=0.0, rows=None, c_left=None):
    """py-layout convenience wrapper (batched: x0 (M,x), fx (M,N,x,x) ...).  `weights` (M,): per-particle cost
    multipliers (cf. scale_probs_cost!, main.jl:96-112)."""
    f = _f64(f)
    M, N, xdim = f.shape
    udim = np.shape(fu)[-1]
    nanx, nanu = np.full((M, N, xdim), np.nan), np.full((M, N, udim), np.nan)
    bx = lambda z, d: d if z is None or np.size(z) == 0 else np.broadcast_to(_f64(z), d.shape).copy()
    sr = np.full(M, np.nan) if slew_reg is None else np.broadcast_to(_f64(slew_reg), (M,)).copy()
    sr0 = np.full(M, np.nan) if slew_reg0 is None else np.broadcast_to(_f64(slew_reg0), (M,)).copy()
    um1 = np.full((M, udim), np.nan) if slew_um1 is None else np.broadcast_to(_f64(slew_um1), (M, udim)).copy()
    return lqp_solve_abi(
        xdim, udim, N, M, Nc, _f64(x0), f, to_abi_mat(fx), to_abi_mat(fu), _f64(X_prev), _f64(U_prev),
        to_abi_mat(Q), to_abi_mat(R), _f64(X_ref), _f64(U_ref), bx(x_l, nanx), bx(x_u, nanx), bx(u_l, nanu),
        bx(u_u, nanu), float(reg_x), float(reg_u), sr, sr0, um1, return_info=return_info, weights=weights,
        barrier_mu=barrier_mu, rows=rows, c_left=c_left)


# -------------------------------------------------------------------------------------------------
# cone path: PMPC.jl/src/main.jl:194-354 through the C ABI (k = M, no extra_cstrs, hard boxes)
# -------------------------------------------------------------------------------------------------
COST_ANCHOR_EPS = 1e-3  # main.jl:223


def particle_costs_py(X, U, X_prev, U_prev, Q, R, X_ref, U_ref, *, reg_x, reg_u, slew_reg=None, slew_reg0=None,
                      slew_um1=None):
    """J_i = 1/2 z'P_i z + q_i'z + r_i of qp_repr_Pq (PMPC.jl/src/qp_utils.jl:60-162) — the left-hand side of the
    i-th second-order-cone epigraph (Pqr2Gh, cone_utils.jl:25-61).  py layout; the slew part of r is absent upstream
    (:140-160), so the slew0 term is 1/2 s0 |u_0|^2 - s0 u_0'u_{-1} without its constant."""
    X, U = _f64(X), _f64(U)
    M = X.shape[0]
    dx, du = X - X_ref, U - U_ref
    J = 0.5 * np.einsum("mnr,mnrt,mnt->m", dx, Q, dx) + 0.5 * np.einsum("mnr,mnrt,mnt->m", du, R, du)
    J += 0.5 * reg_x * np.sum((X - X_prev) ** 2, axis=(1, 2)) + 0.5 * reg_u * np.sum((U - U_prev) ** 2, axis=(1, 2))
    if slew_reg is not None:
        J += 0.5 * np.broadcast_to(_f64(slew_reg), (M,)) * np.sum((U[:, 1:] - U[:, :-1]) ** 2, axis=(1, 2))
    if slew_reg0 is not None and slew_um1 is not None:
        s0 = np.broadcast_to(_f64(slew_reg0), (M,))
        J += 0.5 * s0 * np.sum(U[:, 0] ** 2, -1) - s0 * np.sum(U[:, 0] * np.broadcast_to(_f64(slew_um1), U[:, 0].shape), -1)
    return J


def cone_objective(J, eps=COST_ANCHOR_EPS):
    """min over (y >= 0, t) of (1+eps) sum y_i + (1-eps) M t  s.t.  J_i <= y_i + t  (main.jl:224-238, k = M): the cost is
    piecewise linear and convex in t with breakpoints at the J_i, so the minimum sits on one of them."""
    J = np.asarray(J, dtype=np.float64)
    M = J.size
    return min((1 + eps) * np.sum(np.maximum(J - t, 0.0)) + (1 - eps) * M * t for t in J)


def lcone_solve_py(x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref, *, reg_x, reg_u, Nc=-1, return_info=False,
                   smooth_alpha=float("nan"), **kw):
    """Exact minimiser of the reference's cone-path problem for M < (1+eps)/(2 eps) ~ 500 particles, where eliminating
    (y, t) leaves  (1+eps) sum_i J_i - 2 eps M min_i J_i = sum_i w_i J_i  with w = 1+eps except for the cheapest
    particle(s), which share the deficit 2 eps M.  Search: every single candidate a (weighted exact QP, accepted iff a
    is the argmin of J at its own solution), then pairs on the kink J_a = J_b (root of the gap in the split theta).
    The returned certificate is the weighted QP's KKT certificate plus the support condition (down-weighted particles
    attain min J), i.e. the KKT conditions of the epigraph problem with multipliers lambda_i = w_i.
    `smooth_alpha` finite: the boxes enter as the log barrier of `solve_barrier_exact` (mu = 1/alpha) instead."""
    from scipy.optimize import brentq

    eps = COST_ANCHOR_EPS
    bmu = 1.0 / smooth_alpha if smooth_alpha == smooth_alpha and smooth_alpha > 0 else 0.0
    M = np.shape(f)[0]
    k = kw.pop("k", None)
    k = M if k is None or k <= 0 or k >= M else int(k)
    if 2 * eps * M >= 1 + eps or k < M:
        return _lcone_many_particles(x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref, reg_x=reg_x, reg_u=reg_u, Nc=Nc,
                                     return_info=return_info, bmu=bmu, k=k, **kw)
    args = (x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref)
    ckw = dict(reg_x=reg_x, reg_u=reg_u, slew_reg=kw.get("slew_reg"), slew_reg0=kw.get("slew_reg0"), slew_um1=kw.get("slew_um1"))

    def solve(w):
        X, U, info = lqp_solve_py(*args, reg_x=reg_x, reg_u=reg_u, Nc=Nc, weights=w, return_info=True, barrier_mu=bmu, **kw)
        return X, U, particle_costs_py(X, U, X_prev, U_prev, Q, R, X_ref, U_ref, **ckw), info

    hi = 1 + eps
    if M == 1:
        X, U, J, info = solve(np.array([1 - eps]))
        return (X, U, dict(weights=np.array([1 - eps]), J=J, qp=info)) if return_info else (X, U)
    X, U, J, _ = solve(np.full(M, hi))
    order = np.argsort(J)
    tol = lambda J: 1e-9 * max(1.0, abs(float(np.min(J))))
    for a in order[: min(M, 4)]:
        w = np.full(M, hi)
        w[a] = hi - 2 * eps * M
        X, U, J, info = solve(w)
        if J[a] <= np.min(J) + tol(J):  # the down-weighted particle attains the minimum: KKT point of the epigraph problem
            return (X, U, dict(weights=w, J=J, qp=info, kink=False)) if return_info else (X, U)
    cand = order[: min(M, 4)]
    for ia in range(len(cand)):
        for ib in range(ia + 1, len(cand)):
            a, b = cand[ia], cand[ib]

            def gap(th):
                w = np.full(M, hi)
                w[a], w[b] = hi - 2 * eps * M * th, hi - 2 * eps * M * (1 - th)
                _, _, Jt, _ = solve(w)
                return Jt[a] - Jt[b]

            if gap(0.0) < 0.0 < gap(1.0):
                th = brentq(gap, 0.0, 1.0, xtol=1e-14, rtol=1e-14)
                w = np.full(M, hi)
                w[a], w[b] = hi - 2 * eps * M * th, hi - 2 * eps * M * (1 - th)
                X, U, J, info = solve(w)
                if min(J[a], J[b]) <= np.min(J) + tol(J):
                    return (X, U, dict(weights=w, J=J, qp=info, kink=True, theta=th)) if return_info else (X, U)
    raise RuntimeError("cone oracle: no consistent threshold particle / pair found")


def _lcone_many_particles(x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref, *, reg_x, reg_u, Nc, return_info, bmu, w_floor=1e-10, k=None, **kw):
    """Cone path for M >= (1+eps)/(2 eps) ~ 500 particles (main.jl:204-238, k = M).  KKT conditions of the epigraph problem
    min (1+eps) sum y_i + (1-eps) M t  s.t.  J_i(z) <= y_i + t, y >= 0  with multipliers lambda_i of the cone rows:
    lambda_i = 1+eps where J_i > t, 0 where J_i < t, in between on J_i = t, and sum lambda_i = (1-eps) M; stationarity in z
    is that of the weighted QP sum lambda_i J_i.  So with m* = ceil(2 eps M / (1+eps)) the m*-1 cheapest particles carry NO
    weight, the m*-th carries (1+eps) m* - 2 eps M, the rest 1+eps.  A zero-weight particle's free variables do not enter
    the objective at all: the reference's minimiser is not unique in them.  STATED SEMANTICS of this restatement (and of
    the device solver, which uses the floor 1e-4): among the minimisers, the one where every zero-weight particle
    minimises its own J_i given the shared controls — the limit w -> 0+ of a floor weight, here `w_floor` = 1e-10 in an
    exact sparse solve.  Found by fixed-point iteration on the ranking, 2-cycles resolved on the kink J_a = J_b by root
    finding in the interpolation parameter; the returned certificate checks the multiplier conditions above."""
    from scipy.optimize import brentq

    eps, M = COST_ANCHOR_EPS, np.shape(f)[0]
    hi = 1 + eps
    k = M if k is None else k  # the `k` setting (main.jl:204-227): sum of the multipliers = (1 - eps) k
    n_hi = int(np.floor((1 - eps) * k / hi + 1e-12))  # particles at full weight (the costliest ones)
    mstar = max(1, M - n_hi)
    w_thr = (1 - eps) * k - hi * n_hi
    args = (x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref)
    ckw = dict(reg_x=reg_x, reg_u=reg_u, slew_reg=kw.get("slew_reg"), slew_reg0=kw.get("slew_reg0"), slew_um1=kw.get("slew_um1"))

    def solve(w):
        X, U, info = lqp_solve_py(*args, reg_x=reg_x, reg_u=reg_u, Nc=Nc, weights=w, return_info=True, barrier_mu=bmu, **kw)
        return X, U, particle_costs_py(X, U, X_prev, U_prev, Q, R, X_ref, U_ref, **ckw), info

    def weights_of(J):
        order = np.lexsort((np.arange(M), J))[:mstar]  # ascending cost, ties by index (as the device solver ranks)
        w = np.full(M, hi)
        w[order[:-1]] = w_floor
        w[order[-1]] = max(w_thr, w_floor)
        return w

    def certificate(w, J):
        t = float(np.max(J[w < hi])) if np.any(w < hi) else float(np.min(J))
        tol = 1e-9 * max(1.0, abs(t))
        lam = np.where(w <= w_floor, 0.0, w)
        return dict(sum_lambda=float(abs(np.sum(lam) - (1 - eps) * k)), full_below=float(max(0.0, np.max(t - J[w >= hi], initial=0.0)) / max(1.0, abs(t))),
                    ok=bool(abs(np.sum(lam) - (1 - eps) * k) <= 1e-9 * M and np.all(J[w >= hi] >= t - tol)))

    Ncc = np.shape(f)[1] if Nc < 0 else min(int(Nc), np.shape(f)[1])

    def polish_weightless(X, U, w):
        """The joint solve carries the weightless particles at `w_floor`: its KKT certificate (absolute tolerances) says nothing
        about THEIR accuracy.  Each of them is re-solved on its own, weight 1, with the shared controls pinned (box lo = hi)."""
        X, U = X.copy(), U.copy()
        for i in np.where(w <= w_floor)[0]:
            sl = slice(i, i + 1)
            bx = {k: (None if kw.get(k) is None else np.array(np.broadcast_to(_f64(kw[k]), U.shape if k[0] == "u" else X.shape)[sl]))
                  for k in ("x_l", "x_u", "u_l", "u_u")}
            if Ncc > 0:
                if bx["u_l"] is None:
                    bx["u_l"], bx["u_u"] = np.full(U[sl].shape, -np.inf), np.full(U[sl].shape, np.inf)
                bx["u_l"][:, :Ncc], bx["u_u"][:, :Ncc] = U[sl, :Ncc], U[sl, :Ncc]
            sub = {k: (None if kw.get(k) is None else np.broadcast_to(_f64(kw[k]), (M,) + np.shape(kw[k])[1:] if np.ndim(kw[k]) else (M,))[sl])
                   for k in ("slew_reg", "slew_reg0", "slew_um1")}
            Xi, Ui = lqp_solve_py(*[np.asarray(a_)[sl] for a_ in args], reg_x=reg_x, reg_u=reg_u, Nc=Nc, **bx, **sub)
            X[sl], U[sl, Ncc:] = Xi, Ui[:, Ncc:]
        return X, U

    X, U, J, _ = solve(np.full(M, hi))
    w1, w_prev = weights_of(J), None
    for _ in range(12):
        X, U, J, info = solve(w1)
        w2 = weights_of(J)
        if np.array_equal(w2, w1):
            cert = certificate(w1, J)
            assert cert["ok"], cert
            X, U = polish_weightless(X, U, w1)
            return (X, U, dict(weights=w1, J=J, qp=info, kink=False, cone_cert=cert)) if return_info else (X, U)
        if w_prev is not None and np.array_equal(w2, w_prev):
            d = w1 - w2
            a, b = int(np.argmin(d)), int(np.argmax(d))

            def gap(th):
                _, _, Jt, _ = solve(th * w1 + (1 - th) * w2)
                return Jt[a] - Jt[b]

            th = brentq(gap, 0.0, 1.0, xtol=1e-14, rtol=1e-14)
            w = th * w1 + (1 - th) * w2
            X, U, J, info = solve(w)
            lam = np.where(w <= w_floor, 0.0, w)
            jk = 0.5 * (J[a] + J[b])
            others = np.ones(M, bool)
            others[[a, b]] = False
            tol = 1e-9 * max(1.0, abs(jk))
            ok = abs(np.sum(lam) - (1 - eps) * k) <= 1e-9 * M and np.all(J[others & (w >= hi)] >= jk - tol) and np.all(J[others & (w < hi)] <= jk + tol)
            assert ok, "cone oracle: the kink between two rankings is not a KKT point"
            X, U = polish_weightless(X, U, w)
            return (X, U, dict(weights=w, J=J, qp=info, kink=True, theta=th)) if return_info else (X, U)
        w_prev, w1 = w1, w2
    raise RuntimeError("cone oracle: the threshold set did not settle")


def aff_solve(f, fx, fu, x0, X_prev, U_prev, Q, R, X_ref, U_ref, reg_x, reg_u, slew_rate, u_slew, x_l, x_u, u_l,
              u_u, solver_settings=None):
    """Drop-in for `pmpc.scp_mpc.aff_solve` (pmpc/scp_mpc.py:78-167) / `static_backend.aff_solve`
    (pmpc/static_backend.py:198-312) so that the REFERENCE's own SCP loop can run over the oracle
    (QP path only).  Same sentinel rules as static_backend.py:257-272."""
    solver_settings = dict(solver_settings or {})
    f, fx, fu = np.asarray(f, float), np.asarray(fx, float), np.asarray(fu, float)
    if f.ndim == 2:
        f, fx, fu = f[None], fx[None], fu[None]
    M, N, xdim = f.shape
    udim = fu.shape[-1]
    x0 = np.asarray(x0, float).reshape(M, xdim)
    rs = lambda z, d: np.asarray(z, float).reshape((M, N) + d)
    X_prev, X_ref, U_prev, U_ref = rs(X_prev, (xdim,)), rs(X_ref, (xdim,)), rs(U_prev, (udim,)), rs(U_ref, (udim,))
    Q, R = rs(Q, (xdim, xdim)), rs(R, (udim, udim))
    Nc = solver_settings.get("Nc", -1)
    none_if_empty = lambda z: None if z is None or np.size(z) == 0 else z
    X, U = lqp_solve_py(
        x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref, reg_x=reg_x, reg_u=reg_u, Nc=Nc,
        x_l=none_if_empty(x_l), x_u=none_if_empty(x_u), u_l=none_if_empty(u_l), u_u=none_if_empty(u_u),
        slew_reg=slew_rate,  # static_backend.py:262
        slew_reg0=solver_settings.get("slew_reg", None),  # static_backend.py:263-267 (sic)
        slew_um1=u_slew)
    X_traj = np.concatenate([x0[:, None, :], X], -2)  # static_backend.py:311
    return X_traj, U, dict()


# -------------------------------------------------------------------------------------------------
# CPU baseline leg (bench.py): the reference-shaped path, timed
# -------------------------------------------------------------------------------------------------
def reference_shaped_solve_abi(*abi_args, eps=1e-3):
    """What the reference does per SCP iteration, on the CPU: single-threaded CSC assembly
    (lqp_utils.jl) + a FRESH OSQP model (main.jl:145-150): KKT factorisation + ADMM to the OSQP
    default tolerance.  Returns (X, U, timing dict)."""
    t0 = time.perf_counter()
    qp = assemble_abi(*abi_args)
    P = effective_P(qp.P)
    Aa = sp.vstack([qp.A, qp.G], format="csc")
    la, ua = np.concatenate([qp.b, qp.l]), np.concatenate([qp.b, qp.u])
    t1 = time.perf_counter()
    z, _, info = osqp_admm(P, qp.q, Aa, la, ua, eps_abs=eps, eps_rel=eps)
    t2 = time.perf_counter()
    X, U = split_vars(qp, z)
    return X, U, dict(assemble_s=t1 - t0, solve_s=t2 - t1, total_s=time.perf_counter() - t0, **info)


# -------------------------------------------------------------------------------------------------
# stage-wise second-order cones on the controls (config E's thrust cones; extension beyond the reference's C ABI)
# -------------------------------------------------------------------------------------------------
def lsoc_solve_py(x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref, *, reg_x, reg_u, Nc=-1, u_l=None, u_u=None, soc_W=None,
                  soc_w0=None, soc_v=None, soc_v0=0.0, u_interior=None, mu_final=1e-13, return_info=False):
    """The joint QP of `lqp_solve_py` with, on every (particle, stage), the cone ||W u + w0||_2 <= v'u + v0 on that
    stage's controls (consensus stages: once, on the shared control) — the structured case of the reference's
    `extra_cstrs` SOC tuples (README.md:219-239, PMPC.jl/src/main.jl:293-316).  Independent of the device code in its
    linear algebra: primal log-barrier path following on the SPARSE JOINT KKT system, mu -> `mu_final`, Newton to a
    decrement below 1e-12 at every mu.  Returns X (M,N,x), U (M,N,u)."""
    f = _f64(f)
    M, N, xdim = f.shape
    udim = np.shape(fu)[-1]
    nan = np.full(1, np.nan)
    bx = lambda z: nan if z is None else np.broadcast_to(_f64(z), (M, N, udim)).copy()
    qp = assemble_abi(xdim, udim, N, M, Nc, f, to_abi_mat(fx), to_abi_mat(fu), _f64(X_prev), _f64(U_prev), to_abi_mat(Q), to_abi_mat(R),
                      _f64(X_ref), _f64(U_ref), nan, nan, bx(u_l), bx(u_u), float(reg_x), float(reg_u), nan, nan, nan)
    P, q, A, b, G, l, u = effective_P(qp.P), qp.q, qp.A, qp.b, qp.G, qp.l, qp.u
    n = P.shape[0]
    Ncc = qp.Nc
    Nf = N - Ncc
    ncu = Ncc * udim + M * Nf * udim  # control variables come first (lqp_utils.jl:12-15)
    W, w0, v = _f64(soc_W).reshape(-1, udim), _f64(soc_w0).reshape(-1), _f64(soc_v).reshape(udim)
    cones = np.arange(ncu).reshape(-1, udim)  # one cone per control block: Ncc shared blocks, then M*Nf free ones
    z = np.zeros(n)
    z[:ncu] = np.tile(_f64(u_interior).reshape(udim), ncu // udim)
    z[ncu:] = spla.spsolve(A[:, ncu:].tocsc(), b - A[:, :ncu] @ z[:ncu])
    Gt = G.T.tocsc()
    ml, mh = np.isfinite(l), np.isfinite(u)
    S = W.T @ W
    vvT = np.outer(v, v)

    def barrier(z):
        Gz = G @ z
        sl, su = np.where(ml, Gz - np.where(ml, l, 0.0), 1.0), np.where(mh, np.where(mh, u, 0.0) - Gz, 1.0)
        Uc = z[cones]                      # (ncones, udim)
        a = Uc @ v + soc_v0
        bb = Uc @ W.T + w0
        d = a * a - np.sum(bb * bb, -1)
        if np.any(sl <= 0) or np.any(su <= 0) or np.any(a <= 0) or np.any(d <= 0):
            return None
        c = a[:, None] * v[None, :] - bb @ W            # (ncones, udim): grad psi / 2
        g = Gt @ (-np.where(ml, 1.0 / sl, 0.0) + np.where(mh, 1.0 / su, 0.0))
        np.add.at(g, cones.ravel(), (-2.0 * c / d[:, None]).ravel())
        Hd = Gt @ sp.diags(np.where(ml, 1.0 / sl ** 2, 0.0) + np.where(mh, 1.0 / su ** 2, 0.0)) @ G
        blocks = 4.0 * c[:, :, None] * c[:, None, :] / (d * d)[:, None, None] - 2.0 * (vvT - S)[None] / d[:, None, None]
        rows = np.repeat(cones, udim, axis=1).ravel()
        cols = np.tile(cones, (1, udim)).ravel()
        Hc = sp.coo_matrix((blocks.ravel(), (rows, cols)), shape=(n, n)).tocsc()
        val = -np.sum(np.log(sl[ml])) - np.sum(np.log(su[mh])) - np.sum(np.log(d))
        return val, g, (Hd + Hc).tocsc()

    mu, newton = 1.0, 0
    while True:
        for _ in range(100):
            val, g, H = barrier(z)
            grad = P @ z + q + mu * g
            dz, dy, _ = _kkt_solve((P + mu * H).tocsc(), A, sp.csc_matrix((0, n)), -grad, np.zeros(A.shape[0]), np.zeros(0))
            dec = float(-grad @ dz)
            if dec <= 1e-12 * max(1.0, mu):
                break
            t = 1.0
            m0 = 0.5 * z @ (P @ z) + q @ z + mu * val
            while True:
                zt = z + t * dz
                bt = barrier(zt)
                if bt is not None and 0.5 * zt @ (P @ zt) + q @ zt + mu * bt[0] <= m0 - 1e-4 * t * dec:
                    break
                t *= 0.5
                if t < 1e-14:
                    raise RuntimeError("cone oracle: line search failed")
            z = zt
            newton += 1
        if mu <= mu_final:
            break
        mu = max(0.2 * mu, mu_final)
    X, U = split_vars(qp, z)
    return (X, U, dict(newton=newton, mu=mu)) if return_info else (X, U)


# -------------------------------------------------------------------------------------------------
# general conic rows over the joint variable vector (the reference's `extra_cstrs` tuples, PMPC.jl/src/main.jl:293-316 with the
# sign conventions of cone_solver.jl:163-177: linear rows  h - G z >= 0,  second-order cones  G z - h in SOC)
# -------------------------------------------------------------------------------------------------
def lconic_solve_py(x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref, *, reg_x, reg_u, Nc=-1, u_l=None, u_u=None, x_l=None, x_u=None,
                    lin=None, socs=(), weights=None, z0=None, mu_final=1e-13, return_info=False):
    """The joint QP of `lqp_solve_py` (optionally with per-particle cost `weights`) plus ARBITRARY conic rows over
    z = [U_cons; U_free; X] (lqp_utils.jl:12-15):  `lin = (G, h)`: rows  G z <= h;  `socs = [(G_k, h_k), ...]`: G_k z - h_k in the
    second-order cone {(t, x): |x| <= t}.  Independent of the device code: primal log-barrier path following on the sparse joint
    KKT system (boxes, rows and cones all through their barriers), Newton to a decrement below 1e-12 at every mu, mu -> mu_final.
    `z0`: a strictly feasible start for the CONTROL part (n_controls,) (the states follow from the dynamics); default: the
    unconstrained-in-the-cones box centre is NOT guessed — the caller passes one."""
    f = _f64(f)
    M, N, xdim = f.shape
    udim = np.shape(fu)[-1]
    nan = np.full(1, np.nan)
    bxu = lambda z: nan if z is None else np.broadcast_to(_f64(z), (M, N, udim)).copy()
    bxx = lambda z: nan if z is None else np.broadcast_to(_f64(z), (M, N, xdim)).copy()
    qp = assemble_abi(xdim, udim, N, M, Nc, f, to_abi_mat(fx), to_abi_mat(fu), _f64(X_prev), _f64(U_prev), to_abi_mat(Q), to_abi_mat(R),
                      _f64(X_ref), _f64(U_ref), bxx(x_l), bxx(x_u), bxu(u_l), bxu(u_u), float(reg_x), float(reg_u), nan, nan, nan,
                      weights=weights)
    P, q, A, b, G, l, u = effective_P(qp.P), qp.q, qp.A, qp.b, qp.G, qp.l, qp.u
    n = P.shape[0]
    Ncc = qp.Nc
    ncu = Ncc * udim + M * (N - Ncc) * udim
    ml, mh = np.isfinite(l), np.isfinite(u)
    # stack: box sides as linear rows  Gb z <= hb
    rows = [(-G)[ml], G[mh]]
    rhs = [-l[ml], u[mh]]
    if lin is not None:
        rows.append(sp.csr_matrix(lin[0]))
        rhs.append(_f64(lin[1]).reshape(-1))
    GL = sp.vstack(rows).tocsr() if sum(r.shape[0] for r in rows) else sp.csr_matrix((0, n))
    hL = np.concatenate(rhs) if GL.shape[0] else np.zeros(0)
    socs = [(sp.csr_matrix(Gk), _f64(hk).reshape(-1)) for Gk, hk in socs]
    z = np.zeros(n)
    assert z0 is not None, "pass a strictly feasible control vector z0 (ncontrols,)"
    z[:ncu] = _f64(z0).reshape(-1)
    z[ncu:] = spla.spsolve(A[:, ncu:].tocsc(), b - A[:, :ncu] @ z[:ncu])

    def barrier(z):
        sL = hL - GL @ z
        if np.any(sL <= 0):
            return None
        val = -np.sum(np.log(sL))
        g = GL.T @ (1.0 / sL)
        H = (GL.T @ sp.diags(1.0 / sL ** 2) @ GL).tocsc()
        for Gk, hk in socs:
            s = Gk @ z - hk
            d = s[0] * s[0] - np.sum(s[1:] * s[1:])
            if s[0] <= 0 or d <= 0:
                return None
            Js = np.concatenate([[s[0]], -s[1:]])
            val -= np.log(d)
            g = g + Gk.T @ (-2.0 * Js / d)
            J = np.diag(np.concatenate([[1.0], -np.ones(s.size - 1)]))
            W = 4.0 * np.outer(Js, Js) / (d * d) - 2.0 * J / d
            H = H + (Gk.T @ sp.csr_matrix(W) @ Gk).tocsc()
        return val, g, H.tocsc()

    if barrier(z) is None:
        raise ValueError("lconic_solve_py: z0 is not strictly feasible")
    mu, newton = 1.0, 0
    while True:
        for _ in range(200):
            val, g, H = barrier(z)
            grad = P @ z + q + mu * g
            dz, dy, _ = _kkt_solve((P + mu * H).tocsc(), A, sp.csc_matrix((0, n)), -grad, np.zeros(A.shape[0]), np.zeros(0))
            dec = float(-grad @ dz)
            if dec <= 1e-12 * max(1.0, mu):
                break
            t = 1.0
            m0 = 0.5 * z @ (P @ z) + q @ z + mu * val
            while True:
                zt = z + t * dz
                bt = barrier(zt)
                if bt is not None and 0.5 * zt @ (P @ zt) + q @ zt + mu * bt[0] <= m0 - 1e-4 * t * dec:
                    break
                t *= 0.5
                if t < 1e-14:
                    raise RuntimeError("conic oracle: line search failed")
            z = zt
            newton += 1
        if mu <= mu_final:
            break
        mu = max(0.2 * mu, mu_final)
    X, U = split_vars(qp, z)
    return (X, U, dict(newton=newton, mu=mu)) if return_info else (X, U)

