"""Host (numpy) definitions of the dynamics used by the benchmark configs, with analytic
Jacobians, in the reference's `f_fx_fu_fn(X, U) -> f, fx, fu` contract
(README.md:136-145; call site pmpc/scp_mpc.py:338-342):

    f[..., r]      = F_r(x, u)
    fx[..., r, t]  = dF_r/dx_t
    fu[..., r, t]  = dF_r/du_t

* `unicycle`  — the 4-state / 2-control car of the reference's tests/dubins_car.py:48-90
  (the reference differentiates it with torch.autograd, :11-30; the closed form here is the
  same function, checked against that autograd in tests/test_dynamics.py).
* `quadrotor` — the synthetic 12-state / 4-control rigid body of SURVEY.md §8(d) (NOT in the
  reference): x = [p(3), v(3), rpy(3), w(3)], u = [T, tx, ty, tz], explicit Euler, dt = 0.05.

These are the specifications the on-device linearisation kernels (csrc/dynamics.hip) follow.
"""
from __future__ import annotations

import numpy as np

G_ACC = 9.81


# -------------------------------------------------------------------------------------------------
# unicycle (reference tests/dubins_car.py:48-90)
# -------------------------------------------------------------------------------------------------
def unicycle(x, u, p, eps=1e-6):
    """x (...,4) = [px, py, v, th]; u (...,2) = [acc, turn]; p (...,3) = [v_scale, w_scale, T]."""
    x, u, p = np.asarray(x, float), np.asarray(u, float), np.asarray(p, float)
    vs, ws, T = p[..., 0], p[..., 1], p[..., 2]
    u1 = vs * u[..., 0]
    u2 = -ws * u[..., 1]
    u1 = u1 + np.where(u1 >= 0.0, eps, -eps)
    u2 = u2 + np.where(u2 >= 0.0, eps, -eps)
    px, py, v0, th0 = x[..., 0], x[..., 1], x[..., 2], x[..., 3]
    a = T * u2 + th0
    sa, ca, s0, c0 = np.sin(a), np.cos(a), np.sin(th0), np.cos(th0)
    iu2 = 1.0 / u2
    iu22 = iu2 * iu2
    # xp1 = px + (u2 sa v0 + T u1 u2 sa + u1 ca)/u2^2 - (s0 u2 v0 + c0 u1)/u2^2
    n1 = u2 * sa * v0 + T * u1 * u2 * sa + u1 * ca - s0 * u2 * v0 - c0 * u1
    # xp2 = py - (u2 ca v0 - u1 sa + T u1 u2 ca)/u2^2 + (c0 u2 v0 - s0 u1)/u2^2
    n2 = -(u2 * ca * v0 - u1 * sa + T * u1 * u2 * ca) + c0 * u2 * v0 - s0 * u1
    f = np.stack([px + n1 * iu22, py + n2 * iu22, v0 + T * u1, a], -1)

    shp = x.shape[:-1]
    fx = np.zeros(shp + (4, 4))
    fu = np.zeros(shp + (4, 2))
    # d/dv0, d/dth0 (da/dth0 = 1)
    dn1_dv = u2 * sa - s0 * u2
    dn1_dth = u2 * ca * v0 + T * u1 * u2 * ca - u1 * sa - c0 * u2 * v0 + s0 * u1
    dn2_dv = -u2 * ca + c0 * u2
    dn2_dth = -(-u2 * sa * v0 - u1 * ca - T * u1 * u2 * sa) - s0 * u2 * v0 - c0 * u1
    fx[..., 0, 0] = 1.0
    fx[..., 0, 2] = dn1_dv * iu22
    fx[..., 0, 3] = dn1_dth * iu22
    fx[..., 1, 1] = 1.0
    fx[..., 1, 2] = dn2_dv * iu22
    fx[..., 1, 3] = dn2_dth * iu22
    fx[..., 2, 2] = 1.0
    fx[..., 3, 3] = 1.0
    # d/du1, d/du2 (da/du2 = T), then chain to the raw controls: du1/dU0 = vs, du2/dU1 = -ws
    dn1_du1 = T * u2 * sa + ca - c0
    dn2_du1 = sa - T * u2 * ca - s0
    dn1_du2 = sa * v0 + u2 * ca * T * v0 + T * u1 * sa + T * u1 * u2 * ca * T - u1 * sa * T - s0 * v0
    dn2_du2 = -(ca * v0 - u2 * sa * T * v0 - u1 * ca * T + T * u1 * ca - T * T * u1 * u2 * sa) + c0 * v0
    d1_du2 = dn1_du2 * iu22 - 2.0 * n1 * iu22 * iu2
    d2_du2 = dn2_du2 * iu22 - 2.0 * n2 * iu22 * iu2
    fu[..., 0, 0] = dn1_du1 * iu22 * vs
    fu[..., 1, 0] = dn2_du1 * iu22 * vs
    fu[..., 2, 0] = T * vs
    fu[..., 0, 1] = d1_du2 * (-ws)
    fu[..., 1, 1] = d2_du2 * (-ws)
    fu[..., 3, 1] = T * (-ws)
    return f, fx, fu


def unicycle_torch(x, u, p, eps=1e-6):
    """`unicycle` on torch tensors of any device (same closed form, same outputs) — an example of an `f_fx_fu_fn` for
    the device-resident loop `solve(..., device="cuda")`."""
    import torch

    vs, ws, T = p[..., 0], p[..., 1], p[..., 2]
    u1, u2 = vs * u[..., 0], -ws * u[..., 1]
    u1 = u1 + torch.where(u1 >= 0.0, eps, -eps)
    u2 = u2 + torch.where(u2 >= 0.0, eps, -eps)
    px, py, v0, th0 = x[..., 0], x[..., 1], x[..., 2], x[..., 3]
    a = T * u2 + th0
    sa, ca, s0, c0 = torch.sin(a), torch.cos(a), torch.sin(th0), torch.cos(th0)
    iu2 = 1.0 / u2
    iu22 = iu2 * iu2
    n1 = u2 * sa * v0 + T * u1 * u2 * sa + u1 * ca - s0 * u2 * v0 - c0 * u1
    n2 = -(u2 * ca * v0 - u1 * sa + T * u1 * u2 * ca) + c0 * u2 * v0 - s0 * u1
    f = torch.stack([px + n1 * iu22, py + n2 * iu22, v0 + T * u1, a], -1)
    fx = torch.zeros(x.shape[:-1] + (4, 4), dtype=x.dtype, device=x.device)
    fu = torch.zeros(x.shape[:-1] + (4, 2), dtype=x.dtype, device=x.device)
    fx[..., 0, 0] = fx[..., 1, 1] = fx[..., 2, 2] = fx[..., 3, 3] = 1.0
    fx[..., 0, 2] = (u2 * sa - s0 * u2) * iu22
    fx[..., 0, 3] = (u2 * ca * v0 + T * u1 * u2 * ca - u1 * sa - c0 * u2 * v0 + s0 * u1) * iu22
    fx[..., 1, 2] = (-u2 * ca + c0 * u2) * iu22
    fx[..., 1, 3] = (-(-u2 * sa * v0 - u1 * ca - T * u1 * u2 * sa) - s0 * u2 * v0 - c0 * u1) * iu22
    dn1_du1, dn2_du1 = T * u2 * sa + ca - c0, sa - T * u2 * ca - s0
    dn1_du2 = sa * v0 + u2 * ca * T * v0 + T * u1 * sa + T * u1 * u2 * ca * T - u1 * sa * T - s0 * v0
    dn2_du2 = -(ca * v0 - u2 * sa * T * v0 - u1 * ca * T + T * u1 * ca - T * T * u1 * u2 * sa) + c0 * v0
    fu[..., 0, 0], fu[..., 1, 0], fu[..., 2, 0] = dn1_du1 * iu22 * vs, dn2_du1 * iu22 * vs, T * vs
    fu[..., 0, 1] = (dn1_du2 * iu22 - 2.0 * n1 * iu22 * iu2) * (-ws)
    fu[..., 1, 1] = (dn2_du2 * iu22 - 2.0 * n2 * iu22 * iu2) * (-ws)
    fu[..., 3, 1] = T * (-ws)
    return f, fx, fu


# -------------------------------------------------------------------------------------------------
# synthetic quadrotor
# -------------------------------------------------------------------------------------------------
QUAD_DT = 0.05


def quadrotor(x, u, p, dt=QUAD_DT):
    """x (...,12), u (...,4), p (...,4) = [mass, Jx, Jy, Jz]. Explicit Euler step of a rigid body with
    ZYX Euler angles rpy = (phi, theta, psi)."""
    x, u, p = np.asarray(x, float), np.asarray(u, float), np.asarray(p, float)
    m, Jx, Jy, Jz = (p[..., i] for i in range(4))
    v = x[..., 3:6]
    ph, th, ps = x[..., 6], x[..., 7], x[..., 8]
    wx, wy, wz = x[..., 9], x[..., 10], x[..., 11]
    T, tx, ty, tz = (u[..., i] for i in range(4))
    sph, cph, sth, cth, sps, cps = np.sin(ph), np.cos(ph), np.sin(th), np.cos(th), np.sin(ps), np.cos(ps)
    tth, icth = sth / cth, 1.0 / cth
    # thrust direction b = R e3
    bx = cps * sth * cph + sps * sph
    by = sps * sth * cph - cps * sph
    bz = cth * cph
    a = T / m
    shp = x.shape[:-1]
    xd = np.zeros(shp + (12,))
    xd[..., 0:3] = v
    xd[..., 3], xd[..., 4], xd[..., 5] = a * bx, a * by, a * bz - G_ACC
    xd[..., 6] = wx + sph * tth * wy + cph * tth * wz
    xd[..., 7] = cph * wy - sph * wz
    xd[..., 8] = sph * icth * wy + cph * icth * wz
    xd[..., 9] = (tx - (Jz - Jy) * wy * wz) / Jx
    xd[..., 10] = (ty - (Jx - Jz) * wz * wx) / Jy
    xd[..., 11] = (tz - (Jy - Jx) * wx * wy) / Jz
    f = x + dt * xd

    A = np.zeros(shp + (12, 12))
    B = np.zeros(shp + (12, 4))
    for i in range(3):
        A[..., i, 3 + i] = 1.0
    # d b / d(ph, th, ps)
    dbx = (-cps * sth * sph + sps * cph, cps * cth * cph, -sps * sth * cph + cps * sph)
    dby = (-sps * sth * sph - cps * cph, sps * cth * cph, cps * sth * cph + sps * sph)
    dbz = (-cth * sph, -sth * cph, np.zeros_like(ph))
    for k in range(3):
        A[..., 3, 6 + k] = a * dbx[k]
        A[..., 4, 6 + k] = a * dby[k]
        A[..., 5, 6 + k] = a * dbz[k]
    sec2 = icth * icth
    # rpy rates
    A[..., 6, 6] = cph * tth * wy - sph * tth * wz
    A[..., 6, 7] = sph * sec2 * wy + cph * sec2 * wz
    A[..., 6, 9], A[..., 6, 10], A[..., 6, 11] = 1.0, sph * tth, cph * tth
    A[..., 7, 6] = -sph * wy - cph * wz
    A[..., 7, 10], A[..., 7, 11] = cph, -sph
    A[..., 8, 6] = cph * icth * wy - sph * icth * wz
    A[..., 8, 7] = (sph * wy + cph * wz) * sth * sec2
    A[..., 8, 10], A[..., 8, 11] = sph * icth, cph * icth
    # body rates
    A[..., 9, 10], A[..., 9, 11] = -(Jz - Jy) * wz / Jx, -(Jz - Jy) * wy / Jx
    A[..., 10, 9], A[..., 10, 11] = -(Jx - Jz) * wz / Jy, -(Jx - Jz) * wx / Jy
    A[..., 11, 9], A[..., 11, 10] = -(Jy - Jx) * wy / Jz, -(Jy - Jx) * wx / Jz
    B[..., 3, 0], B[..., 4, 0], B[..., 5, 0] = bx / m, by / m, bz / m
    B[..., 9, 1], B[..., 10, 2], B[..., 11, 3] = 1.0 / Jx, 1.0 / Jy, 1.0 / Jz
    fx = np.eye(12) + dt * A
    fu = dt * B
    return f, fx, fu


# -------------------------------------------------------------------------------------------------
# benchmark problem generators (BASELINE.md §2 / SURVEY.md §8d), seeded
# -------------------------------------------------------------------------------------------------
def make_unicycle_problem(M=256, N=30, seed=2020, Nc=1):
    """Config A (M=1: p=[1,1,0.3], x0=1) / config B (per-particle p and x0 jitter)."""
    rng = np.random.default_rng(seed)
    xdim, udim = 4, 2
    if M == 1:
        p = np.array([[1.0, 1.0, 0.3]])
        x0 = np.ones((1, xdim))
    else:
        p = np.stack([1 + 0.1 * rng.standard_normal(M), 1 + 0.1 * rng.standard_normal(M), np.full(M, 0.3)], -1)
        x0 = 1.0 + 0.05 * rng.standard_normal((M, xdim))
    Q = np.tile(np.eye(xdim), (M, N, 1, 1))
    R = np.tile(1e-2 * np.eye(udim), (M, N, 1, 1))
    prob = dict(
        x0=x0, Q=Q, R=R, X_ref=np.zeros((M, N, xdim)), U_ref=np.zeros((M, N, udim)),
        X_prev=np.zeros((M, N, xdim)), U_prev=np.zeros((M, N, udim)),
        u_l=-np.ones((M, N, udim)), u_u=np.ones((M, N, udim)), reg_x=1.0, reg_u=1.0,
        solver_settings=dict(solver="osqp", Nc=Nc),
    )
    pp = p[:, None, :]

    def f_fx_fu_fn(X, U):
        return unicycle(X, U, pp)

    prob["f_fx_fu_fn"] = f_fx_fu_fn
    prob["params"] = p
    return prob


def make_quadrotor_problem(M=1024, N=50, seed=2020, Nc=1):
    """Configs C/D: synthetic quadrotor, box constraints on the controls."""
    rng = np.random.default_rng(seed)
    xdim, udim = 12, 4
    mass = 1.0 * (1 + 0.1 * rng.standard_normal(M))
    J = np.array([0.01, 0.01, 0.02])[None, :] * (1 + 0.1 * rng.standard_normal((M, 1)))
    p = np.concatenate([mass[:, None], J], -1)
    x0 = np.zeros((M, xdim))
    x0[:, :3] = rng.uniform(-2, 2, (M, 3))
    Qd = np.concatenate([10 * np.ones(3), np.ones(3), np.ones(3), 0.1 * np.ones(3)])
    Q = np.tile(np.diag(Qd), (M, N, 1, 1))
    R = np.tile(0.1 * np.eye(udim), (M, N, 1, 1))
    U_ref = np.zeros((M, N, udim))
    U_ref[..., 0] = (mass * G_ACC)[:, None]
    u_l = np.zeros((M, N, udim))
    u_u = np.zeros((M, N, udim))
    u_l[..., 0], u_u[..., 0] = 0.0, (2 * mass * G_ACC)[:, None]
    u_l[..., 1:], u_u[..., 1:] = -0.5, 0.5
    X_prev = np.tile(x0[:, None, :], (1, N, 1))
    prob = dict(
        x0=x0, Q=Q, R=R, X_ref=np.zeros((M, N, xdim)), U_ref=U_ref, X_prev=X_prev, U_prev=U_ref.copy(),
        u_l=u_l, u_u=u_u, reg_x=1.0, reg_u=1e-1, solver_settings=dict(solver="osqp", Nc=Nc),
    )
    pp = p[:, None, :]

    def f_fx_fu_fn(X, U):
        return quadrotor(X, U, pp)

    prob["f_fx_fu_fn"] = f_fx_fu_fn
    prob["params"] = p
    return prob
