"""Prototype (numpy, CPU) of active-set / semismooth-Newton rounds for stage-wise second-order cones next to control boxes
— the method planned for the device (kernels_as.hip with cone terms), validated here against the sparse cone oracle before
any HIP is written.  Test / development aid only.

Per (particle, stage): controls u, box lo <= u <= hi, cone  s = A u + c in K = {(s0, sb): |sb| <= s0},  A = [v'; W], c = (v0, w0).
KKT: grad J(u) = A'z + box multipliers, z in K, s in K, s'z = 0  <=>  Phi = s - Proj_K(s - z) = 0 (natural map).
A round = ONE exact structured Newton solve of the QP with, per cone, the generalised Jacobian of the projection at
w = s - z (three cases: interior -> cone off, polar -> held at the apex, else -> one equality along e- = (1, -wh)/sqrt2 and
a finite curvature term (1-theta)/theta along the tangential directions), boxes by the primal-dual active-set rule."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from tests.support.structured_np import Problem, StructuredLQ, _bounds  # noqa: E402


def proj_soc(w):
    """Projection of w (..., q+1) onto the second-order cone, and the case: 0 interior, 1 boundary, 2 polar (-> 0)."""
    w0, wb = w[..., 0], w[..., 1:]
    nb = np.linalg.norm(wb, axis=-1)
    case = np.where(nb <= w0, 0, np.where(nb <= -w0, 2, 1))
    lam = 0.5 * (w0 + nb)
    wh = wb / np.maximum(nb, 1e-300)[..., None]
    pb = np.concatenate([lam[..., None], lam[..., None] * wh], -1)
    out = np.where((case == 0)[..., None], w, np.where((case == 2)[..., None], 0.0, pb))
    return out, case, wh, nb


class ConeAS:
    def __init__(self, p: Problem, W, w0, v, v0, rho_scale=1e7, verbose=False):
        self.p, self.verbose = p, verbose
        self.A = np.vstack([np.asarray(v, float)[None, :], np.atleast_2d(np.asarray(W, float))])  # (q+1, u)
        self.c = np.concatenate([[float(v0)], np.asarray(w0, float).reshape(-1)])
        self.q1 = self.A.shape[0]
        M, N, u, Nc = p.M, p.N, p.u, p.Nc
        _, _, lu, uu = _bounds(p)
        if Nc > 0:
            lu[:, :Nc], uu[:, :Nc] = lu[0:1, :Nc], uu[0:1, :Nc]
        # a box side implied by the cone's s0 >= 0 (thrust >= 0) is dropped: at the apex the cone holds that control
        v_ = self.A[0]
        nz = np.flatnonzero(v_)
        if nz.size == 1 and v_[nz[0]] > 0:
            k = nz[0]
            red = lu[..., k] <= -self.c[0] / v_[k] + 1e-15
            lu[..., k] = np.where(red, -np.inf, lu[..., k])
        self.lu, self.uu = lu, uu
        # apex hold needs an axis-aligned A (one nonzero per row, distinct columns)
        self.apex_cols = []
        for r in range(self.q1):
            nzr = np.flatnonzero(self.A[r])
            assert nzr.size == 1, "prototype: axis-aligned cone rows only"
            self.apex_cols.append(nzr[0])
        self.apex_cols = np.array(self.apex_cols)
        self.apex_val = -self.c / self.A[np.arange(self.q1), self.apex_cols]
        self.lq = StructuredLQ(p)
        self.rho_scale = rho_scale
        self.big = 1e30

    def s_of(self, U):
        return U @ self.A.T + self.c

    def round(self, Ub, z, act, apex):
        """One round from base controls Ub (M,N,u), cone multipliers z (M,N,q+1), box statuses act (M,N,u), apex flags (M,N).
        Returns new (U, z, act, apex), counters."""
        p, A, big = self.p, self.A, self.big
        M, N, u, Nc = p.M, p.N, p.u, p.Nc
        Xb = p.rollout(Ub)
        s = self.s_of(Ub)
        w = s - z
        _, case, wh, nb = proj_soc(w)
        # ---- Newton terms of every cone ---------------------------------------------------------------------------------
        Hadd = np.zeros((M, N, u, u))
        gadd = np.zeros((M, N, u))
        bnd = case == 1
        kappa = np.where(bnd, w[..., 0] / np.maximum(nb, 1e-300), 0.0)
        theta = 0.5 * (1.0 + kappa)
        curv = np.where(bnd, (1.0 - theta) / np.maximum(theta, 1e-300), 0.0)
        # e- = (1, -wh)/sqrt2;  tangential projector on the sb part: I - wh wh'
        em = np.concatenate([np.ones(wh.shape[:-1] + (1,)), -wh], -1) / np.sqrt(2.0)
        a_m = em @ A                                       # (M,N,u)  normal of the equality e-' s+ = 0
        Ab = A[1:]                                         # (q,u)
        Pt = np.eye(self.q1 - 1) - wh[..., :, None] * wh[..., None, :]      # (M,N,q,q)
        AtP = np.einsum("qa,mnqr,rb->mnab", Ab, Pt, Ab)    # Ab' Pt Ab
        sb = s[..., 1:]
        gt = np.einsum("qa,mnqr,mnr->mna", Ab, Pt, sb)     # Ab' Pt sb
        Rscale = np.trace(p.Rt, axis1=-2, axis2=-1) / u
        rho = self.rho_scale * Rscale
        nu_hat = np.einsum("mnq,mnq->mn", em, z)           # e-' z: multiplier estimate of the equality
        es = np.einsum("mnq,mnq->mn", em, s)
        Hadd += np.where(bnd, curv, 0.0)[..., None, None] * AtP + np.where(bnd, rho, 0.0)[..., None, None] * a_m[..., :, None] * a_m[..., None, :]
        gadd += np.where(bnd, curv, 0.0)[..., None] * gt + (np.where(bnd, -nu_hat + rho * es, 0.0))[..., None] * a_m
        # apex (polar case): A du = -s_b by penalty + multiplier estimate: rho/2 |s_b + A du|^2 - zhat'(A du)
        ap = case == 2
        AtA = A.T @ A
        Hadd += np.where(ap, rho, 0.0)[..., None, None] * AtA
        gadd += np.where(ap[..., None], (-z + rho[..., None] * s) @ A, 0.0)
        held_apex = np.zeros((M, N, u), bool)
        Dbox = np.where(act > 0, big, 0.0)
        Hfull = Hadd + Dbox[..., None] * np.eye(u)
        # ---- structured Newton solve -------------------------------------------------------------------------------------
        gx, gu, gc0 = p.gradient(Xb, Ub)
        gu = gu + gadd * (1.0 if Nc == 0 else np.concatenate([np.zeros((1, Nc, 1)), np.ones((1, N - Nc, 1))], 1))
        gce, Dc = None, None
        if Nc > 0:
            gce = np.zeros(Nc * u)
            gce[:u] += gc0.sum(0)
            gce += gadd[0, :Nc].reshape(-1)
            Dc = np.zeros((Nc * u, Nc * u))
            for j in range(Nc):
                Dc[j * u:(j + 1) * u, j * u:(j + 1) * u] = Hfull[0, j]
        self.lq.factor(None, Hfull, Dc)
        dX, dU = self.lq.solve(gx, gu, gce)
        # ---- box rule ------------------------------------------------------------------------------------------------------
        lu, uu = self.lu, self.uu
        tol_l, tol_p = 1e-9, 1e-13
        lam = np.where(act == 1, -big * dU, big * dU)
        zt = Ub + dU
        rel = (act > 0) & (lam < -tol_l)
        add_l = (act == 0) & ~held_apex & (zt < lu - tol_p * np.maximum(1.0, np.abs(lu)))
        add_u = (act == 0) & ~held_apex & (zt > uu + tol_p * np.maximum(1.0, np.abs(uu)))
        act_new = np.where(rel, 0, np.where(add_l, 1, np.where(add_u, 2, act)))
        U_new = np.where(act > 0, Ub, np.where(add_l, lu, np.where(add_u, uu, zt)))
        U_new = np.where(held_apex, Ub, U_new)
        # ---- cone rule: new multipliers, new cases ------------------------------------------------------------------------
        s_new = self.s_of(U_new)
        s_raw = self.s_of(Ub + dU)   # the stage's own Newton step (before any clamping): what the multiplier updates are valid for
        z_new = np.zeros_like(z)
        # boundary: z+ = nu+ e-  - curv * Pt sb+   (tangential part), nu+ = nu_hat - rho e-' s+
        nu_new = nu_hat - rho * np.einsum("mnq,mnq->mn", em, s_raw)
        zb = nu_new[..., None] * em
        zb[..., 1:] -= curv[..., None] * np.einsum("mnqr,mnr->mnq", Pt, s_raw[..., 1:])
        z_new = np.where(bnd[..., None], zb, z_new)
        # apex: z+ = zhat - rho s+
        z_new = np.where(ap[..., None], z - rho[..., None] * s_raw, z_new)
        w_new = s_new - z_new
        _, case_new, _, _ = proj_soc(w_new)
        apex_new = np.zeros_like(apex)
        z_new = np.where((case_new == 0)[..., None], 0.0, z_new)
        if Nc > 0:  # shared controls: one decision (particle 0's)
            U_new[:, :Nc], z_new[:, :Nc], act_new[:, :Nc], apex_new[:, :Nc] = U_new[0:1, :Nc], z_new[0:1, :Nc], act_new[0:1, :Nc], apex_new[0:1, :Nc]
        phi = s_new - proj_soc(s_new - z_new)[0]
        n_box = int(rel.sum() + add_l.sum() + add_u.sum())
        n_case = int((case_new != case).sum())
        res = float(np.abs(phi).max())
        step = float(np.abs(dU[~((act > 0) | held_apex)]).max()) if (~((act > 0) | held_apex)).any() else 0.0
        return U_new, z_new, act_new, apex_new, dict(box=n_box, case=n_case, phi=res, step=step, nb=int((case_new == 1).sum()), nap=int(apex_new.sum()))

    def solve(self, U0, z0, act0, apex0, max_rounds=30, tol=1e-10):
        U, z, act, apex = U0.copy(), z0.copy(), act0.copy(), apex0.copy()
        for r in range(max_rounds):
            U, z, act, apex, info = self.round(U, z, act, apex)
            if self.verbose:
                print(f"   round {r + 1:2d}: box changes {info['box']:5d} cone case changes {info['case']:5d} |Phi| {info['phi']:.2e} step {info['step']:.2e} "
                      f"boundary {info['nb']} apex {info['nap']}")
            if info["box"] == 0 and info["case"] == 0 and info["phi"] <= tol and info["step"] <= tol:
                return U, z, act, apex, r + 1, True
        return U, z, act, apex, max_rounds, False


def initial_from_solution(cas: ConeAS, U):
    """Statuses and multipliers from a (near-)optimal solution U of a neighbouring problem: boxes by position, cone multipliers
    from the stationarity residual — here simply z = 0 / apex flags from the position (the first round's Newton step sorts it out)."""
    lu, uu = cas.lu, cas.uu
    act = np.where(U <= lu, 1, np.where(U >= uu, 2, 0))
    s = cas.s_of(U)
    apex = np.abs(s).max(-1) <= 1e-12
    z = np.zeros(s.shape)
    return act, apex, z


def main():
    import argparse

    from oracle import lqp_oracle as orc
    from pmpc_amd import dynamics as dyn

    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=16)
    ap.add_argument("--N", type=int, default=100)
    ap.add_argument("--Nc", type=int, default=1)
    ap.add_argument("--scp", type=int, default=6)
    ap.add_argument("--oracle", action="store_true")
    ap.add_argument("--thrust-min", type=float, default=0.0)
    ap.add_argument("--seed", type=int, default=2020)
    a = ap.parse_args()
    prob = dyn.make_quadrotor_problem(M=a.M, N=a.N, Nc=a.Nc, seed=a.seed)
    if a.thrust_min > 0:
        prob["u_l"][..., 0] = a.thrust_min * 0.5 * prob["u_u"][..., 0]
    W = np.zeros((2, 4)); W[0, 1] = W[1, 2] = 1.0
    v, v0, w0 = np.array([0.3, 0, 0, 0]), 0.0, np.zeros(2)
    Xp, Up = prob["X_prev"].copy(), prob["U_prev"].copy()
    state = None
    for it in range(a.scp):
        X_ = np.concatenate([prob["x0"][:, None, :], Xp[:, :-1]], 1)
        f, fx, fu = prob["f_fx_fu_fn"](X_, Up)
        p = Problem(f, fx, fu, Xp, Up, prob["Q"], prob["R"], prob["X_ref"], prob["U_ref"], prob["reg_x"], prob["reg_u"], Nc=a.Nc,
                    u_l=prob["u_l"], u_u=prob["u_u"])
        cas = ConeAS(p, W, w0, v, v0, verbose=True)
        t0 = time.time()
        if state is None:  # cold: start at hover (strictly feasible), everything free
            U0 = np.tile(np.array([9.81, 0, 0, 0.0]), (a.M, a.N, 1))
            act0, apex0, z0 = initial_from_solution(cas, U0)
        else:
            U0, z0, act0, apex0 = state
        U, z, act, apex, rounds, ok = cas.solve(U0, z0, act0, apex0)
        X = p.rollout(U)
        print(f"SCP it {it + 1}: rounds {rounds} ok {ok}  ({time.time() - t0:.1f}s)  cone viol {np.max(np.linalg.norm(U[..., 1:3], axis=-1) - 0.3 * U[..., 0]):.2e}")
        if a.oracle:
            Xo, Uo = orc.lsoc_solve_py(prob["x0"], f, fx, fu, Xp, Up, prob["Q"], prob["R"], prob["X_ref"], prob["U_ref"], reg_x=prob["reg_x"],
                                       reg_u=prob["reg_u"], Nc=a.Nc, u_l=prob["u_l"], u_u=prob["u_u"], soc_W=W, soc_w0=w0, soc_v=v, soc_v0=v0,
                                       u_interior=np.array([9.81, 0, 0, 0.0]))
            print(f"      vs oracle: X {np.linalg.norm(X - Xo) / np.linalg.norm(Xo):.2e}  U {np.linalg.norm(U - Uo) / np.linalg.norm(Uo):.2e}")
        if not ok:
            break
        state = (U, z, act, apex)
        Xp, Up = X, U


if __name__ == "__main__":
    main()
