"""numpy model of the DEVICE algorithm (test/dev aid only — never imported by pmpc_amd).

It mirrors, step for step, what pmpc_amd/csrc does on the GPU so that the algorithm can be
validated against the oracle on the CPU and so that the world_size-2 gloo test can exercise
the particle-sharded consensus reduction:

  * `StructuredLQ.factor / solve`  — per-particle Riccati recursion over the free stages,
    forward-sensitivity condensing of the Nc consensus stages, SUM over particles of the
    reduced (H_i, g_i)  [the only cross-particle / cross-GPU step], dense solve, forward sweep.
    Slew penalties are handled by augmenting the stage state with the previous control.
  * `ipm_solve` — Mehrotra predictor-corrector on the box constraints whose Newton systems
    are the structured solves above (same factorisation for predictor and corrector).

Everything is batched over particles in "py" layout (see oracle/lqp_oracle.py docstring).
"""
from __future__ import annotations

import numpy as np


def sym_triu(A):
    """OSQP keeps triu(P): effective symmetric block = triu(A) + triu(A,1)'."""
    U = np.triu(A)
    return U + np.swapaxes(np.triu(A, 1), -1, -2)


class Problem:
    """Holds the ABI inputs (py layout) after sentinel processing."""

    def __init__(self, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref, reg_x, reg_u, Nc=-1, x_l=None, x_u=None,
                 u_l=None, u_u=None, slew_reg=None, slew_reg0=None, slew_um1=None):
        self.f, self.fx, self.fu = (np.asarray(a, float) for a in (f, fx, fu))
        self.M, self.N, self.x = self.f.shape
        self.u = self.fu.shape[-1]
        M, N, x, u = self.M, self.N, self.x, self.u
        self.X_prev, self.U_prev = np.asarray(X_prev, float), np.asarray(U_prev, float)
        self.X_ref, self.U_ref = np.asarray(X_ref, float), np.asarray(U_ref, float)
        self.Q, self.R = np.asarray(Q, float), np.asarray(R, float)
        self.reg_x, self.reg_u = float(reg_x), float(reg_u)
        self.Nc = N if Nc < 0 else int(Nc)
        nan = lambda a: a is None or np.size(a) == 0 or np.any(np.isnan(a))
        self.has_xb = not (nan(x_l) or nan(x_u))
        self.has_ub = not (nan(u_l) or nan(u_u))
        self.x_l = np.broadcast_to(np.asarray(x_l, float), (M, N, x)) if self.has_xb else None
        self.x_u = np.broadcast_to(np.asarray(x_u, float), (M, N, x)) if self.has_xb else None
        self.u_l = np.broadcast_to(np.asarray(u_l, float), (M, N, u)) if self.has_ub else None
        self.u_u = np.broadcast_to(np.asarray(u_u, float), (M, N, u)) if self.has_ub else None
        self.s = np.zeros(M) if nan(slew_reg) else np.broadcast_to(np.asarray(slew_reg, float), (M,)).copy()
        if nan(slew_reg0) or nan(slew_um1):
            self.s0, self.um1 = np.zeros(M), np.zeros((M, u))
        else:
            self.s0 = np.broadcast_to(np.asarray(slew_reg0, float), (M,)).copy()
            self.um1 = np.broadcast_to(np.asarray(slew_um1, float), (M, u)).copy()
        self.slew = bool(np.any(self.s != 0.0))
        # effective symmetric Hessian blocks (without IPM diagonals)
        self.Qt = sym_triu(self.Q) + self.reg_x * np.eye(x)
        d = np.empty((M, N))  # slew diagonal, lqp_utils.jl:31-39
        for j in range(N):
            d[:, j] = (self.s0 + self.s) if j == 0 else (self.s if j == N - 1 else 2 * self.s)
        self.Rt = sym_triu(self.R) + (self.reg_u + d)[:, :, None, None] * np.eye(u)
        self.qx = -self.reg_x * self.X_prev - np.einsum("mjrt,mjt->mjr", self.Q, self.X_ref)
        self.qu = -self.reg_u * self.U_prev - np.einsum("mjrt,mjt->mjr", self.R, self.U_ref)

    # ---- operators of the joint objective ---------------------------------------------------
    def rollout(self, U):
        X = np.empty((self.M, self.N, self.x))
        for j in range(self.N):
            X[:, j] = self.f[:, j] + np.einsum("mrt,mt->mr", self.fu[:, j], U[:, j] - self.U_prev[:, j])
            if j > 0:
                X[:, j] += np.einsum("mrt,mt->mr", self.fx[:, j], X[:, j - 1] - self.X_prev[:, j - 1])
        return X

    def gradient(self, X, U):
        """Per-particle gradient of the smooth objective at (X,U): gx (M,N,x), gu (M,N,u).
        For consensus stages the true gradient is the SUM over particles of gu[:, j] plus the
        slew_reg0 linear term (returned separately as gc0)."""
        gx = np.einsum("mjrt,mjt->mjr", self.Qt, X) + self.qx
        gu = np.einsum("mjrt,mjt->mjr", self.Rt, U) + self.qu
        s = self.s[:, None, None]
        gu[:, 1:] -= s * U[:, :-1]
        gu[:, :-1] -= s * U[:, 1:]
        # lqp_utils.jl:165: q[1:udim] += sum_i -slew_reg0_i*um1_i; lost when Nc == 0 (:190 overwrites)
        gc0 = -(self.s0[:, None] * self.um1) if self.Nc >= 1 else np.zeros((self.M, self.u))
        return gx, gu, gc0


class StructuredLQ:
    """Equality-constrained Newton system  min 1/2 dz'(P+D)dz + g'dz  s.t. A dz = 0."""

    def __init__(self, p: Problem, allreduce=None):
        self.p = p
        self.allreduce = allreduce or (lambda a: a)  # sums (H,g) over shards

    def factor(self, Dx=None, Du=None, Dc=None):
        """Dx (M,N,x), Du (M,N,u) extra diagonals — or (M,N,u,u) full symmetric blocks — (Du ignored on consensus stages),
        Dc (Nc*u,) diagonal or (Nc*u, Nc*u) full block of the summed consensus system."""
        p = self.p
        M, N, x, u, Nc = p.M, p.N, p.x, p.u, p.Nc
        w = u if p.slew else 0
        n = x + w
        Dx = np.zeros((M, N, x)) if Dx is None else Dx
        Du = np.zeros((M, N, u)) if Du is None else Du
        self.n, self.w = n, w
        # stage matrices F_j = [A~ | B~]  (n x (n+u))
        F = np.zeros((M, N, n, n + u))
        F[:, 1:, :x, :x] = p.fx[:, 1:]
        F[:, :, :x, n:] = p.fu
        if w:
            F[:, :, x:, n:] = np.eye(u)
        self.F = F
        Qd = p.Qt + Dx[..., None] * np.eye(x)
        S = np.zeros((M, n, n))
        S[:, :x, :x] = Qd[:, N - 1]
        self.K = np.zeros((M, N, u, n))
        self.Huu_inv = np.zeros((M, N, u, u))
        for j in range(N - 1, Nc - 1, -1):
            H = np.einsum("mab,mac,mcd->mbd", F[:, j], S, F[:, j])
            H[:, n:, n:] += p.Rt[:, j] + (Du[:, j] if Du.ndim == 4 else Du[:, j][..., None] * np.eye(u))  # (ndim 4: full u x u blocks)
            if w and j > 0:
                H[:, n:, x:n] -= p.s[:, None, None] * np.eye(u)
                H[:, x:n, n:] -= p.s[:, None, None] * np.eye(u)
            Huu_inv = np.linalg.inv(H[:, n:, n:])
            K = Huu_inv @ H[:, n:, :n]
            self.K[:, j], self.Huu_inv[:, j] = K, Huu_inv
            S = H[:, :n, :n] - np.swapaxes(H[:, n:, :n], -1, -2) @ K
            S = 0.5 * (S + np.swapaxes(S, -1, -2))
            if j > 0:
                S[:, :x, :x] += Qd[:, j - 1]
        self.S_c = S  # S_{Nc-1} (value incl. stage Nc-1 state cost); unused if Nc == 0
        if Nc > 0:
            # forward sensitivities Phi_j (n x Nc*u) and the condensed per-particle Hessian
            nc = Nc * u
            Phi = np.zeros((M, Nc, n, nc))
            Hc = np.zeros((M, nc, nc))
            prev = np.zeros((M, n, nc))
            for j in range(Nc):
                E = np.zeros((u, nc))
                E[:, j * u:(j + 1) * u] = np.eye(u)
                Phi[:, j] = F[:, j, :, :n] @ prev + F[:, j, :, n:] @ E
                prev = Phi[:, j]
                Mj = np.zeros((M, n, n))
                if j == Nc - 1:
                    Mj = S
                else:
                    Mj[:, :x, :x] = Qd[:, j]
                Hc += np.swapaxes(Phi[:, j], -1, -2) @ Mj @ Phi[:, j]
                Hc[:, j * u:(j + 1) * u, j * u:(j + 1) * u] += p.Rt[:, j]
                if j > 0:
                    blk = p.s[:, None, None] * np.eye(u)
                    Hc[:, j * u:(j + 1) * u, (j - 1) * u:j * u] -= blk
                    Hc[:, (j - 1) * u:j * u, j * u:(j + 1) * u] -= blk
            self.Phi = Phi
            Hsum = self.allreduce(Hc.sum(0))
            if Dc is not None:
                Hsum = Hsum + (Dc if np.ndim(Dc) == 2 else np.diag(Dc))
            self.Hc_chol = np.linalg.cholesky(Hsum)
        return self

    def solve(self, gx, gu, gc_extra=None):
        """gx (M,N,x), gu (M,N,u) per-particle gradients (consensus-stage gu are summed over
        particles); gc_extra (Nc*u,) added once to the consensus gradient. Returns dX, dU."""
        p = self.p
        M, N, x, u, Nc, n, w = p.M, p.N, p.x, p.u, p.Nc, self.n, self.w
        F = self.F
        s = np.zeros((M, n))
        s[:, :x] = gx[:, N - 1]
        k = np.zeros((M, N, u))
        for j in range(N - 1, Nc - 1, -1):
            h = np.einsum("mab,ma->mb", F[:, j], s)
            hu = h[:, n:] + gu[:, j]
            k[:, j] = np.einsum("mab,mb->ma", self.Huu_inv[:, j], hu)
            s = h[:, :n] - np.einsum("mab,ma->mb", self.K[:, j], hu)
            if j > 0:
                s[:, :x] += gx[:, j - 1]
        dU = np.zeros((M, N, u))
        xi = np.zeros((M, n))
        if Nc > 0:
            nc = Nc * u
            gc = np.zeros((M, nc))
            for j in range(Nc):
                mj = np.zeros((M, n))
                if j == Nc - 1:
                    mj = s
                else:
                    mj[:, :x] = gx[:, j]
                gc += np.einsum("mab,ma->mb", self.Phi[:, j], mj)
                gc[:, j * u:(j + 1) * u] += gu[:, j]
            gsum = self.allreduce(gc.sum(0))
            if gc_extra is not None:
                gsum = gsum + gc_extra
            L = self.Hc_chol
            duc = -np.linalg.solve(L.T, np.linalg.solve(L, gsum))
            dU[:, :Nc] = duc.reshape(Nc, u)[None]
        dX = np.zeros((M, N, x))
        for j in range(N):
            if j >= Nc:
                dU[:, j] = -np.einsum("mab,mb->ma", self.K[:, j], xi) - k[:, j]
            xi = np.einsum("mab,mb->ma", F[:, j, :, :n], xi) + np.einsum("mab,mb->ma", F[:, j, :, n:], dU[:, j])
            dX[:, j] = xi[:, :x]
        return dX, dU


def _bounds(p: Problem):
    """Bound arrays with the consensus-control bounds taken from particle 0 (lqp_utils.jl:329-330)."""
    M, N, x, u, Nc = p.M, p.N, p.x, p.u, p.Nc
    inf = np.inf
    lx = p.x_l if p.has_xb else np.full((M, N, x), -inf)
    ux = p.x_u if p.has_xb else np.full((M, N, x), inf)
    lu = np.array(p.u_l) if p.has_ub else np.full((M, N, u), -inf)
    uu = np.array(p.u_u) if p.has_ub else np.full((M, N, u), inf)
    return lx, ux, lu, uu


def ipm_solve(p: Problem, allreduce=None, allreduce_min=None, allreduce_max=None, tol=1e-12, max_iter=60,
              verbose=False):
    """Mehrotra predictor-corrector; returns X (M,N,x), U (M,N,u), info.

    Shared (consensus) controls carry ONE set of slacks/multipliers (those of particle 0's
    bounds); per-particle copies of them are kept identical so the arrays stay rectangular.
    `allreduce*` hooks model the cross-shard reductions (sum / min / max)."""
    ar = allreduce or (lambda a: a)
    armin = allreduce_min or (lambda a: a)
    armax = allreduce_max or (lambda a: a)
    M, N, x, u, Nc = p.M, p.N, p.x, p.u, p.Nc
    lq = StructuredLQ(p, allreduce=ar)
    lx, ux, lu, uu = _bounds(p)
    if Nc > 0:  # consensus bounds from (global) particle 0 — caller passes them replicated
        lu[:, :Nc], uu[:, :Nc] = lu[0:1, :Nc], uu[0:1, :Nc]

    # ---- unconstrained Newton step from a dynamics-consistent base point ---------------------
    U = np.array(p.U_prev)
    if Nc > 0:
        U[:, :Nc] = U[0:1, :Nc]  # base point must respect consensus (shard-local particle 0 is fine: any value works)
        U[:, :Nc] = 0.0
    X = p.rollout(U)
    gx, gu, gc0 = p.gradient(X, U)
    lq.factor()
    gce = np.zeros(Nc * u)
    if Nc > 0:
        gce[:u] = ar(gc0.sum(0))
    dX, dU = lq.solve(gx, gu, gce if Nc > 0 else None)
    X, U = X + dX, U + dU
    info = dict(iters=0, status="unconstrained")
    if not (p.has_xb or p.has_ub):
        return X, U, info
    viol = max(np.max(lx - X), np.max(X - ux), np.max(lu - U), np.max(U - uu))
    if armax(viol) <= 0.0:
        info["status"] = "unconstrained-feasible"
        return X, U, info

    # ---- IPM ------------------------------------------------------------------------------------
    # weights: a consensus bound is ONE constraint although stored M times
    wu = np.ones((M, N, u))
    if Nc > 0:
        wu[:, :Nc] = 0.0
        wu[0, :Nc] = 1.0  # counted on (global) particle 0 only; hook below handles shards
    cons_owner = getattr(p, "owns_consensus", True)
    if Nc > 0 and not cons_owner:
        wu[0, :Nc] = 0.0
    fin = lambda a: np.isfinite(a)
    mlx, mux, mlu, muu = fin(lx), fin(ux), fin(lu), fin(uu)
    m_cnt = ar(np.array([float(mlx.sum() + mux.sum() + (wu * mlu).sum() + (wu * muu).sum())]))[0]

    # pull the controls strictly inside their box (10 % of the width, or 0.1*max(1,|bound|) for a
    # one-sided bound) and re-roll the states: control slack residuals start at zero
    wid = np.where(mlu & muu, uu - lu, np.maximum(1.0, np.where(mlu, np.abs(lu), np.where(muu, np.abs(uu), 1.0))))
    lo_in = np.where(mlu, lu + 0.1 * wid, -np.inf)
    hi_in = np.where(muu, uu - 0.1 * wid, np.inf)
    U = np.clip(U, lo_in, hi_in)
    X = p.rollout(U)

    def init_slack(z, lo, hi, mlo, mhi):
        width = np.where(mlo & mhi, hi - lo, 1.0)
        thr = np.maximum(1e-2 * width, 1e-4)
        tl = np.where(mlo, np.maximum(z - np.where(mlo, lo, 0.0), thr), 1.0)
        tu = np.where(mhi, np.maximum(np.where(mhi, hi, 0.0) - z, thr), 1.0)
        return tl, tu

    tlx, tux = init_slack(X, lx, ux, mlx, mux)
    tlu, tuu = init_slack(U, lu, uu, mlu, muu)
    mu0 = 1.0
    llx, lux = np.where(mlx, mu0 / tlx, 0.0), np.where(mux, mu0 / tux, 0.0)
    llu, luu = np.where(mlu, mu0 / tlu, 0.0), np.where(muu, mu0 / tuu, 0.0)

    def comp_sum(tl, ll, tu, lu_, ml, mh, wgt=1.0):
        return float(np.sum(wgt * (np.where(ml, tl * ll, 0.0) + np.where(mh, tu * lu_, 0.0))))

    nu = 1.0
    mu_peak = 1.0  # complementarity tolerance relative to the dual scale (solver.hip)
    for it in range(1, max_iter + 1):
        mu = ar(np.array([comp_sum(tlx, llx, tux, lux, mlx, mux) + comp_sum(tlu, llu, tuu, luu, mlu, muu, wu)]))[0] / m_cnt
        # slack residuals
        rlx, rux = np.where(mlx, X - np.where(mlx, lx, 0) - tlx, 0.0), np.where(mux, np.where(mux, ux, 0) - X - tux, 0.0)
        rlu, ruu = np.where(mlu, U - np.where(mlu, lu, 0) - tlu, 0.0), np.where(muu, np.where(muu, uu, 0) - U - tuu, 0.0)
        res = armax(max(np.max(np.abs(rlx)), np.max(np.abs(rux)), np.max(np.abs(rlu)), np.max(np.abs(ruu))))
        if verbose:
            print(f"ipm it {it:2d} mu {mu:9.3e} res {res:9.3e} nu {nu:9.3e}")
        mu_peak = max(mu_peak, mu)
        if mu <= tol * mu_peak and res <= tol and nu <= 1e-8:
            break
        Dx = np.where(mlx, llx / tlx, 0.0) + np.where(mux, lux / tux, 0.0)
        Du = np.where(mlu, llu / tlu, 0.0) + np.where(muu, luu / tuu, 0.0)
        Dc = Du[0, :Nc].reshape(-1) if Nc > 0 else None
        if Nc > 0 and not cons_owner:
            Dc = np.zeros(Nc * u)
        if Nc > 0:
            Dc = ar(Dc)  # owner shard contributes, others add zeros
        lq.factor(Dx, Du, Dc)
        gx, gu, gc0 = p.gradient(X, U)

        def newton(sig_mu, cx=None, cu=None):
            clx, cux, clu, cuu = (0.0, 0.0, 0.0, 0.0) if cx is None else (*cx, *cu)
            wlx = np.where(mlx, (sig_mu - clx - llx * rlx) / tlx, 0.0)
            wux = np.where(mux, (sig_mu - cux - lux * rux) / tux, 0.0)
            wlu = np.where(mlu, (sig_mu - clu - llu * rlu) / tlu, 0.0)
            wuu = np.where(muu, (sig_mu - cuu - luu * ruu) / tuu, 0.0)
            gxx = gx - wlx + wux
            guu = gu.copy()
            guu[:, Nc:] += (-wlu + wuu)[:, Nc:]
            gce = np.zeros(Nc * u)
            if Nc > 0:
                gce[:u] = ar(gc0.sum(0))
                own = (-wlu + wuu)[0, :Nc].reshape(-1) if cons_owner else np.zeros(Nc * u)
                gce += ar(own)
            dX, dU = lq.solve(gxx, guu, gce if Nc > 0 else None)
            dtlx, dtux, dtlu, dtuu = dX + rlx, -dX + rux, dU + rlu, -dU + ruu
            dllx = np.where(mlx, wlx - llx - (llx / tlx) * dX, 0.0)
            dlux = np.where(mux, wux - lux + (lux / tux) * dX, 0.0)
            dllu = np.where(mlu, wlu - llu - (llu / tlu) * dU, 0.0)
            dluu = np.where(muu, wuu - luu + (luu / tuu) * dU, 0.0)
            return dX, dU, (dtlx, dtux, dtlu, dtuu), (dllx, dlux, dllu, dluu)

        def max_step(ts, dts, ls, dls, masks):
            a = 1.0
            for v, dv, m in zip(list(ts) + list(ls), list(dts) + list(dls), list(masks) * 2):
                neg = m & (dv < 0)
                if neg.any():
                    a = min(a, float(np.min(-v[neg] / dv[neg])))
            return armin(a)

        ts, ls, masks = (tlx, tux, tlu, tuu), (llx, lux, llu, luu), (mlx, mux, mlu, muu)
        dXa, dUa, dta, dla = newton(0.0)
        a_aff = max_step(ts, dta, ls, dla, masks)
        mu_aff = ar(np.array([
            float(np.sum(np.where(mlx, (tlx + a_aff * dta[0]) * (llx + a_aff * dla[0]), 0.0))
                  + np.sum(np.where(mux, (tux + a_aff * dta[1]) * (lux + a_aff * dla[1]), 0.0))
                  + np.sum(wu * np.where(mlu, (tlu + a_aff * dta[2]) * (llu + a_aff * dla[2]), 0.0))
                  + np.sum(wu * np.where(muu, (tuu + a_aff * dta[3]) * (luu + a_aff * dla[3]), 0.0)))]))[0] / m_cnt
        sigma = (mu_aff / mu) ** 3
        cx = (dta[0] * dla[0], dta[1] * dla[1])
        cu = (dta[2] * dla[2], dta[3] * dla[3])
        dX, dU, dt, dl = newton(sigma * mu, cx, cu)
        a = max_step(ts, dt, ls, dl, masks)
        a = min(1.0, max(0.99, 1.0 - mu) * a) if a < 1.0 else 1.0
        a = min(1.0, a)
        X, U = X + a * dX, U + a * dU
        tlx, tux, tlu, tuu = (t + a * d for t, d in zip(ts, dt))
        llx, lux, llu, luu = (l_ + a * d for l_, d in zip(ls, dl))
        nu *= (1.0 - a)
        info = dict(iters=it, status="ipm", mu=mu, alpha=a)
    return X, U, info


def active_set_solve(p: Problem, allreduce=None, act0=None, U0=None, max_rounds=25, verbose=False, ignore_xb=False):
    """numpy model of the device's primal-dual active-set iteration on the control boxes (solver.hip `active_set_solve`,
    check-pass variant of the generic kernels): per round ONE structured solve from a dynamics-consistent base point whose
    held controls sit ON their bounds, the held controls penalised by `big` on their step (-/+ big du is their multiplier),
    then the KKT sign check and the active-set update.  Cross-shard reductions: the sums inside StructuredLQ ([Hc | gc]) and
    ONE sum of the change counter per round — exactly what the GPU path all-reduces.  Cold start (act0 / U0 None): the
    equality-only optimum, set = its violated boxes.  Returns X, U, info(rounds, act)."""
    ar = allreduce or (lambda a: a)
    M, N, u, Nc = p.M, p.N, p.u, p.Nc
    assert (p.has_ub or ignore_xb) and (ignore_xb or not p.has_xb)
    lq = StructuredLQ(p, allreduce=ar)
    _, _, lu, uu = _bounds(p)
    if Nc > 0:
        lu[:, :Nc], uu[:, :Nc] = lu[0:1, :Nc], uu[0:1, :Nc]
    owner = getattr(p, "owns_consensus", True)
    big, tol_p, tol_l = 1e30, 1e-13, 1e-11

    def newton(Xb, Ub, Du):
        gx, gu, gc0 = p.gradient(Xb, Ub)
        gce = np.zeros(Nc * u)
        if Nc > 0:
            gce[:u] = ar(gc0.sum(0))
        Dc = None
        if Nc > 0:  # a consensus bound is ONE constraint: its penalty enters the summed system once, from the owner shard
            Dc = ar(Du[0, :Nc].reshape(-1) if owner else np.zeros(Nc * u))
        lq.factor(None, Du, Dc)
        return lq.solve(gx, gu, gce if Nc > 0 else None)

    if U0 is None:
        U0 = np.array(p.U_prev)
        if Nc > 0:
            U0[:, :Nc] = 0.0
        X0 = p.rollout(U0)
        dX, dU = newton(X0, U0, np.zeros((M, N, u)))
        U0 = U0 + dU
    if act0 is None:
        act0 = np.where(U0 < lu, 1, np.where(U0 > uu, 2, 0))
    act, Ufree = np.array(act0), np.clip(U0, lu, uu)
    for r in range(max_rounds):
        Ub = np.where(act == 1, lu, np.where(act == 2, uu, Ufree))
        Xb = p.rollout(Ub)
        dX, dU = newton(Xb, Ub, np.where(act > 0, big, 0.0))
        lam = np.where(act == 1, -big * dU, big * dU)
        zt = Ub + dU
        rel = (act > 0) & (lam < -tol_l)
        add_l = (act == 0) & (zt < lu - tol_p * np.maximum(1.0, np.abs(lu)))
        add_u = (act == 0) & (zt > uu + tol_p * np.maximum(1.0, np.abs(uu)))
        changes = int(ar(np.array([float(rel.sum() + add_l.sum() + add_u.sum())]))[0])
        if verbose:
            print(f"active set round {r + 1}: {int(rel.sum())} released, {int(add_l.sum() + add_u.sum())} activated")
        if changes == 0:
            return Xb + dX, np.where(act > 0, Ub, np.clip(zt, lu, uu)), dict(rounds=r + 1, act=act)
        act = np.where(rel, 0, np.where(add_l, 1, np.where(add_u, 2, act)))
        Ufree = np.where(act > 0, Ub, zt)  # (held controls are replaced by their bounds above)
    raise RuntimeError("active set did not settle")


def active_set_solve_xb(p: Problem, allreduce=None, max_rounds=30, verbose=False, rho_scale=1e7, act_frac=0.5, tol=1e-9):
    """numpy model of the STATE ROWS of the device's active-set rounds (pmpc_amd/csrc/kernels_xbox.hip, solver.hip mode 4): the cold
    start in two phases — control boxes alone (`active_set_solve(ignore_xb=True)`), then rounds in which a binding state box is a
    row of a semismooth Newton iteration on  s - max(0, s - z) = 0  (s = x - lo or hi - x): held by the penalty rho on the
    diagonal of the stage's state cost and  +-(rho s_b - z)  in its gradient, multiplier z+ = z - rho s_new afterwards (not in a
    particle whose sweep clamped a control), status changes with a hysteresis margin, and per round only the violated rows within
    `act_frac` of the particle's largest violation are newly held.  Accepted when no status (control or state) changed and no held
    row is further than `tol` off its bound.  Cross-shard reductions: the [Hc | gc] sums inside StructuredLQ and ONE sum of
    {changes, open rows} per round — what the GPU path all-reduces.  Returns X, U, info(rounds, phase1_rounds, held)."""
    ar = allreduce or (lambda a: a)
    M, N, x, u, Nc = p.M, p.N, p.x, p.u, p.Nc
    assert p.has_xb
    X, U, info1 = active_set_solve(p, allreduce=ar, ignore_xb=True, verbose=verbose)
    act = info1["act"]
    lq = StructuredLQ(p, allreduce=ar)
    lx, ux, lu, uu = _bounds(p)
    if Nc > 0:
        lu[:, :Nc], uu[:, :Nc] = lu[0:1, :Nc], uu[0:1, :Nc]
    owner = getattr(p, "owns_consensus", True)
    big, tol_p, tol_l = 1e30, 1e-13, 1e-11
    rho = rho_scale * (np.abs(np.einsum("mjrr->mjr", p.Q)).reshape(M, -1).max(1) + p.reg_x)[:, None, None]  # per particle
    st, z = np.zeros((M, N, x), int), np.zeros((M, N, x))

    def classify(Xb, st, zn, finish):
        sl, sh = Xb - lx, ux - Xb
        mg = (10.0 * tol_l + 1e-13 * np.maximum(1.0, np.abs(Xb))) if finish else 0.0
        wl, wh = sl - np.where(st == 1, zn, 0.0), sh - np.where(st == 2, zn, 0.0)
        nst = np.where(wl < np.where(st == 1, mg, -mg), 1, np.where(wh < np.where(st == 2, mg, -mg), 2, 0))
        viol = np.maximum(-sl, -sh)
        vmax = np.where(st == 0, viol, 0.0).reshape(M, -1).max(1)[:, None, None]
        deferred = (st == 0) & (nst != 0) & (viol < act_frac * vmax)
        nst = np.where(deferred, 0, nst)
        zo = np.where((nst != 0) & (nst == st), zn, 0.0)
        return nst, zo, deferred

    def newton(Xb, Ub, Du, st, z):
        gx, gu, gc0 = p.gradient(Xb, Ub)
        sl, sh = Xb - lx, ux - Xb
        Dx = np.where(st != 0, rho, 0.0)
        gx = gx + np.where(st == 1, rho * sl - z, 0.0) - np.where(st == 2, rho * sh - z, 0.0)
        gce = np.zeros(Nc * u)
        Dc = None
        if Nc > 0:
            gce[:u] = ar(gc0.sum(0))
            Dc = ar(Du[0, :Nc].reshape(-1) if owner else np.zeros(Nc * u))
        lq.factor(Dx, Du, Dc)
        return lq.solve(gx, gu, gce if Nc > 0 else None)

    Ub, Xb = U, X  # the first phase's optimum: controls on their bounds where held, states rolled out
    st, z, _ = classify(Xb, st, z, False)
    for r in range(max_rounds):
        dX, dU = newton(Xb, Ub, np.where(act > 0, big, 0.0), st, z)
        lam = np.where(act == 1, -big * dU, big * dU)
        zt = Ub + dU
        rel = (act > 0) & (lam < -tol_l)
        add_l = (act == 0) & (zt < lu - tol_p * np.maximum(1.0, np.abs(lu)))
        add_u = (act == 0) & (zt > uu + tol_p * np.maximum(1.0, np.abs(uu)))
        act = np.where(rel, 0, np.where(add_l, 1, np.where(add_u, 2, act)))
        Ub = np.where(act == 1, lu, np.where(act == 2, uu, np.clip(zt, lu, uu)))
        if Nc > 0:  # (the shared controls' clamping is decided on the summed system: identical on every shard)
            Ub[:, :Nc] = Ub[0:1, :Nc]
        Xb = p.rollout(Ub)  # the forward sweep's states are those of the clamped controls
        clamped = (add_l | add_u).reshape(M, -1).any(1)[:, None, None]
        sl, sh = Xb - lx, ux - Xb
        zn = np.where(clamped, z, np.where(st == 1, z - rho * sl, np.where(st == 2, z - rho * sh, 0.0)))
        zn = np.where(st == 0, 0.0, zn)
        nst, zo, deferred = classify(Xb, st, zn, True)
        sv, bd = np.where(nst == 1, sl, sh), np.where(nst == 1, lx, ux)
        still = (nst == st) & (nst != 0)
        with np.errstate(invalid="ignore"):
            open_rows = still & ((np.abs(sv) > tol * np.maximum(1.0, np.abs(bd))) | (np.abs(zo - z) > 1e-6 * np.maximum(1.0, np.abs(zo))))
        n_ctl = float(rel.sum() + add_l.sum() + add_u.sum())
        n_row = float(((nst != st) | deferred).sum())
        tot = ar(np.array([n_ctl + n_row, float(open_rows.sum())]))
        st, z = nst, zo
        if verbose:
            print(f"state rows round {r + 1}: {int(n_ctl)} control changes, {int(n_row)} row changes, {int(open_rows.sum())} open")
        if tot[0] == 0 and tot[1] == 0:
            return Xb, Ub, dict(rounds=r + 1, phase1_rounds=info1["rounds"], held=int((st != 0).sum()))
    raise RuntimeError("state rows did not settle")
