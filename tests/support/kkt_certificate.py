"""An algorithm-independent optimality certificate of a returned (X, U) for the joint QP of PMPC.jl/src/lqp_utils.jl:2-393 with one
consensus block (0 <= Nc <= N), control boxes and, optionally, one second-order cone  |W u + w0| <= v'u + v0  per stage — computed in
numpy from the ABI data alone (py layout: fx (M,N,x,x), fu (M,N,x,u) row-major blocks), with nothing of the solver's own algorithm:

  primal   dynamics  x_j = f_j + fx_j (x_{j-1} - Xp_{j-1}) + fu_j (u_j - Up_j)   (x_{-1} - Xp_{-1} := 0: lqp_utils.jl:288-296),
           boxes, cones, equality of the shared controls across the particles;
  dual     the multipliers of the dynamics are DETERMINED by stationarity in the states (a backward recursion through fx');
           what is left is the reduced gradient r of every control, summed over the particles for a shared one (`stationarity_shared`,
           relative to M x the gradient scale):
               r = 0                     strictly inside box and cone,
               r_b <= 0 / >= 0           at the upper / lower bound (the box multiplier's sign),
               r + lam grad g = same     on the boundary of the cone g(u) = |W u + w0| - v'u - v0 = 0, lam >= 0 fitted by least squares.

Returns a dict of max-norm residuals, each relative to the scale of the quantities it compares."""
import numpy as np


def kkt_certificate(x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref, reg_x, reg_u, Nc, u_l, u_u, X, U, soc=None, tol_act=1e-9):
    M, N, x = f.shape
    u = fu.shape[-1]
    Ncc = N if Nc < 0 else int(Nc)
    # ---- primal ------------------------------------------------------------------------------------------------------------------
    dXm = np.concatenate([np.zeros((M, 1, x)), X[:, :-1] - X_prev[:, :-1]], 1)  # x_{j-1} - Xp_{j-1}, zero for j = 0
    pred = f + np.einsum("mnrt,mnt->mnr", fx, dXm) + np.einsum("mnrt,mnt->mnr", fu, U - U_prev)
    res = dict(dynamics=np.abs(X - pred).max() / max(1.0, np.abs(X).max()))
    res["consensus"] = np.abs(U[:, :Ncc] - U[:1, :Ncc]).max() if Ncc > 0 else 0.0
    lo = np.broadcast_to(u_l, U.shape).copy()
    hi = np.broadcast_to(u_u, U.shape).copy()
    lo[:, :Ncc], hi[:, :Ncc] = lo[:1, :Ncc], hi[:1, :Ncc]  # the shared controls take particle 0's bounds (lqp_utils.jl:329-330)
    scale_u = np.maximum(1.0, np.maximum(np.abs(lo), np.abs(hi)))
    res["box"] = max(0.0, float(((lo - U) / scale_u).max()), float(((U - hi) / scale_u).max()))
    # ---- gradient of the cost, costates, reduced gradient ------------------------------------------------------------------------
    gx = np.einsum("mnrt,mnt->mnr", Q, X - X_ref) + reg_x * (X - X_prev)
    gu = np.einsum("mnrt,mnt->mnr", R, U - U_ref) + reg_u * (U - U_prev)
    nu = np.empty_like(X)  # d cost-to-go / d x_j along the dynamics
    nu[:, N - 1] = gx[:, N - 1]
    for j in range(N - 2, -1, -1):
        nu[:, j] = gx[:, j] + np.einsum("mtr,mt->mr", fx[:, j + 1], nu[:, j + 1])
    r = gu + np.einsum("mnrt,mnr->mnt", fu, nu)  # (M, N, u)
    rs = r.copy()
    if Ncc > 0:  # a shared control: ONE variable whose gradient is the sum over the particles
        rs[:, :Ncc] = r[:, :Ncc].sum(0, keepdims=True)
    gscale = max(1.0, np.abs(gu).max(), np.abs(nu).max())
    at_lo = U <= lo + tol_act * scale_u
    at_hi = U >= hi - tol_act * scale_u
    if soc is not None:
        # stationarity with a cone row  s = A u + c in K,  A = [v'; W]:  r = A'z  with  z in K (self-dual),  z's = 0:
        #   inside the cone z = 0;  on its boundary z = lam (1, -sb / |sb|), lam >= 0;  at the apex s = 0 ANY z in K.
        # z is fitted per active stage by least squares on the components strictly inside their boxes and projected onto the cone.
        W, w0, v, v0 = (np.asarray(soc[k], dtype=np.float64) for k in ("W", "w0", "v", "v0"))
        A = np.vstack([v[None, :], W])  # (1 + q, u)
        sb = np.einsum("qk,mnk->mnq", W, U) + w0
        nb = np.linalg.norm(sb, axis=-1)
        s0 = U @ v + v0
        res["cone"] = max(0.0, float(((nb - s0) / np.maximum(1.0, np.abs(s0))).max()))
        active = nb >= s0 - tol_act * np.maximum(1.0, np.abs(s0))
        if Ncc > 0:
            active[1:, :Ncc] = False  # the shared stages' cone is ONE cone, fitted on the summed gradient in particle 0's row
        free = ~(at_lo | at_hi)
        n_apex = 0
        for i, j in zip(*np.nonzero(active)):
            fr = free[i, j]
            if s0[i, j] > tol_act * 10 and nb[i, j] > 0.0:  # boundary, away from the apex
                d = np.concatenate([[1.0], -sb[i, j] / nb[i, j]])
                a_d = A.T @ d
                den = float(a_d[fr] @ a_d[fr])
                lam = max(float(rs[i, j][fr] @ a_d[fr]) / den, 0.0) if den > 0.0 else 0.0
                z = lam * d
            else:  # apex
                n_apex += 1
                z = np.linalg.lstsq(A.T[fr], rs[i, j][fr], rcond=None)[0] if fr.any() else np.zeros(A.shape[0])
                nz = np.linalg.norm(z[1:])
                if nz > z[0]:
                    if not np.any(fr & (v != 0.0)):
                        # z0 only meets components that sit on a bound (the thrust at T = 0): the least-squares fit left it at its
                        # minimum-norm value; any z0 >= |zb| is in the cone, and the bound's multiplier takes the rest (sign checked below)
                        z[0] = nz
                    else:  # projection onto the second-order cone
                        z = np.zeros_like(z) if nz <= -z[0] else 0.5 * (1.0 + z[0] / nz) * np.concatenate([[nz], z[1:]])
            rs[i, j] = rs[i, j] - A.T @ z
        res["cone_active"] = int(active.sum())
        res["cone_apex"] = n_apex
    viol = np.where(at_lo & at_hi, 0.0, np.where(at_lo, np.maximum(-rs, 0.0), np.where(at_hi, np.maximum(rs, 0.0), np.abs(rs))))
    res["stationarity_shared"] = 0.0
    if Ncc > 0:  # a shared control's condition is on a SUM of M particle gradients: relative to M x the gradient scale
        res["stationarity_shared"] = float(viol[0, :Ncc].max()) / (gscale * M)
        viol[:, :Ncc] = 0.0
    res["stationarity"] = float(viol.max()) / gscale
    res["active_bounds"] = int((at_lo | at_hi).sum())
    return res


def smoothed_cone_certificate(x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref, reg_x, reg_u, Nc, u_l, u_u, X, U, alpha, k=None, eps=1e-3):
    """Optimality certificate of a returned (X, U) for the reference's DEFAULT objective with log-barrier smoothing of the control boxes
    (PMPC.jl/src/main.jl:204-262):   min (1 + eps) sum y_i + (1 - eps) k t - (1 / alpha) sum log(alpha slack)   s.t.  J_i(z_i) <= y_i + t,  y >= 0,
    the dynamics and the consensus of the first Nc controls — from the ABI data alone, no solver in between.
    The epigraph multipliers are not returned by any ABI; they are DETERMINED by the point: particle i's own (non-shared) controls are stationary
    iff  lam_i r_J + r_B = 0  with r_J the reduced gradient of J_i (adjoint recursion) and r_B the barrier's gradient — lam_i is the least-squares
    fit, its residual the first check.  Then: lam_i in [0, 1 + eps], sum lam = (1 - eps) k, lam_i = 1 + eps above the threshold cost, 0 below it
    (the threshold = the cost of the particles with a fractional multiplier), and stationarity of the SHARED controls under the summed gradient.
    Returns a dict of residuals (relative) and the multipliers."""
    M, N, x = f.shape
    u = fu.shape[-1]
    Ncc = N if Nc < 0 else int(Nc)
    k = M if k is None or k < 0 else k
    cap, K = 1.0 + eps, (1.0 - eps) * k
    dXm = np.concatenate([np.zeros((M, 1, x)), X[:, :-1] - X_prev[:, :-1]], 1)
    pred = f + np.einsum("mnrt,mnt->mnr", fx, dXm) + np.einsum("mnrt,mnt->mnr", fu, U - U_prev)
    res = dict(dynamics=np.abs(X - pred).max() / max(1.0, np.abs(X).max()), consensus=(np.abs(U[:, :Ncc] - U[:1, :Ncc]).max() if Ncc else 0.0))
    eX, eU, pX, pU = X - X_ref, U - U_ref, X - X_prev, U - U_prev
    gx = np.einsum("mnrt,mnt->mnr", Q, eX) + reg_x * pX
    gu = np.einsum("mnrt,mnt->mnr", R, eU) + reg_u * pU
    J = 0.5 * (np.einsum("mnr,mnr->m", eX, np.einsum("mnrt,mnt->mnr", Q, eX)) + np.einsum("mnr,mnr->m", eU, np.einsum("mnrt,mnt->mnr", R, eU))
               + reg_x * np.einsum("mnr,mnr->m", pX, pX) + reg_u * np.einsum("mnr,mnr->m", pU, pU))
    nu = np.empty_like(X)
    nu[:, N - 1] = gx[:, N - 1]
    for j in range(N - 2, -1, -1):
        nu[:, j] = gx[:, j] + np.einsum("mtr,mt->mr", fx[:, j + 1], nu[:, j + 1])
    rJ = gu + np.einsum("mnrt,mnr->mnt", fu, nu)
    lo, hi = np.broadcast_to(u_l, U.shape), np.broadcast_to(u_u, U.shape)
    sl, sh = U - lo, hi - U
    res["slack"] = float(min(sl.min(), sh.min()))
    rB = (-1.0 / sl + 1.0 / sh) / alpha
    own = slice(Ncc, N)
    num = -np.einsum("mnr,mnr->m", rJ[:, own], rB[:, own])
    den = np.einsum("mnr,mnr->m", rJ[:, own], rJ[:, own])
    lam = num / np.where(den > 0.0, den, 1.0)
    scale = np.maximum(1.0, np.abs(rB[:, own]).reshape(M, -1).max(1))
    res["own_controls"] = float((np.abs(lam[:, None, None] * rJ[:, own] + rB[:, own]).reshape(M, -1).max(1) / scale).max())
    res["lam_range"] = float(max(0.0, -lam.min(), lam.max() - cap))
    res["lam_sum"] = abs(float(lam.sum()) - K) / max(1.0, K)
    frac = (lam > 1e-6 * cap) & (lam < cap * (1 - 1e-6))
    jscale = max(1.0, np.abs(J).max())
    if frac.any():
        t = float(J[frac].mean())
        res["threshold_spread"] = float(np.abs(J[frac] - t).max()) / jscale
    else:
        t = 0.5 * (J[lam <= 1e-6 * cap].max(initial=-np.inf) + J[lam >= cap * (1 - 1e-6)].min(initial=np.inf))
        res["threshold_spread"] = 0.0
    res["complementarity"] = float(max(0.0, (J[lam <= 1e-6 * cap] - t).max(initial=0.0), (t - J[lam >= cap * (1 - 1e-6)]).max(initial=0.0))) / jscale
    if Ncc:
        rs = np.einsum("m,mnr->nr", lam, rJ[:, :Ncc]) + rB[0, :Ncc]  # the shared controls' box rows are particle 0's, once (lqp_utils.jl:329-330)
        res["shared_controls"] = float(np.abs(rs).max() / max(1.0, np.abs(rB[0, :Ncc]).max(), np.abs(lam[:, None, None] * rJ[:, :Ncc]).sum(0).max()))
    res["fractional_multipliers"] = int(frac.sum())
    res["_lam"], res["_t"], res["_J"] = lam, t, J
    return res


def hard_cone_certificate(x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref, reg_x, reg_u, Nc, u_l, u_u, X, U, k=None, eps=1e-3, tol_act=1e-9, tie_tol=1e-8):
    """The same for HARD control boxes (main.jl:204-238 without smoothing).  Here a particle's own stationarity does not name its multiplier
    (lam_i r_J = 0 holds for any lam_i once r_J = 0), so the checks are: (1) every particle is at its conditional optimum given the shared
    controls — reduced gradient of J_i zero inside the box, of the right sign on a bound; (2) multipliers exist: lam_i = 1 + eps strictly above
    the threshold cost, 0 strictly below, in [0, 1 + eps] on it, summing to (1 - eps) k, with the shared controls stationary under
    sum lam_i r_J (bounded least squares over the costs on the threshold)."""
    from scipy.optimize import lsq_linear

    M, N, x = f.shape
    u = fu.shape[-1]
    Ncc = N if Nc < 0 else int(Nc)
    k = M if k is None or k < 0 else k
    cap, K = 1.0 + eps, (1.0 - eps) * k
    dXm = np.concatenate([np.zeros((M, 1, x)), X[:, :-1] - X_prev[:, :-1]], 1)
    pred = f + np.einsum("mnrt,mnt->mnr", fx, dXm) + np.einsum("mnrt,mnt->mnr", fu, U - U_prev)
    res = dict(dynamics=np.abs(X - pred).max() / max(1.0, np.abs(X).max()), consensus=(np.abs(U[:, :Ncc] - U[:1, :Ncc]).max() if Ncc else 0.0))
    eX, eU, pX, pU = X - X_ref, U - U_ref, X - X_prev, U - U_prev
    gx = np.einsum("mnrt,mnt->mnr", Q, eX) + reg_x * pX
    gu = np.einsum("mnrt,mnt->mnr", R, eU) + reg_u * pU
    J = 0.5 * (np.einsum("mnr,mnr->m", eX, np.einsum("mnrt,mnt->mnr", Q, eX)) + np.einsum("mnr,mnr->m", eU, np.einsum("mnrt,mnt->mnr", R, eU))
               + reg_x * np.einsum("mnr,mnr->m", pX, pX) + reg_u * np.einsum("mnr,mnr->m", pU, pU))
    nu = np.empty_like(X)
    nu[:, N - 1] = gx[:, N - 1]
    for j in range(N - 2, -1, -1):
        nu[:, j] = gx[:, j] + np.einsum("mtr,mt->mr", fx[:, j + 1], nu[:, j + 1])
    rJ = gu + np.einsum("mnrt,mnr->mnt", fu, nu)
    lo, hi = np.broadcast_to(u_l, U.shape).copy(), np.broadcast_to(u_u, U.shape).copy()
    lo[:, :Ncc], hi[:, :Ncc] = lo[:1, :Ncc], hi[:1, :Ncc]
    scale_u = np.maximum(1.0, np.maximum(np.abs(lo), np.abs(hi)))
    res["box"] = max(0.0, float(((lo - U) / scale_u).max()), float(((U - hi) / scale_u).max()))
    at_lo, at_hi = U <= lo + tol_act * scale_u, U >= hi - tol_act * scale_u
    gscale = max(1.0, np.abs(gu).max(), np.abs(nu).max())
    own = slice(Ncc, N)
    v = np.where(at_lo & at_hi, 0.0, np.where(at_lo, np.maximum(-rJ, 0.0), np.where(at_hi, np.maximum(rJ, 0.0), np.abs(rJ))))
    res["own_controls"] = float(v[:, own].max()) / gscale
    # multipliers: fixed off the threshold, fitted on it
    order = np.argsort(-J)
    n_hi = int(np.floor(K / cap + 1e-12))
    t = float(J[order[min(n_hi, M - 1)]])  # the cost that carries the remainder
    jscale = max(1.0, np.abs(J).max())
    above, below = J > t + tie_tol * jscale, J < t - tie_tol * jscale
    tied = ~(above | below)
    lam = np.where(above, cap, 0.0)
    rem = K - cap * above.sum()
    res["tied"] = int(tied.sum())
    res["remainder_feasible"] = float(max(0.0, -rem, rem - cap * tied.sum())) / max(1.0, K)
    if Ncc:
        rc = rJ[:, :Ncc].reshape(M, -1)  # (M, Ncc u)
        fixed = lam @ rc
        free_c = ~(at_lo[0, :Ncc] | at_hi[0, :Ncc]).reshape(-1)
        wsum = 1e3 * max(1.0, np.abs(rc).max())
        A = np.vstack([rc[tied][:, free_c].T, wsum * np.ones((1, tied.sum()))])
        b = np.concatenate([-fixed[free_c], [wsum * rem]])
        sol = lsq_linear(A, b, bounds=(0.0, cap), tol=1e-14, max_iter=200)
        lam[tied] = sol.x
        rs = lam @ rc
        vs = np.where((at_lo[0, :Ncc] & at_hi[0, :Ncc]).reshape(-1), 0.0, np.where(at_lo[0, :Ncc].reshape(-1), np.maximum(-rs, 0.0), np.where(at_hi[0, :Ncc].reshape(-1), np.maximum(rs, 0.0), np.abs(rs))))
        res["shared_controls"] = float(vs.max()) / (gscale * max(1.0, K))
    res["lam_sum"] = abs(float(lam.sum()) - K) / max(1.0, K)
    res["_lam"], res["_t"], res["_J"] = lam, t, J
    return res
