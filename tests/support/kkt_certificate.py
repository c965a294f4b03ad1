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
