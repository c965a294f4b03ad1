"""How far an fp32 Riccati recursion lands from the fp64 one on config D's data (numpy, CPU): the gains K_j and the unconstrained
optimum (X, U) of one particle's LQ problem, fp32 arithmetic throughout against fp64.  Companion of stage_f32_vs_f64.hip (VERDICT r04
item 9): the kernel is 1.4 - 1.6x faster in fp32, this is what a refinement round would have to remove."""
import sys

import numpy as np

sys.path.insert(0, ".")
from pmpc_amd import dynamics as dyn

M, N = 64, 50
prob = dyn.make_quadrotor_problem(M=M, N=N)
rng = np.random.default_rng(0)
Xp = prob["X_prev"] + 0.3 * rng.standard_normal(prob["X_prev"].shape)  # a linearisation point away from hover
Up = prob["U_prev"] + 0.3 * rng.standard_normal(prob["U_prev"].shape)
f, fx, fu = prob["f_fx_fu_fn"](np.concatenate([prob["x0"][:, None], Xp[:, :-1]], 1), Up)


def solve(dt):
    c = lambda a: np.asarray(a, dtype=dt)
    F, A, B, Q, R = c(f), c(fx), c(fu), c(prob["Q"]), c(prob["R"])
    Xr, Ur, Xpp, Upp, x0 = c(prob["X_ref"]), c(prob["U_ref"]), c(Xp), c(Up), c(prob["x0"])
    rx, ru = dt(prob["reg_x"]), dt(prob["reg_u"])
    x, u = 12, 4
    Ks, ks = np.zeros((M, N, u, x), dt), np.zeros((M, N, u), dt)
    # affine dynamics x_j = A_j x_{j-1} + B_j u_j + c_j
    cj = F - np.einsum("mnrt,mnt->mnr", B, Upp)
    cj[:, 1:] -= np.einsum("mnrt,mnt->mnr", A[:, 1:], Xpp[:, :-1])
    S = Q[:, N - 1] + rx * np.eye(x, dtype=dt)
    s = -(np.einsum("mrt,mt->mr", Q[:, N - 1], Xr[:, N - 1]) + rx * Xpp[:, N - 1])
    for j in range(N - 1, -1, -1):
        Aj, Bj = A[:, j], B[:, j]
        SB = S @ Bj
        Huu = R[:, j] + ru * np.eye(u, dtype=dt) + np.swapaxes(Bj, -1, -2) @ SB
        Hux = np.swapaxes(SB, -1, -2) @ Aj
        w = s + np.einsum("mrt,mt->mr", S, cj[:, j])
        hu = -(np.einsum("mrt,mt->mr", R[:, j], Ur[:, j]) + ru * Upp[:, j]) + np.einsum("mtr,mt->mr", Bj, w)
        hx = np.einsum("mtr,mt->mr", Aj, w)
        L = np.linalg.cholesky(Huu)
        K = np.linalg.solve(Huu, Hux).astype(dt)
        k = np.linalg.solve(Huu, hu[..., None])[..., 0].astype(dt)
        Ks[:, j], ks[:, j] = K, k
        if j == 0:
            break
        S = (Q[:, j - 1] + rx * np.eye(x, dtype=dt) + np.swapaxes(Aj, -1, -2) @ S @ Aj - np.swapaxes(Hux, -1, -2) @ K).astype(dt)
        S = dt(0.5) * (S + np.swapaxes(S, -1, -2))
        s = (hx - np.einsum("mtr,mt->mr", K, hu) - (np.einsum("mrt,mt->mr", Q[:, j - 1], Xr[:, j - 1]) + rx * Xpp[:, j - 1])).astype(dt)
    X, U = np.zeros((M, N, x), dt), np.zeros((M, N, u), dt)
    xprev = np.zeros((M, x), dt)  # stage 0 has no incoming state (A~_0 = 0, lqp_utils.jl:288-296)
    for j in range(N):
        Axp = np.einsum("mrt,mt->mr", A[:, j], xprev) if j else np.zeros((M, x), dt)
        uj = -(np.einsum("mrt,mt->mr", Ks[:, j], xprev) if j else 0) - ks[:, j]
        xj = Axp + np.einsum("mrt,mt->mr", B[:, j], uj) + cj[:, j]
        X[:, j], U[:, j], xprev = xj, uj, xj
    return Ks, X, U


K64, X64, U64 = solve(np.float64)
K32, X32, U32 = solve(np.float32)
rel = lambda a, b: np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b)
print(f"fp32 vs fp64 Riccati, quadrotor x12 u4 N={N}, {M} particles: gains rel 2-norm diff {rel(K32, K64):.2e}, trajectories X {rel(X32, X64):.2e}, U {rel(U32, U64):.2e} "
      f"(north star: 1e-6)")
