"""Device-resident API on the GPU: on-device linearisation vs the numpy specification, device solve vs
the host-pointer ABI, and full-size (BASELINE config C/D shape) size-independent properties."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def solver():
    import torch

    from pmpc_amd.device import DeviceSolver

    assert torch.cuda.is_available()
    s = DeviceSolver(0)
    yield s
    s.close()


def _lin_args(prob):
    X_ = np.concatenate([prob["x0"][:, None, :], prob["X_prev"][:, :-1]], 1)
    return X_, prob["U_prev"]


@pytest.mark.parametrize("model", ["unicycle", "quadrotor"])
def test_linearize_matches_numpy(solver, model):
    import torch

    from pmpc_amd import dynamics as dyn
    from pmpc_amd.device import MODEL_QUADROTOR, MODEL_UNICYCLE, to_device_problem

    rng = np.random.default_rng(1)
    prob = dyn.make_unicycle_problem(M=37, N=11) if model == "unicycle" else dyn.make_quadrotor_problem(M=37, N=11)
    prob["X_prev"] = prob["X_prev"] + 0.3 * rng.standard_normal(prob["X_prev"].shape)
    dU = 0.3 * rng.standard_normal(prob["U_prev"].shape)
    if model == "unicycle":  # keep |u| >= 0.1: the closed form has n/u2^3 cancellations near u2 = 0
        dU = np.sign(dU) * (0.1 + np.abs(dU))
    prob["U_prev"] = prob["U_prev"] + dU
    f, fx, fu = prob["f_fx_fu_fn"](*_lin_args(prob))
    d = to_device_problem(prob)
    fd, fxd, fud = solver.linearize(MODEL_UNICYCLE if model == "unicycle" else MODEL_QUADROTOR, d["x0"], d["X_prev"],
                                    d["U_prev"], d["params"])
    solver.sync()
    # the unicycle closed form divides differences of O(1) terms by u2^2 (tests/dubins_car.py:66-85):
    # device vs host sincos differ in the last ulp, amplified by that cancellation
    rtol = 1e-8 if model == "unicycle" else 1e-12
    np.testing.assert_allclose(fd.cpu().numpy(), f, rtol=rtol, atol=1e-12)
    np.testing.assert_allclose(fxd.cpu().numpy().swapaxes(-1, -2), fx, rtol=rtol, atol=1e-12)
    np.testing.assert_allclose(fud.cpu().numpy().swapaxes(-1, -2), fu, rtol=rtol, atol=1e-12)


def _device_solve(solver, prob, Nc, **kw):
    from pmpc_amd.device import MODEL_QUADROTOR, to_device_problem

    d = to_device_problem(prob)
    f, fx, fu = solver.linearize(MODEL_QUADROTOR, d["x0"], d["X_prev"], d["U_prev"], d["params"])
    X, U, status = solver.lqp_solve(f=f, fx=fx, fu=fu, X_prev=d["X_prev"], U_prev=d["U_prev"], Q=d["Q"], R=d["R"],
                                    X_ref=d["X_ref"], U_ref=d["U_ref"], reg_x=prob["reg_x"], reg_u=prob["reg_u"], Nc=Nc,
                                    x0=d["x0"], lu=d.get("lu"), uu=d.get("uu"), symmetric_cost=True, **kw)
    solver.sync()
    return X.cpu().numpy(), U.cpu().numpy(), status, (f, fx, fu)


@pytest.mark.parametrize("Nc", [0, 1, 7, -1])
@pytest.mark.parametrize("model", ["unicycle", "quadrotor"])
def test_fast_path_matches_generic_path(solver, model, Nc):
    """The register-resident MFMA kernels and the LDS generic kernels solve the same Newton systems."""
    from pmpc_amd import dynamics as dyn
    from pmpc_amd.device import MODEL_QUADROTOR, MODEL_UNICYCLE, to_device_problem

    prob = dyn.make_unicycle_problem(M=33, N=30, Nc=Nc) if model == "unicycle" else dyn.make_quadrotor_problem(M=33, N=50, Nc=Nc)
    d = to_device_problem(prob)
    f, fx, fu = solver.linearize(MODEL_UNICYCLE if model == "unicycle" else MODEL_QUADROTOR, d["x0"], d["X_prev"],
                                 d["U_prev"], d["params"])
    out = {}
    for force in (False, True):
        X, U, status = solver.lqp_solve(f=f, fx=fx, fu=fu, X_prev=d["X_prev"], U_prev=d["U_prev"], Q=d["Q"], R=d["R"],
                                        X_ref=d["X_ref"], U_ref=d["U_ref"], reg_x=prob["reg_x"], reg_u=prob["reg_u"], Nc=Nc,
                                        x0=d["x0"], lu=d["lu"], uu=d["uu"], symmetric_cost=True, force_generic=force)
        solver.sync()
        assert status == 0
        assert solver.last_info["fast_path"] == (0 if force else 1)
        out[force] = (X.cpu().numpy(), U.cpu().numpy())
    assert np.linalg.norm(out[False][0] - out[True][0]) / np.linalg.norm(out[True][0]) < 1e-9
    assert np.linalg.norm(out[False][1] - out[True][1]) / np.linalg.norm(out[True][1]) < 1e-9


def test_quadrotor_device_solve_matches_oracle(solver, oracle):
    from pmpc_amd import dynamics as dyn

    prob = dyn.make_quadrotor_problem(M=24, N=50)
    X, U, status, _ = _device_solve(solver, prob, 1)
    assert status == 0
    f, fx, fu = prob["f_fx_fu_fn"](*_lin_args(prob))
    Xo, Uo = oracle.lqp_solve_py(prob["x0"], f, fx, fu, prob["X_prev"], prob["U_prev"], prob["Q"], prob["R"], prob["X_ref"],
                                 prob["U_ref"], reg_x=prob["reg_x"], reg_u=prob["reg_u"], Nc=1, u_l=prob["u_l"], u_u=prob["u_u"])
    assert np.linalg.norm(X - Xo) / np.linalg.norm(Xo) < 1e-7
    assert np.linalg.norm(U - Uo) / np.linalg.norm(Uo) < 1e-7
    info = solver.last_info  # the thrust / torque boxes are active on this problem: more than the equality-only solve ran
    assert info["ipm_iters"] + info["active_set_rounds"] > 0 and info["max_violation"] > 0


def _rollout_np(prob, f, fx, fu, U):
    f, fx, fu = f.cpu().numpy(), fx.cpu().numpy().swapaxes(-1, -2), fu.cpu().numpy().swapaxes(-1, -2)
    X = np.empty_like(f)
    for j in range(f.shape[1]):  # PMPC.jl/src/types.jl:161-173
        X[:, j] = f[:, j] + np.einsum("mrt,mt->mr", fu[:, j], U[:, j] - prob["U_prev"][:, j])
        if j:
            X[:, j] += np.einsum("mrt,mt->mr", fx[:, j], X[:, j - 1] - prob["X_prev"][:, j - 1])
    return X


def _objective(prob, X, U):
    dx, du = X - prob["X_ref"], U - prob["U_ref"]
    return 0.5 * (np.einsum("mnr,mnrt,mnt->", dx, prob["Q"], dx) + np.einsum("mnr,mnrt,mnt->", du, prob["R"], du)
                  + prob["reg_x"] * np.sum((X - prob["X_prev"]) ** 2) + prob["reg_u"] * np.sum((U - prob["U_prev"]) ** 2))


@pytest.mark.parametrize("M,Nc", [(1024, 1), (4096, 1), (1024, -1), (1024, 8)])
def test_full_size_properties(solver, M, Nc):
    """BASELINE configs C (M=1024) and D (M=4096), N=50, x12 u4, Nc=1 — and the reference's default full consensus
    (Nc = N, nc = 200: condensing kernel with several column-tile waves per particle, blocked dense Cholesky) —
    through properties that do not need the oracle: consensus equality, box feasibility, exact linearised dynamics,
    and optimality by perturbation (no feasible, consensus-respecting direction from a random sample lowers the
    objective of PMPC.jl/src/lqp_utils.jl:2-216)."""
    from pmpc_amd import dynamics as dyn

    prob = dyn.make_quadrotor_problem(M=M, N=50, Nc=Nc)
    X, U, status, (f, fx, fu) = _device_solve(solver, prob, Nc)
    assert status == 0 and np.all(np.isfinite(X)) and np.all(np.isfinite(U))
    k = 50 if Nc < 0 else Nc
    assert np.all(U[:, :k] == U[0:1, :k])
    tol = 1e-9
    assert np.all(U >= prob["u_l"] - tol) and np.all(U <= prob["u_u"] + tol)
    Xr = _rollout_np(prob, f, fx, fu, U)
    assert np.max(np.abs(Xr - X)) < 1e-8 * max(1.0, np.max(np.abs(X)))
    J0 = _objective(prob, Xr, U)
    rng = np.random.default_rng(M + k)
    for scale in (1e-2, 1e-4):
        for _ in range(3):
            dU = rng.standard_normal(U.shape)
            dU[:, :k] = dU[0:1, :k]  # keep the consensus
            U2 = np.clip(U + scale * dU, prob["u_l"], prob["u_u"])
            assert _objective(prob, _rollout_np(prob, f, fx, fu, U2), U2) >= J0 * (1 - 1e-12)


def test_full_size_thrust_cones(solver):
    """Config E's constraint set at config C's size (quadrotor M=1024, N=50, fp64): control boxes plus the thrust cone
    ||(tau_x, tau_y)|| <= 0.3 T on every stage — consensus, feasibility, exact linearised dynamics and optimality by
    feasible perturbations."""
    import torch

    from pmpc_amd import dynamics as dyn
    from pmpc_amd.device import MODEL_QUADROTOR, to_device_problem

    M, N = 1024, 50
    prob = dyn.make_quadrotor_problem(M=M, N=N)
    d = to_device_problem(prob)
    f, fx, fu = solver.linearize(MODEL_QUADROTOR, d["x0"], d["X_prev"], d["U_prev"], d["params"])
    dev = lambda a: torch.tensor(np.asarray(a, dtype=np.float64), device="cuda")
    W = np.zeros((2, 4)); W[0, 1] = W[1, 2] = 1.0
    X, U, status = solver.lsoc_solve(f=f, fx=fx, fu=fu, X_prev=d["X_prev"], U_prev=d["U_prev"], Q=d["Q"], R=d["R"], X_ref=d["X_ref"],
                                     U_ref=d["U_ref"], reg_x=prob["reg_x"], reg_u=prob["reg_u"], Nc=1, x0=d["x0"], lu=d["lu"], uu=d["uu"],
                                     symmetric_cost=True, soc_W=dev(W), soc_w0=dev(np.zeros(2)), soc_v=dev([0.3, 0, 0, 0]), soc_v0=0.0,
                                     soc_u_interior=dev([9.81, 0, 0, 0]))
    solver.sync()
    assert status == 0
    X, U = X.cpu().numpy(), U.cpu().numpy()
    cone = lambda V: 0.3 * V[..., 0] - np.linalg.norm(V[..., 1:3], axis=-1)
    assert np.all(U[:, 0] == U[0:1, 0]) and cone(U).min() > -1e-9 and (cone(U) < 1e-6).sum() > 100  # active on many stages
    assert np.all(U >= prob["u_l"] - 1e-9) and np.all(U <= prob["u_u"] + 1e-9)
    Xr = _rollout_np(prob, f, fx, fu, U)
    assert np.max(np.abs(Xr - X)) < 1e-8 * max(1.0, np.max(np.abs(X)))
    J0 = _objective(prob, Xr, U)
    rng = np.random.default_rng(3)
    for scale in (1e-2, 1e-4):
        for _ in range(3):
            dU = rng.standard_normal(U.shape)
            dU[:, :1] = dU[0:1, :1]
            U2 = np.clip(U + scale * dU, prob["u_l"], prob["u_u"])
            viol = cone(U2) < 0  # pull violators back onto their cone: shrink the lateral torques
            nrm = np.maximum(np.linalg.norm(U2[..., 1:3], axis=-1), 1e-300)
            shrink = np.where(viol, np.maximum(0.3 * U2[..., 0], 0.0) / nrm, 1.0)
            U2[..., 1:3] *= shrink[..., None]
            U2[:, :1] = U2[0:1, :1]
            if cone(U2).min() < -1e-12:
                continue
            assert _objective(prob, _rollout_np(prob, f, fx, fu, U2), U2) >= J0 * (1 - 1e-12)


def test_scp_residual_kernel_matches_the_reference_formula():
    """pmpc_scp_residual_device vs max(max_ij |dX_ij|_2, max_ij |dU_ij|_2) of pmpc/scp_mpc.py:397-403; NaN -> inf."""
    import torch

    from pmpc_amd.device import DeviceSolver

    s = DeviceSolver(0)
    g = torch.Generator(device="cuda").manual_seed(3)
    for (M, N, x, u) in [(1, 1, 1, 1), (7, 13, 12, 4), (300, 50, 4, 2)]:
        X, Xp = (torch.randn((M, N, x), dtype=torch.float64, device="cuda", generator=g) for _ in range(2))
        U, Up = (torch.randn((M, N, u), dtype=torch.float64, device="cuda", generator=g) for _ in range(2))
        if M > 1:
            U[M // 2, N // 3] += 40.0  # the maximum sits on the control side
        torch.cuda.synchronize()
        r = s.scp_residual(X, Xp, U, Up)
        s.sync()
        ref = torch.maximum(torch.linalg.vector_norm(X - Xp, dim=-1).max(), torch.linalg.vector_norm(U - Up, dim=-1).max())
        assert abs(float(r[0]) - float(ref)) <= 1e-13 * float(ref)
    X[0, 0, 0] = float("nan")
    torch.cuda.synchronize()
    r = s.scp_residual(X, Xp, U, Up)
    s.sync()
    assert float(r[0]) == float("inf")
    s.close()


@pytest.mark.parametrize("generic", [False, True], ids=["fast", "generic"])
@pytest.mark.parametrize("case", [(16, 12, 12, 4, 1), (9, 10, 4, 2, -1), (8, 9, 5, 3, 0)], ids=["quadrotor-dims-Nc1", "NcN", "Nc0"])
def test_warm_active_set_sequence_is_exact(case, generic, oracle):
    """SCP-like sequence of related sub-problems through ONE context: the first solve ends in the active-set finish of the
    interior-point iteration, every later one starts from the previous accepted set and solution (no interior-point
    iteration at all) and must land on the oracle's vertex — tolerance 1e-10, far inside the 1e-7 of the interior-point path.
    A solve with the cold-start flag (interior-point path) agrees."""
    import torch

    from pmpc_amd.device import DeviceSolver
    from tests.support.problems import rand_problem

    M, N, x, u, Nc = case
    rng = np.random.default_rng(77)
    args, kw = rand_problem(rng, M, N, x, u, 0.25)
    s = DeviceSolver(0)
    dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda")
    T = lambda a: dev(np.swapaxes(a, -1, -2))
    rel = lambda a, b: np.linalg.norm(a - b) / max(np.linalg.norm(b), 1.0)
    for t in range(4):
        if t:
            x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
            args = (x0, f + 0.03 * rng.standard_normal(f.shape), fx * (1 + 0.03 * rng.standard_normal(fx.shape)),
                    fu * (1 + 0.03 * rng.standard_normal(fu.shape)), X_prev + 0.03 * rng.standard_normal(X_prev.shape),
                    U_prev + 0.03 * rng.standard_normal(U_prev.shape), Q, R, X_ref, U_ref)
        x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
        Xo, Uo = oracle.lqp_solve_py(*args, Nc=Nc, **kw)
        opt = dict(f=dev(f), fx=T(fx), fu=T(fu), X_prev=dev(X_prev), U_prev=dev(U_prev), Q=T(Q), R=T(R), X_ref=dev(X_ref),
                   U_ref=dev(U_ref), reg_x=kw["reg_x"], reg_u=kw["reg_u"], Nc=Nc, symmetric_cost=True, lu=dev(kw["u_l"]),
                   uu=dev(kw["u_u"]), force_generic=generic)
        X, U, status = s.lqp_solve(**opt)
        s.sync()
        info = dict(s.last_info)
        assert status == 0 and info["fast_path"] == (0 if generic else 1)
        assert info["active_set_rounds"] >= 1, info
        if t:
            assert info["ipm_iters"] == 0, info  # warm start: the active-set rounds alone
        assert rel(X.cpu().numpy(), Xo) < 1e-10 and rel(U.cpu().numpy(), Uo) < 1e-10, (t, info)
        Xc, Uc, status = s.lqp_solve(cold_start=True, **opt)
        s.sync()
        assert status == 0 and rel(Xc.cpu().numpy(), Xo) < 1e-7 and rel(Uc.cpu().numpy(), Uo) < 1e-7
        # (the cold solve's accepted set is as good a start for the next sub-problem as the warm one's)
    s.close()


def test_state_boxes_binding_or_not_stay_in_the_active_set_rounds(oracle):
    """State boxes that do not bind leave the control-box active-set iteration in charge; since r03 a binding one is a row of the
    same rounds (kernels_xbox.hip) instead of sending the solve — and later solves of the shape — to the interior-point iteration."""
    import torch

    from pmpc_amd.device import DeviceSolver
    from tests.support.problems import rand_problem

    M, N, x, u, Nc = 6, 10, 4, 2, 1
    dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda")
    T = lambda a: dev(np.swapaxes(a, -1, -2))
    rel = lambda a, b: np.linalg.norm(a - b) / max(np.linalg.norm(b), 1.0)
    for bx, expect_as in ((1e3, True), (3.5, False)):  # |x| <= 3.5 binds at one state of this problem (max |x| = 3.68 without it)
        args, kw = rand_problem(np.random.default_rng(12), M, N, x, u, 0.3, bx)
        x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
        Xo, Uo = oracle.lqp_solve_py(*args, Nc=Nc, **kw)
        assert (abs(np.abs(Xo).max() - bx) < 1e-9) != expect_as  # (the second box binds)
        s = DeviceSolver(0)
        for rep in range(2):
            X, U, status = s.lqp_solve(f=dev(f), fx=T(fx), fu=T(fu), X_prev=dev(X_prev), U_prev=dev(U_prev), Q=T(Q), R=T(R),
                                       X_ref=dev(X_ref), U_ref=dev(U_ref), reg_x=kw["reg_x"], reg_u=kw["reg_u"], Nc=Nc,
                                       symmetric_cost=True, lu=dev(kw["u_l"]), uu=dev(kw["u_u"]), lx=dev(kw["x_l"]), ux=dev(kw["x_u"]))
            s.sync()
            info = dict(s.last_info)
            assert status == 0 and rel(X.cpu().numpy(), Xo) < 1e-7 and rel(U.cpu().numpy(), Uo) < 1e-7, (bx, rep, info)
            assert info["ipm_iters"] == 0 and info["active_set_rounds"] >= 1, info
            if rep:
                assert info["structured_solves"] <= 2, info  # warm start from the stored set (and multipliers of the state rows)
        s.close()


@pytest.mark.parametrize("case", [(16, 12, 12, 4, 1), (9, 10, 4, 2, 3), (8, 9, 5, 3, 0)], ids=["quadrotor-dims-Nc1", "Nc3", "Nc0"])
def test_scp_like_sequence_without_rollout(case, oracle):
    """PMPC_PREV_IS_LAST_SOLUTION: each sub-problem's X_prev / U_prev are the previous solve's outputs (an SCP loop), so the
    warm start takes the linearisation point as its base and carries the dynamics defect f - X_prev through the sweeps
    instead of a rollout.  Exact against the oracle; a broken promise (X_prev / U_prev perturbed) is detected on the device
    and still ends on the optimum."""
    import torch

    from pmpc_amd.device import DeviceSolver
    from tests.support.problems import rand_problem

    M, N, x, u, Nc = case
    rng = np.random.default_rng(101)
    args, kw = rand_problem(rng, M, N, x, u, 0.25)
    s = DeviceSolver(0)
    dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda")
    T = lambda a: dev(np.swapaxes(a, -1, -2))
    rel = lambda a, b: np.linalg.norm(a - b) / max(np.linalg.norm(b), 1.0)
    Xl = Ul = None
    for t in range(5):
        x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
        promise = t > 0
        if t:  # next SCP iteration: re-linearised dynamics around the last solution
            f = f + 0.03 * rng.standard_normal(f.shape)
            fx = fx * (1 + 0.03 * rng.standard_normal(fx.shape))
            fu = fu * (1 + 0.03 * rng.standard_normal(fu.shape))
            X_prev, U_prev = Xl, Ul
            if t == 3:  # a caller that breaks the promise (keeps the flag, feeds something else back)
                U_prev = Ul + 1e-3 * rng.standard_normal(Ul.shape)
        args = (x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref)
        Xo, Uo = oracle.lqp_solve_py(*args, Nc=Nc, **kw)
        X, U, status = s.lqp_solve(f=dev(f), fx=T(fx), fu=T(fu), X_prev=dev(X_prev), U_prev=dev(U_prev), Q=T(Q), R=T(R),
                                   X_ref=dev(X_ref), U_ref=dev(U_ref), reg_x=kw["reg_x"], reg_u=kw["reg_u"], Nc=Nc, symmetric_cost=True,
                                   lu=dev(kw["u_l"]), uu=dev(kw["u_u"]), prev_is_last_solution=promise)
        s.sync()
        info = dict(s.last_info)
        Xl, Ul = X.cpu().numpy(), U.cpu().numpy()
        assert status == 0 and rel(Xl, Xo) < 1e-9 and rel(Ul, Uo) < 1e-9, (t, info, rel(Xl, Xo), rel(Ul, Uo))
        if t in (1, 2, 4):
            assert info["ipm_iters"] == 0 and info["active_set_rounds"] >= 1, (t, info)
            # warm: the rounds alone, no equality-only solve in front of them (also with several consensus stages, where the
            # base point is the caller's U_prev by rollout instead of the no-rollout start)
            if Nc > 1 and t in (1, 2):
                assert info["structured_solves"] == info["active_set_rounds"], (t, info)
    s.close()


@pytest.mark.parametrize("model,M,N,Nc", [("quadrotor", 96, 20, 1), ("unicycle", 40, 12, 3)])
def test_library_scp_loop_equals_the_python_driven_loop(solver, model, M, N, Nc):
    """pmpc_scp_loop_device (linearise -> sub-problem -> residual -> swap inside the library, follow-up work enqueued behind the
    rounds before their outcome is known) walks exactly the sequence a Python-driven loop of the same calls walks: identical
    iterates and residuals after 1..5 iterations, cold first iteration included."""
    import torch

    from pmpc_amd import dynamics as dyn
    from pmpc_amd.device import MODEL_QUADROTOR, MODEL_UNICYCLE, to_device_problem

    prob = dyn.make_quadrotor_problem(M=M, N=N, Nc=Nc) if model == "quadrotor" else dyn.make_unicycle_problem(M=M, N=N, Nc=Nc)
    mid = MODEL_QUADROTOR if model == "quadrotor" else MODEL_UNICYCLE
    d = to_device_problem(prob)
    common = dict(Q=d["Q"], R=d["R"], X_ref=d["X_ref"], U_ref=d["U_ref"], reg_x=prob["reg_x"], reg_u=prob["reg_u"], Nc=Nc, x0=d["x0"], lu=d["lu"],
                  uu=d["uu"], symmetric_cost=True)
    for steps in (1, 2, 5):
        # Python-driven
        Xa, Ua = d["X_prev"].clone(), d["U_prev"].clone()
        Xb, Ub = torch.empty_like(Xa), torch.empty_like(Ua)
        res_py = []
        for it in range(steps):
            f, fx, fu = solver.linearize(mid, d["x0"], Xa, Ua, d["params"])
            _, _, st = solver.lqp_solve(f=f, fx=fx, fu=fu, X_prev=Xa, U_prev=Ua, X_out=Xb, U_out=Ub, static_cons_bounds=True,
                                        prev_is_last_solution=it > 0, cold_start=it == 0, **common)
            assert st == 0
            res_py.append(float(solver.scp_residual(Xb, Xa, Ub, Ua)[0].item()))
            Xa, Xb, Ua, Ub = Xb, Xa, Ub, Ua
        X_py, U_py = Xa.clone(), Ua.clone()
        # inside the library (a fresh shape memory: cold like the loop above)
        Xa, Ua = d["X_prev"].clone(), d["U_prev"].clone()
        Xb, Ub = torch.empty_like(Xa), torch.empty_like(Ua)
        x, u = Xa.shape[-1], Ua.shape[-1]
        mk = lambda *shape: torch.empty(shape, dtype=torch.float64, device="cuda")
        bufs = [mk(M, N, x), mk(M, N, x, x), mk(M, N, u, x), mk(M, N, x), mk(M, N, x, x), mk(M, N, u, x)]
        solver.lqp_solve(f=bufs[0].zero_(), fx=bufs[1].zero_(), fu=bufs[2].zero_(), X_prev=Xa, U_prev=Ua, X_out=Xb, U_out=Ub, cold_start=True,
                         **dict(common, lu=None, uu=None))  # (forget the warm-start memory of this shape)
        res, infos, last_in_out, done = solver.scp_loop(mid, d["params"], steps, f=bufs[0], fx=bufs[1], fu=bufs[2], f2=bufs[3], fx2=bufs[4],
                                                        fu2=bufs[5], X_prev=Xa, U_prev=Ua, X_out=Xb, U_out=Ub, first_cold=True, **common)
        solver.sync()
        assert done == steps and all(i["status"] == 0 for i in infos)
        X_lib, U_lib = (Xb, Ub) if last_in_out else (Xa, Ua)
        assert last_in_out == bool(steps & 1)
        np.testing.assert_array_equal(res.cpu().numpy(), np.array(res_py))
        assert torch.equal(X_lib, X_py) and torch.equal(U_lib, U_py)


def test_bench_line_keeps_the_driver_contract():
    """`python bench.py --gpus 1 --steps K --warmup W` prints ONE JSON line with the keys the driver reads, the metric / unit of
    BASELINE.json, and the `roofline` and `cpu_baseline` objects (small sizes here: the contract, not the numbers)."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "3", "--warmup", "1", "--M", "64", "--N", "12"],
                       cwd=str(root), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["unit"] == "SCP iterations/s" and d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["per_solve"]["frac"] > 0 and set(rf["kernel_ms_per_step"]) >= {"bwd_factor", "fwd", "linearize", "scp_residual"}
    rp = d["repeats"]
    assert rp["windows"] == 5 and len(rp["values"]) == 5 and rp["min"] <= rp["median"] <= rp["max"] and abs(rp["values"][0] - d["value"]) < 1e-9 * d["value"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and "sample" in cb and "julia" in cb


def test_a_context_says_once_when_it_leaves_the_register_resident_path(capfd):
    """a caller who forgets symmetric_cost=True gets a several times slower solver — said once per context on
    stderr (and in info["fast_path"]), silenced by the option warn_slow_path."""
    import torch

    from pmpc_amd.device import DeviceSolver
    from tests.support.problems import rand_problem

    M, N, x, u = 4, 6, 4, 2
    args, kw = rand_problem(np.random.default_rng(3), M, N, x, u, 0.4)
    dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda")
    T = lambda a: dev(np.swapaxes(a, -1, -2))
    x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
    opt = dict(f=dev(f), fx=T(fx), fu=T(fu), X_prev=dev(X_prev), U_prev=dev(U_prev), Q=T(Q), R=T(R), X_ref=dev(X_ref), U_ref=dev(U_ref),
               reg_x=kw["reg_x"], reg_u=kw["reg_u"], Nc=1, lu=dev(kw["u_l"]), uu=dev(kw["u_u"]))
    s = DeviceSolver(0)
    capfd.readouterr()
    for rep in range(2):
        _, _, status = s.lqp_solve(**opt)  # symmetric_cost not declared
        s.sync()
        assert status == 0 and s.last_info["fast_path"] == 0
    err = capfd.readouterr().err
    assert err.count("runs on the generic kernels") == 1 and "PMPC_SYMMETRIC_COST" in err, err
    _, _, status = s.lqp_solve(symmetric_cost=True, **opt)
    s.sync()
    assert status == 0 and s.last_info["fast_path"] == 1
    s.close()
    quiet = DeviceSolver(0)
    quiet.set_option("warn_slow_path", 0)
    quiet.lqp_solve(**opt)
    quiet.sync()
    assert "generic kernels" not in capfd.readouterr().err
    quiet.close()


def test_every_documented_option_exists_with_its_documented_default():
    """include/pmpc_abi.h lists the per-context options (key, environment variable, default): each is known to pmpc_get_option and
    starts at that default (the variables are not set in the test environment), and can be set and read back."""
    import os
    import re
    from pathlib import Path

    from pmpc_amd.device import DeviceSolver

    header = (Path(__file__).resolve().parents[1] / "include" / "pmpc_abi.h").read_text()
    rows = re.findall(r"^ \*   (\w+) (PMPC_\w+) ([0-9.e+-]+)", header, flags=re.M)
    rows += re.findall(r", (\w+) (PMPC_\w+) ([0-9.e+-]+)", header)
    keys = {k: (env, float(d)) for k, env, d in rows}
    assert {"as_warm", "xbox_as", "cone_as", "polish_mu", "as_cold_rounds", "cone_rank_memory", "as_wave_cons", "warn_slow_path"} <= set(keys), keys
    s = DeviceSolver(0)
    for k, (env, dflt) in keys.items():
        if env in os.environ:
            continue
        assert s.get_option(k) == dflt, (k, s.get_option(k), dflt)
        s.set_option(k, dflt + 1.0)
        assert s.get_option(k) == dflt + 1.0
    s.close()


def test_graft_entry_smoke_passes():
    """The driver's smoke entry (one small invocation of every part of the hot path against the oracle) as a test: a change that breaks
    it must not get past `pytest -m gpu` (r04: a line-search heuristic of the smoothed cone path did, on exactly its case)."""
    import __graft_entry__ as entry

    entry.smoke()
