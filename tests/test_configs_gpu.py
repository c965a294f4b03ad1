"""BASELINE.json configs B, C, D and E's shape on the GPU against an INDEPENDENT CPU solver on the same data — at full size:

  B  unicycle M=256 N=30 (per-particle parameters and x0), Nc = 1 and Nc = N: every sub-problem of bench.py's SCP loop
     (cold first solve, then the warm-started / no-rollout calling pattern) vs the exact sparse oracle `lqp_solve_py`;
  C, D  quadrotor M=1024 / 4096, N=50: the same loop vs oracle/structured_cpu.c (Riccati + Mehrotra in C, itself pinned to
     the exact oracle in tests/test_oracle_golden.py) — cold solve and the warm DEFECT sequence;
  E-shape  N=100 with control boxes + thrust cone vs the cone oracle `lsoc_solve_py` (M=8), and the M=4096, N=100 problem
     through size-independent properties;
  c_lcone_solve at M=600 (> 500 particles: threshold rank m* = 2) vs the cone oracle, with the semantics stated there.
Tolerance: 1e-7 relative on trajectories (north-star bar: 1e-6)."""
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


def _rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1.0)


def _scp_loop_vs_cpu(solver, prob, model, Nc, steps, cpu_solve, tol=1e-7, kkt=False):
    """bench.py's calling pattern: on-device linearisation, device solve with the SCP-loop promises (same boxes,
    X_prev / U_prev = previous outputs), ping-pong buffers.  Every sub-problem is also solved on the CPU from the SAME
    linearisation (downloaded) and compared."""
    import torch

    from pmpc_amd.device import to_device_problem

    d = to_device_problem(prob)
    Xa, Ua = d["X_prev"].clone(), d["U_prev"].clone()
    Xb, Ub = torch.empty_like(Xa), torch.empty_like(Ua)
    worst, rounds = 0.0, []
    for it in range(steps):
        f, fx, fu = solver.linearize(model, d["x0"], Xa, Ua, d["params"])
        _, _, status = solver.lqp_solve(f=f, fx=fx, fu=fu, X_prev=Xa, U_prev=Ua, Q=d["Q"], R=d["R"], X_ref=d["X_ref"], U_ref=d["U_ref"],
                                        reg_x=prob["reg_x"], reg_u=prob["reg_u"], Nc=Nc, x0=d["x0"], lu=d["lu"], uu=d["uu"], X_out=Xb, U_out=Ub,
                                        symmetric_cost=True, static_cons_bounds=True, prev_is_last_solution=it > 0)
        solver.sync()
        assert status == 0, (it, solver.last_info)
        assert solver.last_info["fast_path"] == 1
        rounds.append(solver.last_info["active_set_rounds"])
        Xc, Uc = cpu_solve(f.cpu().numpy(), fx.cpu().numpy().swapaxes(-1, -2), fu.cpu().numpy().swapaxes(-1, -2), Xa.cpu().numpy(), Ua.cpu().numpy())
        ex, eu = _rel(Xb.cpu().numpy(), Xc), _rel(Ub.cpu().numpy(), Uc)
        print(f"  SCP iteration {it + 1}: rel err X {ex:.1e} U {eu:.1e}, rounds {rounds[-1]}", flush=True)
        worst = max(worst, ex, eu)
        assert ex < tol and eu < tol, (it, ex, eu, solver.last_info)
        if kkt:  # evidence that does not pass through either solver's algorithm: the joint QP's KKT conditions at the returned point
            from tests.support.kkt_certificate import kkt_certificate

            cert = kkt_certificate(prob["x0"], f.cpu().numpy(), fx.cpu().numpy().swapaxes(-1, -2), fu.cpu().numpy().swapaxes(-1, -2), Xa.cpu().numpy(),
                                   Ua.cpu().numpy(), prob["Q"], prob["R"], prob["X_ref"], prob["U_ref"], prob["reg_x"], prob["reg_u"], Nc, prob["u_l"],
                                   prob["u_u"], Xb.cpu().numpy(), Ub.cpu().numpy())
            print(f"    KKT certificate: {cert}", flush=True)
            assert cert["active_bounds"] > 0
            assert max(cert["dynamics"], cert["consensus"], cert["box"], cert["stationarity"], cert["stationarity_shared"]) < 1e-9, cert
        Xa, Xb, Ua, Ub = Xb, Xa, Ub, Ua
    return worst, rounds


@pytest.mark.parametrize("M,Nc,cpu_kind", [(256, 1, "exact"), (256, -1, "structured"), (24, -1, "exact")])
def test_config_B_unicycle_scp_loop_matches_cpu(solver, oracle, M, Nc, cpu_kind):
    """BASELINE configs[1] as SURVEY.md section 8(d) specifies it: unicycle M=256, N=30, p_i = [1+0.1 xi, 1+0.1 xi, 0.3],
    x0_i = 1 + 0.05 N(0, I), |u| <= 1, 5 SCP iterations.  Nc = 1 (primary) against the exact sparse oracle; Nc = N (the
    reference's default) at full size against the structured C solver (the exact oracle needs 4 minutes per solve there: its
    KKT matrix has a dense 60-column border) and at M = 24 against the exact oracle."""
    import os

    from pmpc_amd import dynamics as dyn
    from pmpc_amd.device import MODEL_UNICYCLE

    prob = dyn.make_unicycle_problem(M=M, N=30, Nc=Nc)
    threads = min(16, len(os.sched_getaffinity(0)))

    def cpu(f, fx, fu, Xp, Up):
        if cpu_kind == "exact":
            return oracle.lqp_solve_py(prob["x0"], f, fx, fu, Xp, Up, prob["Q"], prob["R"], prob["X_ref"], prob["U_ref"], reg_x=prob["reg_x"],
                                       reg_u=prob["reg_u"], Nc=Nc, u_l=prob["u_l"], u_u=prob["u_u"])
        X, U, info = oracle.structured_cpu_solve_py(prob["x0"], f, fx, fu, Xp, Up, prob["Q"], prob["R"], prob["X_ref"], prob["U_ref"],
                                                    prob["reg_x"], prob["reg_u"], Nc=Nc, u_l=prob["u_l"], u_u=prob["u_u"], threads=threads)
        assert info["status"] == 0, info
        return X, U

    worst, rounds = _scp_loop_vs_cpu(solver, prob, MODEL_UNICYCLE, Nc, 5, cpu, kkt=True)
    assert rounds[-1] >= 1  # the later sub-problems went through the warm-started active-set rounds
    print(f"config B M={M} Nc={Nc}: worst rel err {worst:.2e}, active-set rounds {rounds}")


@pytest.mark.parametrize("M", [1024, 4096])
def test_configs_C_D_full_size_scp_loop_matches_structured_cpu(solver, oracle, M):
    """BASELINE configs[2] (M=1024) and configs[3] (M=4096) at FULL size, N=50, Nc=1, control boxes: the cold first solve and
    three warm-started sub-problems (the DEFECT / no-rollout sequence bench.py times) against the structured C solver."""
    import os

    from pmpc_amd import dynamics as dyn
    from pmpc_amd.device import MODEL_QUADROTOR

    prob = dyn.make_quadrotor_problem(M=M, N=50, Nc=1)
    threads = min(16, len(os.sched_getaffinity(0)))

    def cpu(f, fx, fu, Xp, Up):
        X, U, info = oracle.structured_cpu_solve_py(prob["x0"], f, fx, fu, Xp, Up, prob["Q"], prob["R"], prob["X_ref"], prob["U_ref"],
                                                    prob["reg_x"], prob["reg_u"], Nc=1, u_l=prob["u_l"], u_u=prob["u_u"], threads=threads)
        assert info["status"] == 0, info
        return X, U

    worst, rounds = _scp_loop_vs_cpu(solver, prob, MODEL_QUADROTOR, 1, 4, cpu, kkt=True)
    print(f"config {'C' if M == 1024 else 'D'} M={M}: worst rel err {worst:.2e}, active-set rounds {rounds}")


def _thrust_cone(dev):
    W = np.zeros((2, 4))
    W[0, 1] = W[1, 2] = 1.0
    return dict(soc_W=dev(W), soc_w0=dev(np.zeros(2)), soc_v=dev([0.3, 0, 0, 0]), soc_v0=0.0, soc_u_interior=dev([9.81, 0, 0, 0])), W


def test_config_E_shape_N100_cones_match_cone_oracle(solver, oracle):
    """Config E's horizon and constraint set (N=100, control boxes + thrust cone ||(tau_x, tau_y)|| <= 0.3 T on every stage),
    fp64, on a particle count the cone oracle (sparse log-barrier path following) finishes in seconds."""
    import torch

    from pmpc_amd import dynamics as dyn
    from pmpc_amd.device import MODEL_QUADROTOR, to_device_problem

    M, N = 8, 100
    prob = dyn.make_quadrotor_problem(M=M, N=N)
    d = to_device_problem(prob)
    f, fx, fu = solver.linearize(MODEL_QUADROTOR, d["x0"], d["X_prev"], d["U_prev"], d["params"])
    dev = lambda a: torch.tensor(np.asarray(a, dtype=np.float64), device="cuda")
    soc, W = _thrust_cone(dev)
    X, U, status = solver.lsoc_solve(f=f, fx=fx, fu=fu, X_prev=d["X_prev"], U_prev=d["U_prev"], Q=d["Q"], R=d["R"], X_ref=d["X_ref"],
                                     U_ref=d["U_ref"], reg_x=prob["reg_x"], reg_u=prob["reg_u"], Nc=1, x0=d["x0"], lu=d["lu"], uu=d["uu"],
                                     symmetric_cost=True, **soc)
    solver.sync()
    assert status == 0
    fn, fxn, fun = f.cpu().numpy(), fx.cpu().numpy().swapaxes(-1, -2), fu.cpu().numpy().swapaxes(-1, -2)
    Xo, Uo = oracle.lsoc_solve_py(prob["x0"], fn, fxn, fun, prob["X_prev"], prob["U_prev"], prob["Q"], prob["R"], prob["X_ref"], prob["U_ref"],
                                  reg_x=prob["reg_x"], reg_u=prob["reg_u"], Nc=1, u_l=prob["u_l"], u_u=prob["u_u"], soc_W=W, soc_w0=np.zeros(2),
                                  soc_v=np.array([0.3, 0, 0, 0]), soc_v0=0.0, u_interior=np.array([9.81, 0, 0, 0]))
    ex, eu = _rel(X.cpu().numpy(), Xo), _rel(U.cpu().numpy(), Uo)
    assert ex < 1e-6 and eu < 1e-6, (ex, eu)  # (the cone oracle itself stops at mu = 1e-13: ~1e-8 on trajectories)


@pytest.mark.parametrize("M", [8])  # (the cone oracle on the CPU is the slow part: 35 s at M = 8, 39 s at M = 16 — which passes too —, 123 s at M = 32)
def test_config_E_as_stated_fp32_storage_matches_the_fp64_cone_oracle(solver, oracle, M):
    """BASELINE configs[4] AS STATED: quadrotor, N = 100, control boxes + thrust cones, **fp32** — here: fx, fu, Q, R and the factor
    records stored in float32 (PMPC_F32_MATRICES; half the HBM bytes of the dominant arrays), every value widened on load, all
    arithmetic fp64.  The reference is fp64-only (PMPC.jl/src/c_interface.jl:6-25), so the comparator is the fp64 cone oracle on
    the fp64 data.  STATED TOLERANCE: 1e-6 relative on X and U (north star's bar; measured ~1e-8: the rounding of the data to
    fp32, relative 6e-8, times the problem's conditioning).  Three SCP iterations: the first is a cold start (widened copies, fp64
    kernels), the second and third run the warm-started active-set rounds on the float arrays (pmpc_info.fast_path == 2)."""
    import torch

    from pmpc_amd import dynamics as dyn
    from pmpc_amd.device import MODEL_QUADROTOR, to_device_problem

    N = 100
    prob = dyn.make_quadrotor_problem(M=M, N=N)
    d = to_device_problem(prob)
    dev = lambda a: torch.tensor(np.asarray(a, dtype=np.float64), device="cuda")
    soc, W = _thrust_cone(dev)
    Q32, R32 = d["Q"].to(torch.float32).contiguous(), d["R"].to(torch.float32).contiguous()
    fx = torch.empty((M, N, 12, 12), dtype=torch.float32, device="cuda")
    fu = torch.empty((M, N, 4, 12), dtype=torch.float32, device="cuda")
    f = torch.empty((M, N, 12), dtype=torch.float64, device="cuda")
    Xa, Ua = d["X_prev"].clone(), d["U_prev"].clone()
    Xb, Ub = torch.empty_like(Xa), torch.empty_like(Ua)
    for it in range(3):
        solver.linearize(MODEL_QUADROTOR, d["x0"], Xa, Ua, d["params"], f, fx, fu)
        X, U, status = solver.lsoc_solve(f=f, fx=fx, fu=fu, X_prev=Xa, U_prev=Ua, Q=Q32, R=R32, X_ref=d["X_ref"], U_ref=d["U_ref"],
                                         reg_x=prob["reg_x"], reg_u=prob["reg_u"], Nc=1, x0=d["x0"], lu=d["lu"], uu=d["uu"], X_out=Xb, U_out=Ub,
                                         symmetric_cost=True, static_cons_bounds=True, prev_is_last_solution=it > 0, **soc)
        solver.sync()
        assert status == 0
        if it > 0:
            assert solver.last_info["fast_path"] == 2 and solver.last_info["ipm_iters"] == 0 and solver.last_info["active_set_rounds"] > 0
        if it == 2:  # this sub-problem in fp64 on the CPU: linearisation of the reference-style numpy dynamics at the same point
            Xp, Up = Xa.cpu().numpy(), Ua.cpu().numpy()
            fn, fxn, fun = prob["f_fx_fu_fn"](np.concatenate([prob["x0"][:, None, :], Xp[:, :-1]], 1), Up)
            Xo, Uo = oracle.lsoc_solve_py(prob["x0"], fn, fxn, fun, Xp, Up, prob["Q"], prob["R"], prob["X_ref"], prob["U_ref"], reg_x=prob["reg_x"],
                                          reg_u=prob["reg_u"], Nc=1, u_l=prob["u_l"], u_u=prob["u_u"], soc_W=W, soc_w0=np.zeros(2),
                                          soc_v=np.array([0.3, 0, 0, 0]), soc_v0=0.0, u_interior=np.array([9.81, 0, 0, 0]))
            ex, eu = _rel(Xb.cpu().numpy(), Xo), _rel(Ub.cpu().numpy(), Uo)
            assert ex < 1e-6 and eu < 1e-6, (ex, eu)
        Xa, Xb, Ua, Ub = Xb, Xa, Ub, Ua


def test_config_E_size_M4096_N100_thrust_cones_properties(solver):
    """BASELINE configs[4]'s size (quadrotor M=4096, N=100, boxes + thrust cones; fp64 — the reference has no fp32): consensus,
    box and cone feasibility with the cone active on many stages, exact linearised dynamics, optimality by feasible perturbations."""
    import torch

    from pmpc_amd import dynamics as dyn
    from pmpc_amd.device import MODEL_QUADROTOR, to_device_problem
    from tests.test_device_gpu import _objective, _rollout_np

    M, N = 4096, 100
    prob = dyn.make_quadrotor_problem(M=M, N=N)
    d = to_device_problem(prob)
    f, fx, fu = solver.linearize(MODEL_QUADROTOR, d["x0"], d["X_prev"], d["U_prev"], d["params"])
    dev = lambda a: torch.tensor(np.asarray(a, dtype=np.float64), device="cuda")
    soc, _ = _thrust_cone(dev)
    X, U, status = solver.lsoc_solve(f=f, fx=fx, fu=fu, X_prev=d["X_prev"], U_prev=d["U_prev"], Q=d["Q"], R=d["R"], X_ref=d["X_ref"],
                                     U_ref=d["U_ref"], reg_x=prob["reg_x"], reg_u=prob["reg_u"], Nc=1, x0=d["x0"], lu=d["lu"], uu=d["uu"],
                                     symmetric_cost=True, **soc)
    solver.sync()
    assert status == 0
    X, U = X.cpu().numpy(), U.cpu().numpy()
    cone = lambda V: 0.3 * V[..., 0] - np.linalg.norm(V[..., 1:3], axis=-1)
    assert np.all(U[:, 0] == U[0:1, 0]) and cone(U).min() > -1e-9 and (cone(U) < 1e-6).sum() > 400
    assert np.all(U >= prob["u_l"] - 1e-9) and np.all(U <= prob["u_u"] + 1e-9)
    Xr = _rollout_np(prob, f, fx, fu, U)
    assert np.max(np.abs(Xr - X)) < 1e-8 * max(1.0, np.max(np.abs(X)))
    # the KKT conditions of the joint cone program at the returned point, from the ABI data alone (tests/support/kkt_certificate.py): no
    # solver's algorithm in between — costates by the adjoint recursion, box multipliers by their signs, one fitted multiplier per active cone
    from tests.support.kkt_certificate import kkt_certificate

    cert = kkt_certificate(prob["x0"], f.cpu().numpy(), fx.cpu().numpy().swapaxes(-1, -2), fu.cpu().numpy().swapaxes(-1, -2), prob["X_prev"], prob["U_prev"],
                           prob["Q"], prob["R"], prob["X_ref"], prob["U_ref"], prob["reg_x"], prob["reg_u"], 1, prob["u_l"], prob["u_u"], X, U,
                           soc=dict(W=np.array([[0, 1.0, 0, 0], [0, 0, 1.0, 0]]), w0=np.zeros(2), v=np.array([0.3, 0, 0, 0]), v0=0.0), tol_act=1e-8)
    print(f"config E size: KKT certificate {cert}", flush=True)
    assert cert["cone_active"] > 400
    # (the cones' Newton iteration stops at steps of 1e-6 relative, kernels_cone.hip tol_step: its stationarity residual is what that leaves)
    assert max(cert["dynamics"], cert["consensus"], cert["box"], cert["cone"]) < 1e-9 and max(cert["stationarity"], cert["stationarity_shared"]) < 1e-8, cert
    J0 = _objective(prob, Xr, U)
    rng = np.random.default_rng(11)
    for scale in (1e-2, 1e-4):
        dU = rng.standard_normal(U.shape)
        dU[:, :1] = dU[0:1, :1]
        U2 = np.clip(U + scale * dU, prob["u_l"], prob["u_u"])
        nrm = np.maximum(np.linalg.norm(U2[..., 1:3], axis=-1), 1e-300)
        U2[..., 1:3] *= np.where(cone(U2) < 0, np.maximum(0.3 * U2[..., 0], 0.0) / nrm, 1.0)[..., None]
        U2[:, :1] = U2[0:1, :1]
        if cone(U2).min() >= -1e-12:
            assert _objective(prob, _rollout_np(prob, f, fx, fu, U2), U2) >= J0 * (1 - 1e-12)


def test_config_E_full_size_fp32_storage_properties(solver):
    """BASELINE configs[4] AS STATED at FULL size — quadrotor M = 4096, N = 100, boxes + thrust cones, fp32 STORAGE (fx, fu, Q, R and the
    factor records float32, arithmetic fp64) — by properties of the warm-started second SCP iteration (pmpc_info.fast_path == 2): consensus,
    box and cone feasibility, the linearised dynamics of the float-stored Jacobians reproduced to round-off, and optimality of the
    float-stored problem by feasible perturbations.  (The small-M comparison with the fp64 cone oracle is the test above.)"""
    import torch

    from pmpc_amd import dynamics as dyn
    from pmpc_amd.device import MODEL_QUADROTOR, to_device_problem
    from tests.test_device_gpu import _objective, _rollout_np

    M, N = 4096, 100
    prob = dyn.make_quadrotor_problem(M=M, N=N)
    d = to_device_problem(prob)
    dev = lambda a: torch.tensor(np.asarray(a, dtype=np.float64), device="cuda")
    soc, _ = _thrust_cone(dev)
    Q32, R32 = d["Q"].to(torch.float32).contiguous(), d["R"].to(torch.float32).contiguous()
    fx = torch.empty((M, N, 12, 12), dtype=torch.float32, device="cuda")
    fu = torch.empty((M, N, 4, 12), dtype=torch.float32, device="cuda")
    f = torch.empty((M, N, 12), dtype=torch.float64, device="cuda")
    Xa, Ua = d["X_prev"].clone(), d["U_prev"].clone()
    Xb, Ub = torch.empty_like(Xa), torch.empty_like(Ua)
    for it in range(2):
        solver.linearize(MODEL_QUADROTOR, d["x0"], Xa, Ua, d["params"], f, fx, fu)
        _, _, status = solver.lsoc_solve(f=f, fx=fx, fu=fu, X_prev=Xa, U_prev=Ua, Q=Q32, R=R32, X_ref=d["X_ref"], U_ref=d["U_ref"], reg_x=prob["reg_x"],
                                         reg_u=prob["reg_u"], Nc=1, x0=d["x0"], lu=d["lu"], uu=d["uu"], X_out=Xb, U_out=Ub, symmetric_cost=True,
                                         static_cons_bounds=True, prev_is_last_solution=it > 0, **soc)
        solver.sync()
        assert status == 0, solver.last_info
        if it == 0:
            Xa, Xb, Ua, Ub = Xb, Xa, Ub, Ua
    assert solver.last_info["fast_path"] == 2 and solver.last_info["ipm_iters"] == 0 and solver.last_info["active_set_rounds"] > 0, solver.last_info
    X, U = Xb.cpu().numpy(), Ub.cpu().numpy()
    # the problem the float-stored solve solved: fp32-rounded cost blocks and Jacobians, linearised at (Xa, Ua)
    p32 = dict(prob, Q=Q32.double().cpu().numpy().swapaxes(-1, -2), R=R32.double().cpu().numpy().swapaxes(-1, -2), X_prev=Xa.cpu().numpy(), U_prev=Ua.cpu().numpy())
    cone = lambda V: 0.3 * V[..., 0] - np.linalg.norm(V[..., 1:3], axis=-1)
    assert np.all(U[:, 0] == U[0:1, 0]) and cone(U).min() > -1e-9
    assert np.all(U >= prob["u_l"] - 1e-9) and np.all(U <= prob["u_u"] + 1e-9)
    Xr = _rollout_np(p32, f, fx.double(), fu.double(), U)
    assert np.max(np.abs(Xr - X)) < 1e-8 * max(1.0, np.max(np.abs(X)))
    J0 = _objective(p32, Xr, U)
    rng = np.random.default_rng(12)
    for scale in (1e-2, 1e-4):
        dU = rng.standard_normal(U.shape)
        dU[:, :1] = dU[0:1, :1]
        U2 = np.clip(U + scale * dU, prob["u_l"], prob["u_u"])
        nrm = np.maximum(np.linalg.norm(U2[..., 1:3], axis=-1), 1e-300)
        U2[..., 1:3] *= np.where(cone(U2) < 0, np.maximum(0.3 * U2[..., 0], 0.0) / nrm, 1.0)[..., None]
        U2[:, :1] = U2[0:1, :1]
        if cone(U2).min() >= -1e-12:
            assert _objective(p32, _rollout_np(p32, f, fx.double(), fu.double(), U2), U2) >= J0 * (1 - 1e-12)


@pytest.mark.parametrize("Nc", [1, -1])
def test_c_lcone_solve_600_particles_matches_cone_oracle(oracle, Nc):
    """`c_lcone_solve` beyond 500 particles (the reference's default solver "ecos" at any of the BASELINE particle counts):
    threshold rank m* = 2, so one particle carries no weight in the reference's objective and its free variables are not
    unique there.  Expected semantics (oracle `_lcone_many_particles`): that particle minimises its own cost given the shared
    controls.  The device solver keeps a floor weight 1e-4 on it, which moves the shared controls by O(1e-4 / M)."""
    from pmpc_amd import backend
    from tests.support.problems import abi_args, rand_problem

    rng = np.random.default_rng(5)
    args, kw = rand_problem(rng, 600, 5, 3, 2, 0.4)
    Xo, Uo, info = oracle.lcone_solve_py(*args, Nc=Nc, return_info=True, **kw)
    assert np.sum(info["weights"] < 0.5) == 1  # one zero-weight particle (m* - 1)
    X, U = backend.lcone_solve(*abi_args(args, kw, Nc), smooth_alpha=float("nan"), solver="ecos")
    assert np.all(np.isfinite(X)) and np.all(np.isfinite(U))
    ex, eu = _rel(X, Xo), _rel(U, Uo)
    assert ex < 1e-6 and eu < 1e-6, (ex, eu)


def test_consensus_horizon_beyond_N_is_refused_like_the_reference():
    """Nc > N: the reference indexes out of bounds (lqp_utils.jl:17-61), the oracle raises, the device solver fails the solve
    (NaN outputs -> (None, None, None) in the host loop) instead of clamping silently."""
    from oracle import lqp_oracle
    from pmpc_amd import backend
    from tests.support.problems import abi_args, rand_problem

    args, kw = rand_problem(np.random.default_rng(1), 3, 5, 3, 2, 0.4)
    X, U = backend.lqp_solve(*abi_args(args, kw, 6))
    assert np.all(np.isnan(X)) and np.all(np.isnan(U))
    with pytest.raises(ValueError):
        lqp_oracle.lqp_solve_py(*args, Nc=6, **kw)


def test_config_D_size_smoothed_cone_objective_certificate(solver):
    """The reference's DEFAULT solver path with log-barrier smoothing (c_lcone_solve, smooth_alpha = 10) at config D's size — r04 had a
    bench line for it and no parity evidence beyond M ~ 20.  No oracle finishes 4096 x 50 x 16 variables with 4096 cones in test time;
    instead the returned point's optimality is certified from the ABI data alone (tests/support/kkt_certificate.smoothed_cone_certificate):
    the epigraph multipliers are determined by each particle's own stationarity, then their range, sum, sides of the threshold cost and
    the shared controls' stationarity are checked.  Cold solve + one warm solve of the SCP loop."""
    import torch

    from pmpc_amd import dynamics as dyn
    from pmpc_amd.device import MODEL_QUADROTOR, to_device_problem
    from tests.support.kkt_certificate import smoothed_cone_certificate

    M, N, alpha = 4096, 50, 10.0
    prob = dyn.make_quadrotor_problem(M=M, N=N)
    d = to_device_problem(prob)
    Xa, Ua = d["X_prev"].clone(), d["U_prev"].clone()
    Xb, Ub = torch.empty_like(Xa), torch.empty_like(Ua)
    for it in range(2):
        f, fx, fu = solver.linearize(MODEL_QUADROTOR, d["x0"], Xa, Ua, d["params"])
        _, _, status = solver.lcone_solve(smooth_alpha=alpha, f=f, fx=fx, fu=fu, X_prev=Xa, U_prev=Ua, Q=d["Q"], R=d["R"], X_ref=d["X_ref"], U_ref=d["U_ref"],
                                          reg_x=prob["reg_x"], reg_u=prob["reg_u"], Nc=1, x0=d["x0"], lu=d["lu"], uu=d["uu"], X_out=Xb, U_out=Ub, symmetric_cost=True)
        solver.sync()
        assert status == 0, solver.last_info
        n = lambda t_: t_.cpu().numpy()
        cert = smoothed_cone_certificate(prob["x0"], n(f), n(fx).swapaxes(-1, -2), n(fu).swapaxes(-1, -2), n(Xa), n(Ua), prob["Q"], prob["R"], prob["X_ref"],
                                         prob["U_ref"], prob["reg_x"], prob["reg_u"], 1, prob["u_l"], prob["u_u"], n(Xb), n(Ub), alpha)
        print(f"  smoothed cone objective, SCP iteration {it + 1}: " + ", ".join(f"{k_} {v:.2e}" if isinstance(v, float) else f"{k_} {v}" for k_, v in cert.items() if not k_.startswith("_")),
              flush=True)
        assert cert["slack"] > 0.0 and cert["fractional_multipliers"] >= 1
        assert max(cert["dynamics"], cert["consensus"]) < 1e-9
        assert max(cert["own_controls"], cert["shared_controls"], cert["lam_range"], cert["lam_sum"], cert["threshold_spread"], cert["complementarity"]) < 1e-6, cert
        Xa, Xb, Ua, Ub = Xb, Xa, Ub, Ua


def test_config_D_size_hard_cone_objective_certificate(solver):
    """The reference's DEFAULT solver path (c_lcone_solve, hard boxes) at config D's size: optimality certified from the ABI data alone
    (tests/support/kkt_certificate.hard_cone_certificate) — every particle at its conditional optimum given the shared controls, and
    multipliers of the epigraph rows that sit on the right sides of the threshold cost, sum to (1 - eps) k and leave the shared controls
    stationary (fitted over the costs on the threshold).  The M = 600 comparison with the cone oracle is further up."""
    import torch

    from pmpc_amd import dynamics as dyn
    from pmpc_amd.device import MODEL_QUADROTOR, to_device_problem
    from tests.support.kkt_certificate import hard_cone_certificate

    M, N = 4096, 50
    prob = dyn.make_quadrotor_problem(M=M, N=N)
    d = to_device_problem(prob)
    Xa, Ua = d["X_prev"].clone(), d["U_prev"].clone()
    Xb, Ub = torch.empty_like(Xa), torch.empty_like(Ua)
    for it in range(3):
        f, fx, fu = solver.linearize(MODEL_QUADROTOR, d["x0"], Xa, Ua, d["params"])
        _, _, status = solver.lcone_solve(f=f, fx=fx, fu=fu, X_prev=Xa, U_prev=Ua, Q=d["Q"], R=d["R"], X_ref=d["X_ref"], U_ref=d["U_ref"], reg_x=prob["reg_x"],
                                          reg_u=prob["reg_u"], Nc=1, x0=d["x0"], lu=d["lu"], uu=d["uu"], X_out=Xb, U_out=Ub, symmetric_cost=True,
                                          static_cons_bounds=True, prev_is_last_solution=it > 0)
        solver.sync()
        assert status == 0, solver.last_info
        n = lambda t_: t_.cpu().numpy()
        cert = hard_cone_certificate(prob["x0"], n(f), n(fx).swapaxes(-1, -2), n(fu).swapaxes(-1, -2), n(Xa), n(Ua), prob["Q"], prob["R"], prob["X_ref"],
                                     prob["U_ref"], prob["reg_x"], prob["reg_u"], 1, prob["u_l"], prob["u_u"], n(Xb), n(Ub))
        print(f"  cone objective (hard boxes), SCP iteration {it + 1}: " + ", ".join(f"{k_} {v:.2e}" if isinstance(v, float) else f"{k_} {v}" for k_, v in cert.items() if not k_.startswith("_")),
              flush=True)
        assert max(cert["dynamics"], cert["consensus"], cert["box"]) < 1e-9
        assert max(cert["own_controls"], cert["shared_controls"], cert["lam_sum"], cert["remainder_feasible"]) < 1e-8, cert
        Xa, Xb, Ua, Ub = Xb, Xa, Ub, Ua
