"""Cone path (`c_lcone_solve`, PMPC.jl/src/main.jl:194-354 through the C ABI) on the GPU against the oracle's
exact minimiser of the same epsilon-anchored epigraph problem, the per-particle cost kernel against numpy, and the
per-particle `weights` of the device API against the oracle's weighted joint QP."""
import numpy as np
import pytest

from tests.support.problems import abi_args, rand_problem

pytestmark = pytest.mark.gpu
TOL = 1e-6  # BASELINE.json north_star: fp64 trajectories within 1e-6 relative


def _rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1.0)


def _make_kink(oracle, args, kw, Nc):
    """Scale the cost blocks (Q, R) of the second-cheapest particle until it ties with the cheapest one under uniform
    weights: down-weighting either then lifts it above the other, so the cone optimum sits on the kink J_a = J_b and
    the two share the deficit 2 eps M (the 2-cycle + bisection branch of pmpc_lcone_solve_device)."""
    from scipy.optimize import brentq

    ckw = dict(reg_x=kw["reg_x"], reg_u=kw["reg_u"], slew_reg=kw.get("slew_reg"), slew_reg0=kw.get("slew_reg0"), slew_um1=kw.get("slew_um1"))

    def costs(a_):
        X, U = oracle.lqp_solve_py(*a_, Nc=Nc, **kw)
        return oracle.particle_costs_py(X, U, *a_[4:], **ckw)

    order = np.argsort(costs(args))
    a, b = order[0], order[1]

    def scaled(c):
        a_ = [np.array(v, copy=True) for v in args]
        a_[6][b] *= c
        a_[7][b] *= c
        return tuple(a_)

    def gap(c):
        Jc = costs(scaled(c))
        return Jc[b] - Jc[a]

    c = brentq(gap, 0.05, 1.0, xtol=1e-12)
    return scaled(c * (1 + 1e-9)), kw


# (M, N, x, u, Nc, u-bound, x-bound, slew, slew0, kink construction)
CONE_CASES = [
    (1, 6, 3, 2, 0, 0.3, None, None, None, None),
    (4, 8, 3, 2, 2, 0.3, None, None, None, None),
    (120, 6, 2, 1, 1, 0.5, None, None, None, None),
    (300, 4, 2, 1, -1, 0.5, None, None, None, None),
    (40, 6, 3, 2, 2, 0.4, 6.0, 0.5, 0.3, None),
    (60, 6, 4, 2, 1, 1.0, None, None, None, True),
    (100, 6, 2, 1, -1, 0.3, None, None, None, True),
    (30, 6, 3, 2, 2, 0.3, None, None, None, True),
    (25, 7, 12, 4, 3, 0.4, None, None, None, None),
]


@pytest.mark.parametrize("case", CONE_CASES, ids=[f"M{c[0]}N{c[1]}x{c[2]}u{c[3]}Nc{c[4]}" + ("kink" if c[9] else "") + ("slew" if c[7] else "") for c in CONE_CASES])
def test_c_lcone_solve_matches_cone_oracle(case, oracle):
    from pmpc_amd import backend

    M, N, x, u, Nc = case[:5]
    args, kw = rand_problem(np.random.default_rng(4000 + CONE_CASES.index(case)), M, N, x, u, *case[5:9])
    if case[9]:
        args, kw = _make_kink(oracle, args, kw, Nc)
    Xo, Uo, info = oracle.lcone_solve_py(*args, Nc=Nc, return_info=True, **kw)
    X, U = backend.lcone_solve(*abi_args(args, kw, Nc), smooth_alpha=float("nan"), solver="ecos")
    assert bool(info.get("kink", False)) == bool(case[9])
    assert _rel(X, Xo) < TOL and _rel(U, Uo) < TOL, (info["weights"][np.argsort(info["J"])[:3]], info.get("kink"))
    if M > 1:
        # the device result minimises the reference's objective: value within round-off of the oracle's
        ckw = dict(reg_x=kw["reg_x"], reg_u=kw["reg_u"], slew_reg=kw.get("slew_reg"), slew_reg0=kw.get("slew_reg0"), slew_um1=kw.get("slew_um1"))
        J = oracle.particle_costs_py(X, U, *args[4:], **ckw)
        assert abs(oracle.cone_objective(J) - oracle.cone_objective(info["J"])) <= 1e-8 * abs(oracle.cone_objective(info["J"]))


@pytest.mark.parametrize("alpha", [1e1, 1e3])
@pytest.mark.parametrize("idx", [0, 1, 2, 4, 6, 8])
def test_c_lcone_solve_logbarrier_smoothing(idx, alpha, oracle):
    """smooth_alpha finite => smooth_cstr = "logbarrier" (PMPC.jl/src/main.jl:240-262): every box row becomes
    -1/alpha log(alpha slack) in the objective; device: the interior-point iteration stops AT mu = 1/alpha."""
    from pmpc_amd import backend

    case = CONE_CASES[idx]
    M, N, x, u, Nc = case[:5]
    args, kw = rand_problem(np.random.default_rng(4000 + idx), M, N, x, u, *case[5:9])
    if case[9]:
        args, kw = _make_kink(oracle, args, kw, Nc)
    Xo, Uo, info = oracle.lcone_solve_py(*args, Nc=Nc, return_info=True, smooth_alpha=alpha, **kw)
    X, U = backend.lcone_solve(*abi_args(args, kw, Nc), smooth_alpha=alpha, solver="ecos")
    assert _rel(X, Xo) < TOL and _rel(U, Uo) < TOL
    Xh, Uh = oracle.lcone_solve_py(*args, Nc=Nc, **kw)
    assert _rel(U, Uh) > 1e-5  # and it is not the hard-constrained optimum


@pytest.mark.parametrize("dims", [(6, 9, 12, 4, 1, False), (6, 9, 12, 4, 3, False), (5, 8, 3, 2, 2, True), (7, 6, 4, 2, -1, False)])
def test_weighted_lqp_and_particle_costs(dims, oracle):
    import torch

    from pmpc_amd.device import DeviceSolver

    M, N, x, u, Nc, slew = dims
    rng = np.random.default_rng(99 + M)
    args, kw = rand_problem(rng, M, N, x, u, 0.3, None, 0.5 if slew else None, 0.3 if slew else None)
    wts = 0.2 + rng.random(M)
    Xo, Uo = oracle.lqp_solve_py(*args, Nc=Nc, weights=wts, **kw)
    x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
    dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda")
    T = lambda a: dev(np.swapaxes(a, -1, -2))
    s = DeviceSolver(0)
    common = dict(f=dev(f), fx=T(fx), fu=T(fu), X_prev=dev(X_prev), U_prev=dev(U_prev), Q=T(Q), R=T(R), X_ref=dev(X_ref),
                  U_ref=dev(U_ref), reg_x=kw["reg_x"], reg_u=kw["reg_u"], Nc=Nc, lu=dev(kw["u_l"]), uu=dev(kw["u_u"]),
                  symmetric_cost=True)
    if slew:
        common.update(slew_reg=dev(kw["slew_reg"]), slew_reg0=dev(kw["slew_reg0"]), slew_um1=dev(kw["slew_um1"]))
    X, U, status = s.lqp_solve(weights=dev(wts), **common)
    s.sync()
    assert status == 0 and s.last_info["fast_path"] == 1  # (boxed slew: the increment form with state-box rounds since r03)
    assert _rel(X.cpu().numpy(), Xo) < TOL and _rel(U.cpu().numpy(), Uo) < TOL
    J = s.particle_costs(X, U, **common).cpu().numpy()
    Jo = oracle.particle_costs_py(X.cpu().numpy(), U.cpu().numpy(), X_prev, U_prev, Q, R, X_ref, U_ref, reg_x=kw["reg_x"],
                                  reg_u=kw["reg_u"], slew_reg=kw.get("slew_reg"), slew_reg0=kw.get("slew_reg0"),
                                  slew_um1=kw.get("slew_um1"))
    np.testing.assert_allclose(J, Jo, rtol=1e-12)
    s.close()


def test_warm_start_paths_give_the_cold_start_optimum(oracle):
    """The interior-point warm start across calls: from a related problem, from an UNRELATED one
    of the same shape, rejected because the remembered controls violate tighter boxes, and switched off — always the
    oracle's optimum."""
    import torch

    from pmpc_amd.device import DeviceSolver

    M, N, x, u, Nc = 9, 12, 4, 2, 1
    dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda")
    T = lambda a: dev(np.swapaxes(a, -1, -2))
    s = DeviceSolver(0)

    def solve(args, kw, **extra):
        x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
        X, U, status = s.lqp_solve(f=dev(f), fx=T(fx), fu=T(fu), X_prev=dev(X_prev), U_prev=dev(U_prev), Q=T(Q), R=T(R), X_ref=dev(X_ref),
                                   U_ref=dev(U_ref), reg_x=kw["reg_x"], reg_u=kw["reg_u"], Nc=Nc, lu=dev(kw["u_l"]), uu=dev(kw["u_u"]),
                                   symmetric_cost=True, **extra)
        s.sync()
        assert status == 0
        return X.cpu().numpy(), U.cpu().numpy(), dict(s.last_info)

    def check(args, kw, **extra):
        Xo, Uo = oracle.lqp_solve_py(*args, Nc=Nc, **kw)
        X, U, info = solve(args, kw, **extra)
        assert _rel(X, Xo) < 1e-7 and _rel(U, Uo) < 1e-7
        return info

    argsA, kwA = rand_problem(np.random.default_rng(1), M, N, x, u, 0.4)
    check(argsA, kwA)                                   # cold (nothing remembered for this shape yet)
    argsA2 = tuple(a + 0.02 * np.random.default_rng(2).standard_normal(a.shape) if k in (1, 4, 5) else a for k, a in enumerate(argsA))
    warm = check(argsA2, kwA)                           # related problem: warm
    cold = check(argsA2, kwA, cold_start=True)
    assert warm["structured_solves"] <= cold["structured_solves"]
    argsB, kwB = rand_problem(np.random.default_rng(3), M, N, x, u, 0.4)
    check(argsB, kwB)                                   # unrelated problem of the same shape: warm from A2's iterate
    kwT = dict(kwB, u_l=-0.05 * np.ones((M, N, u)), u_u=0.05 * np.ones((M, N, u)))
    check(argsB, kwT)                                   # remembered controls lie outside the tighter boxes: rejected, cold
    check(argsB, kwT)                                   # and warm again
    s.close()


# (M, N, x, u, Nc, u-bound): stage cone || (u_1, .., u_q) || <= 0.5 u_0 + 0.05 on every stage's controls
SOC_CASES = [(3, 6, 12, 4, 1, 0.8), (4, 8, 4, 2, 0, 0.6), (5, 6, 3, 3, -1, 0.6), (6, 10, 5, 3, 2, None), (40, 12, 12, 4, 1, 0.8)]


@pytest.mark.parametrize("case", SOC_CASES, ids=[str(c) for c in SOC_CASES])
def test_stage_cones_match_cone_oracle(case, oracle):
    """Config E's constraint type (stage-wise second-order cones on the controls, next to the boxes) — the device's
    primal log-barrier Newton on the Riccati kernels against the oracle's path following on the sparse joint KKT system."""
    import torch

    from pmpc_amd.device import DeviceSolver

    M, N, x, u, Nc, bu = case
    args, kw = rand_problem(np.random.default_rng(7000 + SOC_CASES.index(case)), M, N, x, u, bu)
    W = np.zeros((u - 1, u))
    W[np.arange(u - 1), np.arange(1, u)] = 1.0
    w0, v, v0 = np.zeros(u - 1), np.eye(u)[0] * 0.5, 0.05
    u_int = np.eye(u)[0] * 0.2
    Xo, Uo = oracle.lsoc_solve_py(*args, Nc=Nc, reg_x=kw["reg_x"], reg_u=kw["reg_u"], u_l=kw.get("u_l"), u_u=kw.get("u_u"), soc_W=W,
                                  soc_w0=w0, soc_v=v, soc_v0=v0, u_interior=u_int)
    x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
    dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda")
    T = lambda a: dev(np.swapaxes(a, -1, -2))
    s = DeviceSolver(0)
    bounds = dict(lu=dev(kw["u_l"]), uu=dev(kw["u_u"])) if bu is not None else {}
    X, U, status = s.lsoc_solve(f=dev(f), fx=T(fx), fu=T(fu), X_prev=dev(X_prev), U_prev=dev(U_prev), Q=T(Q), R=T(R), X_ref=dev(X_ref),
                                U_ref=dev(U_ref), reg_x=kw["reg_x"], reg_u=kw["reg_u"], Nc=Nc, symmetric_cost=True, soc_W=dev(W),
                                soc_w0=dev(w0), soc_v=dev(v), soc_v0=v0, soc_u_interior=dev(u_int), **bounds)
    s.sync()
    assert status == 0
    X, U = X.cpu().numpy(), U.cpu().numpy()
    slack = 0.5 * U[..., 0] + 0.05 - np.linalg.norm(U[..., 1:], axis=-1)
    assert slack.min() > -1e-9 and (slack < 1e-6).sum() > 0  # inside every cone (to round-off), and the cone is active somewhere
    assert _rel(X, Xo) < TOL and _rel(U, Uo) < TOL
    s.close()


@pytest.mark.parametrize("M,k", [(40, 20), (40, 30), (40, 39), (24, 1), (600, 300)])
def test_worst_k_objective_matches_cone_oracle(oracle, M, k):
    """The reference's `k` setting (PMPC.jl/src/main.jl:204-227, pyjulia only): weight (1 - eps) k on the epigraph offset, i.e.
    only the ~k (1 - eps) / (1 + eps) costliest particles carry weight.  Zero-weight particles: stated semantics of the oracle
    (`_lcone_many_particles`) — each minimises its own cost given the shared controls; the device keeps a floor weight 1e-4."""
    from pmpc_amd import backend

    rng = np.random.default_rng(7 + M + k)
    args, kw = rand_problem(rng, M, 6, 4, 2, 0.4)
    args = list(args)
    scale = (1.0 + 2.0 * rng.permutation(M) / M)[:, None, None, None]  # well-separated particle costs: a stable ranking (kinks
    args[6], args[7] = args[6] * scale, args[7] * scale               # between more than two particles are out of scope)
    args = tuple(args)
    Xo, Uo, info = oracle.lcone_solve_py(*args, Nc=1, return_info=True, k=k, **kw)
    X, U = backend.lcone_solve(*abi_args(args, kw, 1), smooth_alpha=float("nan"), solver="ecos", k=k)
    assert np.all(np.isfinite(X)) and np.all(np.isfinite(U))
    assert _rel(X, Xo) < TOL and _rel(U, Uo) < TOL, (_rel(X, Xo), _rel(U, Uo))


def test_host_loop_takes_the_stage_cone_as_an_extra_cstrs_tuple(oracle):
    """`pmpc_amd.solve(..., solver_settings=dict(extra_cstrs=[tuple]))` with the reference-format tuple of the stage-wise thrust
    cone (README.md:219-239): the host back end routes the sub-problem to the device cone solver; one sub-problem against
    the cone oracle."""
    from pmpc_amd import backend
    from pmpc_amd.extra_cstrs import stage_soc_to_extra_cstrs

    M, N, x, u, Nc = 6, 8, 4, 3, 1
    rng = np.random.default_rng(11)
    args, kw = rand_problem(rng, M, N, x, u, 0.6)
    W = np.zeros((2, 3)); W[0, 1] = W[1, 2] = 1.0
    w0, v, v0 = np.zeros(2), np.array([0.5, 0.0, 0.0]), 0.1
    cstr = stage_soc_to_extra_cstrs(W, w0, v, v0, M, N, x, u, Nc)
    x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
    Xo, Uo = oracle.lsoc_solve_py(*args, Nc=Nc, reg_x=kw["reg_x"], reg_u=kw["reg_u"], u_l=kw["u_l"], u_u=kw["u_u"], soc_W=W, soc_w0=w0, soc_v=v,
                                  soc_v0=v0, u_interior=np.zeros(3))
    X, U, _ = backend.aff_solve(f, fx, fu, x0, X_prev, U_prev, Q, R, X_ref, U_ref, kw["reg_x"], kw["reg_u"], 0.0, None, np.zeros((0, 0, 0)),
                                np.zeros((0, 0, 0)), kw["u_l"], kw["u_u"], solver_settings=dict(solver="osqp", Nc=Nc, extra_cstrs=[cstr],
                                                                                              soc_u_interior=np.zeros(3)))
    assert _rel(X[:, 1:], Xo) < TOL and _rel(U, Uo) < TOL, (_rel(X[:, 1:], Xo), _rel(U, Uo))


def test_the_settled_weight_assignment_starts_the_next_cone_solve(oracle):
    """Option cone_rank_memory (default on): the assignment of weights by cost rank that the previous solve of a shape settled on
    is tried first — accepted after ONE weighted QP when the ranking at its optimum reproduces it (the same consistency test, i.e.
    the KKT conditions of the reference's epigraph problem, PMPC.jl/src/main.jl:204-227), otherwise the iteration goes on from
    there.  Same answers as a context without the memory, on a sequence of related problems, and the oracle's."""
    import torch

    from pmpc_amd.device import DeviceSolver

    M, N, x, u, Nc = 40, 8, 4, 2, 1
    rng = np.random.default_rng(321)
    args, kw = rand_problem(rng, M, N, x, u, 0.5)
    dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda")
    T = lambda a: dev(np.swapaxes(a, -1, -2))
    mem, plain = DeviceSolver(0), DeviceSolver(0)
    plain.set_option("cone_rank_memory", 0)
    outer = []
    for t in range(4):
        if t:
            x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
            args = (x0, f + (0.5 if t == 3 else 0.01) * rng.standard_normal(f.shape), fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref)  # t = 3: the ranking changes
        x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
        Xo, Uo = oracle.lcone_solve_py(*args, Nc=Nc, **kw)
        opt = dict(f=dev(f), fx=T(fx), fu=T(fu), X_prev=dev(X_prev), U_prev=dev(U_prev), Q=T(Q), R=T(R), X_ref=dev(X_ref), U_ref=dev(U_ref),
                   reg_x=kw["reg_x"], reg_u=kw["reg_u"], Nc=Nc, lu=dev(kw["u_l"]), uu=dev(kw["u_u"]), symmetric_cost=True)
        res = []
        for s in (mem, plain):
            X, U, status = s.lcone_solve(**opt)
            s.sync()
            assert status == 0
            assert _rel(X.cpu().numpy(), Xo) < TOL and _rel(U.cpu().numpy(), Uo) < TOL, (t, s is mem)
            res.append((X.clone(), U.clone(), s.last_info["outer_solves"]))
        assert (res[0][0] - res[1][0]).abs().max().item() <= 1e-9 and (res[0][1] - res[1][1]).abs().max().item() <= 1e-9
        outer.append((res[0][2], res[1][2]))
    assert outer[0][0] == outer[0][1] >= 2  # nothing remembered yet
    assert outer[1][0] == 1 and outer[2][0] == 1 and outer[1][1] >= 2  # the remembered assignment is consistent: one weighted QP
    mem.close()
    plain.close()


def test_cone_objective_and_particle_costs_on_fp32_stored_matrices(oracle):
    """PMPC_F32_MATRICES meets the cone path: the weighted QPs and the particle-cost kernel read doubles, so the float blocks are
    widened first (`bench.py --cone --fp32` once read them as doubles, past their end).  Same answer as the fp64 call to the storage
    rounding, costs included."""
    import torch

    from pmpc_amd.device import DeviceSolver

    M, N, x, u, Nc = 12, 8, 4, 2, 1
    args, kw = rand_problem(np.random.default_rng(77), M, N, x, u, 0.5)
    x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
    dev = lambda a, dt=torch.float64: torch.tensor(np.ascontiguousarray(a), dtype=dt, device="cuda")
    T = lambda a, dt=torch.float64: dev(np.swapaxes(a, -1, -2), dt)
    s = DeviceSolver(0)
    out = {}
    for name, dt in (("f64", torch.float64), ("f32", torch.float32)):
        opt = dict(f=dev(f), fx=T(fx, dt), fu=T(fu, dt), X_prev=dev(X_prev), U_prev=dev(U_prev), Q=T(Q, dt), R=T(R, dt), X_ref=dev(X_ref),
                   U_ref=dev(U_ref), reg_x=kw["reg_x"], reg_u=kw["reg_u"], Nc=Nc, lu=dev(kw["u_l"]), uu=dev(kw["u_u"]), symmetric_cost=True)
        X, U, status = s.lcone_solve(cold_start=True, **opt)
        s.sync()
        assert status == 0
        J = s.particle_costs(X, U, **opt)
        out[name] = (X.cpu().numpy(), U.cpu().numpy(), J.cpu().numpy())
    Xo, Uo = oracle.lcone_solve_py(*args, Nc=Nc, **kw)
    assert _rel(out["f64"][0], Xo) < TOL and _rel(out["f64"][1], Uo) < TOL
    assert _rel(out["f32"][0], Xo) < 1e-5 and _rel(out["f32"][1], Uo) < 1e-5  # (storage rounding of fx, fu, Q, R: 6e-8 relative each)
    assert np.allclose(out["f32"][2], out["f64"][2], rtol=1e-5)
    s.close()
