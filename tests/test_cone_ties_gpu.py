"""`c_lcone_solve` with MORE than two particle costs on the threshold of the epigraph problem (PMPC.jl/src/main.jl:204-238) — the case
the rank-based weight assignment could not place — against the DIRECT restatement of the reference's cone program
(oracle/cone_oracle.py: Pqr2Gh cones handed to a conic solver, no elimination of (y, t), no search over threshold particles)."""
import numpy as np
import pytest

from tests.support.problems import abi_args, rand_problem

pytestmark = pytest.mark.gpu
TOL = 1e-6  # BASELINE.json north_star


def _rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1.0)


@pytest.fixture(scope="module")
def co():
    from oracle import cone_oracle, lqp_oracle

    lqp_oracle.build()
    return cone_oracle


def tied_problem(rng, copies, others, N=6, x=4, u=2, bu=0.4, Nc=1):
    """`copies` identical particles of the LOWEST cost + `others` costlier ones: the identical ones sit together on the threshold."""
    from oracle import lqp_oracle as orc

    args, kw = rand_problem(rng, copies + others, N, x, u, bu)
    X, U = orc.lqp_solve_py(*args, Nc=Nc, **kw)
    x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
    J = orc.particle_costs_py(X, U, X_prev, U_prev, Q, R, X_ref, U_ref, reg_x=kw["reg_x"], reg_u=kw["reg_u"])
    order = np.argsort(J)
    keep = np.concatenate([[order[0]] * copies, order[len(order) - others:]]).astype(int)
    args = tuple(np.array(np.asarray(a)[keep], copy=True) for a in args)
    kw = {k: (np.array(np.asarray(v)[keep], copy=True) if isinstance(v, np.ndarray) and v.ndim and v.shape[0] == copies + others else v) for k, v in kw.items()}
    return args, kw


@pytest.mark.parametrize("copies,others,dims,Nc", [(3, 3, (6, 4, 2), 1), (5, 3, (6, 4, 2), 1), (9, 4, (6, 4, 2), 1), (3, 2, (5, 12, 4), 1), (4, 3, (6, 4, 2), 2),
                                                    (5, 3, (5, 3, 2), -1)])
def test_hard_boxes_ties_among_identical_particles(co, copies, others, dims, Nc):
    from pmpc_amd import backend

    N, x, u = dims
    args, kw = tied_problem(np.random.default_rng(100 * copies + others + x), copies, others, N, x, u, 0.4, Nc)
    Xo, Uo, info = co.lcone_direct_py(*args, Nc=Nc, return_info=True, **kw)
    X, U = backend.lcone_solve(*abi_args(args, kw, Nc), smooth_alpha=float("nan"), solver="ecos")
    assert np.all(np.isfinite(X)) and np.all(np.isfinite(U))
    assert _rel(X, Xo) <= TOL and _rel(U, Uo) <= TOL, (_rel(X, Xo), _rel(U, Uo))
    for i in range(1, copies):  # identical particles, identical trajectories
        assert np.abs(X[i] - X[0]).max() <= 1e-9 and np.abs(U[i] - U[0]).max() <= 1e-9


@pytest.mark.parametrize("M,dims,Nc", [(8, (6, 4, 2), 1), (16, (5, 12, 4), 1), (6, (6, 4, 2), -1)])
def test_hard_boxes_all_particles_identical(co, M, dims, Nc):
    """The `Problem` builder tiles ONE problem over M particles (pmpc/problem_struct.py:88-102) and the default solver is "ecos": every
    cost ties exactly.  The optimum is the single particle's QP optimum, M times."""
    from oracle import lqp_oracle as orc
    from pmpc_amd import backend

    N, x, u = dims
    a1, kw = rand_problem(np.random.default_rng(7 + M), 1, N, x, u, 0.4)
    args = tuple(np.repeat(np.asarray(a), M, axis=0) for a in a1)
    kwM = {k: (np.repeat(v, M, axis=0) if isinstance(v, np.ndarray) and v.ndim and v.shape[0] == 1 else v) for k, v in kw.items()}
    X1, U1 = orc.lqp_solve_py(*a1, Nc=Nc, **kw)
    X, U = backend.lcone_solve(*abi_args(args, kwM, Nc), smooth_alpha=float("nan"), solver="ecos")
    for i in range(M):
        assert _rel(X[i:i + 1], X1) <= 1e-7 and _rel(U[i:i + 1], U1) <= 1e-7
    if M <= 8:
        Xo, Uo = co.lcone_direct_py(*args, Nc=Nc, **kwM)
        assert _rel(X, Xo) <= TOL and _rel(U, Uo) <= TOL


@pytest.mark.parametrize("seed,M,Nc", [(1, 9, 1), (2, 20, 2), (3, 9, -1)])
def test_hard_boxes_random_particles_match_the_direct_program(co, seed, M, Nc):
    from pmpc_amd import backend

    args, kw = rand_problem(np.random.default_rng(900 + seed), M, 6, 4, 2, 0.4)
    Xo, Uo = co.lcone_direct_py(*args, Nc=Nc, **kw)
    X, U = backend.lcone_solve(*abi_args(args, kw, Nc), smooth_alpha=float("nan"), solver="ecos")
    assert _rel(X, Xo) <= TOL and _rel(U, Uo) <= TOL, (_rel(X, Xo), _rel(U, Uo))


def test_weightless_particles_take_their_minimum_cost_completion():
    """M > (1 + eps) / (2 eps) ~ 500: the cheapest particles carry NO multiplier; the reference's minimiser is not unique in them
    (stated semantics = each of them minimises its own cost given the shared controls).  The epigraph path
    runs the sweeps unweighted, so that is what comes out exactly — no floor weight."""
    from oracle import lqp_oracle as orc
    from pmpc_amd import backend

    M, N, x, u, Nc = 640, 5, 2, 1, 1
    args, kw = rand_problem(np.random.default_rng(77), M, N, x, u, 0.5)
    Xo, Uo, info = orc.lcone_solve_py(*args, Nc=Nc, return_info=True, **kw)
    X, U = backend.lcone_solve(*abi_args(args, kw, Nc), smooth_alpha=float("nan"), solver="ecos")
    assert _rel(X, Xo) <= TOL and _rel(U, Uo) <= TOL, (_rel(X, Xo), _rel(U, Uo))


@pytest.mark.parametrize("model,M,N,Nc,steps", [("quadrotor", 96, 20, 1, 4), ("unicycle", 40, 12, 3, 2), ("quadrotor", 640, 8, 1, 3)])
def test_library_scp_loop_drives_the_cone_objective(model, M, N, Nc, steps):
    """pmpc_scp_loop_device with PMPC_CONE_OBJECTIVE (the reference's default solver path as the sub-problem of every iteration; the
    epigraph rows are checked on the device behind the first batch of rounds) walks the sequence of a Python loop of linearise /
    lcone_solve / residual calls: same residuals and iterates."""
    import torch

    from pmpc_amd import dynamics as dyn
    from pmpc_amd.device import MODEL_QUADROTOR, MODEL_UNICYCLE, DeviceSolver, to_device_problem

    prob = dyn.make_quadrotor_problem(M=M, N=N, Nc=Nc) if model == "quadrotor" else dyn.make_unicycle_problem(M=M, N=N, Nc=Nc)
    mid = MODEL_QUADROTOR if model == "quadrotor" else MODEL_UNICYCLE
    d = to_device_problem(prob)
    common = dict(Q=d["Q"], R=d["R"], X_ref=d["X_ref"], U_ref=d["U_ref"], reg_x=prob["reg_x"], reg_u=prob["reg_u"], Nc=Nc, x0=d["x0"], lu=d["lu"],
                  uu=d["uu"], symmetric_cost=True)
    # (the unicycle's `where(u >= 0, eps, -eps)` turns a 1e-12 difference into 1e-6 within three iterations: two iterations there)
    outs = []
    for lib_loop in (False, True):
        solver = DeviceSolver(0)  # fresh memories for both runs
        Xa, Ua = d["X_prev"].clone(), d["U_prev"].clone()
        Xb, Ub = torch.empty_like(Xa), torch.empty_like(Ua)
        if not lib_loop:
            res = []
            for it in range(steps):
                f, fx, fu = solver.linearize(mid, d["x0"], Xa, Ua, d["params"])
                _, _, st = solver.lcone_solve(f=f, fx=fx, fu=fu, X_prev=Xa, U_prev=Ua, X_out=Xb, U_out=Ub, static_cons_bounds=True,
                                              prev_is_last_solution=it > 0, cold_start=it == 0, **common)
                assert st == 0, (it, solver.last_info)
                res.append(float(solver.scp_residual(Xb, Xa, Ub, Ua)[0].item()))
                Xa, Xb, Ua, Ub = Xb, Xa, Ub, Ua
            outs.append((np.array(res), Xa.clone(), Ua.clone()))
        else:
            x, u = Xa.shape[-1], Ua.shape[-1]
            mk = lambda *shape: torch.empty(shape, dtype=torch.float64, device="cuda")
            bufs = [mk(M, N, x), mk(M, N, x, x), mk(M, N, u, x), mk(M, N, x), mk(M, N, x, x), mk(M, N, u, x)]
            res, infos, last_in_out, done = solver.scp_loop(mid, d["params"], steps, f=bufs[0], fx=bufs[1], fu=bufs[2], f2=bufs[3], fx2=bufs[4],
                                                            fu2=bufs[5], X_prev=Xa, U_prev=Ua, X_out=Xb, U_out=Ub, first_cold=True, cone_objective=True,
                                                            **common)
            solver.sync()
            assert done == steps and all(i["status"] == 0 for i in infos), infos
            X_lib, U_lib = (Xb, Ub) if last_in_out else (Xa, Ua)
            outs.append((res.cpu().numpy(), X_lib.clone(), U_lib.clone()))
        solver.close()
    # (same sequence; not bit for bit: a point whose epigraph rows are consistent to the 1e-9 of the acceptance test is accepted on
    #  either path, and the two reach it through different launch sequences)
    np.testing.assert_allclose(outs[1][0], outs[0][0], rtol=1e-6, atol=1e-9)
    assert torch.allclose(outs[1][1], outs[0][1], rtol=0, atol=1e-6) and torch.allclose(outs[1][2], outs[0][2], rtol=0, atol=1e-6)


@pytest.mark.parametrize("copies,others,alpha,dims", [(3, 3, 10.0, (6, 4, 2)), (5, 3, 10.0, (6, 4, 2)), (9, 4, 1.0, (6, 4, 2)), (3, 3, 100.0, (6, 4, 2))])  # ((3, 2, 10.0, (5, 12, 4)) passes too: 41 s of oracle time)
def test_logbarrier_smoothing_with_ties(co, copies, others, alpha, dims):
    """Smoothed boxes (main.jl:246-262) and several particles on the threshold: with the barrier every particle whose cost range brackets
    the threshold carries a fractional multiplier — the full-space Newton iteration of lcone_smooth_body against the direct program."""
    from pmpc_amd import backend

    N, x, u = dims
    args, kw = tied_problem(np.random.default_rng(300 * copies + others + x), copies, others, N, x, u, 0.4, 1)
    Xo, Uo = co.lcone_direct_py(*args, Nc=1, smooth_alpha=alpha, **kw)
    X, U = backend.lcone_solve(*abi_args(args, kw, 1), smooth_alpha=alpha, solver="ecos")
    assert np.all(np.isfinite(X)) and np.all(np.isfinite(U))
    assert _rel(X, Xo) <= TOL and _rel(U, Uo) <= TOL, (_rel(X, Xo), _rel(U, Uo))
    for i in range(1, copies):
        assert np.abs(X[i] - X[0]).max() <= 1e-8 and np.abs(U[i] - U[0]).max() <= 1e-8


@pytest.mark.parametrize("seed,M,alpha", [(1, 12, 10.0), (2, 14, 1.0), (3, 8, 100.0)])
def test_logbarrier_smoothing_random_particles_match_the_direct_program(co, seed, M, alpha):
    from pmpc_amd import backend

    args, kw = rand_problem(np.random.default_rng(950 + seed), M, 6, 4, 2, 0.4)
    Xo, Uo = co.lcone_direct_py(*args, Nc=1, smooth_alpha=alpha, **kw)
    X, U = backend.lcone_solve(*abi_args(args, kw, 1), smooth_alpha=alpha, solver="ecos")
    assert _rel(X, Xo) <= TOL and _rel(U, Uo) <= TOL, (_rel(X, Xo), _rel(U, Uo))


def test_logbarrier_smoothing_all_particles_identical(co):
    from pmpc_amd import backend

    M, N, x, u = 6, 6, 4, 2
    a1, kw = rand_problem(np.random.default_rng(21), 1, N, x, u, 0.4)
    args = tuple(np.repeat(np.asarray(a), M, axis=0) for a in a1)
    kwM = {k: (np.repeat(v, M, axis=0) if isinstance(v, np.ndarray) and v.ndim and v.shape[0] == 1 else v) for k, v in kw.items()}
    Xo, Uo = co.lcone_direct_py(*args, Nc=1, smooth_alpha=10.0, **kwM)
    X, U = backend.lcone_solve(*abi_args(args, kwM, 1), smooth_alpha=10.0, solver="ecos")
    assert _rel(X, Xo) <= TOL and _rel(U, Uo) <= TOL, (_rel(X, Xo), _rel(U, Uo))


@pytest.mark.parametrize("alpha,beta,copies", [(10.0, 1.0, 1), (10.0, 5.0, 3)])  # ((5, 20, 1) passes too: 49 s of oracle time, run by hand)
def test_squareplus_smoothing_matches_the_direct_program(co, alpha, beta, copies):
    """smooth_cstr = "squareplus" (main.jl:265-279, cone_utils.jl:222-228): every box side costs beta/2 (v + sqrt(v^2 + 1/alpha^2)) of its
    violation v — soft boxes.  The reference states it as three-row second-order cones with a new epigraph variable each; the direct
    oracle solves exactly those rows."""
    from pmpc_amd import backend

    if copies > 1:
        args, kw = tied_problem(np.random.default_rng(41), copies, 3, 6, 4, 2, 0.4, 1)
    else:
        args, kw = rand_problem(np.random.default_rng(40), 7, 6, 4, 2, 0.4)
    Xo, Uo = co.lcone_direct_py(*args, Nc=1, smooth_alpha=alpha, smooth_cstr="squareplus", smooth_beta=beta, **kw)
    X, U = backend.lcone_solve(*abi_args(args, kw, 1), smooth_alpha=alpha, solver="ecos", smooth_cstr="squareplus", smooth_beta=beta)
    assert np.all(np.isfinite(X)) and np.all(np.isfinite(U))
    assert _rel(X, Xo) <= TOL and _rel(U, Uo) <= TOL, (_rel(X, Xo), _rel(U, Uo))
    if beta >= 20.0:  # a steep hinge keeps the controls (nearly) inside the boxes it replaces
        assert np.max(np.abs(U)) <= 0.4 + 0.05


@pytest.mark.parametrize("copies,others,alpha,Nc", [(3, 3, 10.0, 2), (4, 2, 10.0, -1), (1, 6, 1.0, 3)])
def test_logbarrier_smoothing_with_several_consensus_stages(co, copies, others, alpha, Nc):
    """The reference's DEFAULT consensus horizon is Nc = N (main.jl:127-128): smoothing, ties and several shared stages together (the
    condensing kernel supplies the off-diagonal blocks of the per-particle condensed Hessians to the Newton system)."""
    from pmpc_amd import backend

    if copies > 1:
        args, kw = tied_problem(np.random.default_rng(500 + copies + others), copies, others, 6, 4, 2, 0.4, Nc)
    else:
        args, kw = rand_problem(np.random.default_rng(501), others, 6, 4, 2, 0.4)
    Xo, Uo = co.lcone_direct_py(*args, Nc=Nc, smooth_alpha=alpha, **kw)
    X, U = backend.lcone_solve(*abi_args(args, kw, Nc), smooth_alpha=alpha, solver="ecos")
    assert np.all(np.isfinite(X)) and np.all(np.isfinite(U))
    assert _rel(X, Xo) <= TOL and _rel(U, Uo) <= TOL, (_rel(X, Xo), _rel(U, Uo))


@pytest.mark.parametrize("case", [37, 41, 74])
def test_worst_k_with_several_costs_on_the_threshold(case):
    """k < M puts several particle costs on the threshold (here 2, 4 and 3 of 8 / 8 / 7 random particles) with boxes that do not bind:
    no active-set state for the epigraph path to start from, more ties than the ranking iteration knows — the free-particles path
    (`lcone_free_particles_body`: every cost an exact quadratic of the shared controls, one host epigraph solve) answers.  Fixture:
    tests/golden/worstk_ties.npz = problems found by tools/fuzz/fuzz_cone.py + the restated reference program's optimum
    (tools/make_cone_worstk_golden.py).  The particles BELOW the threshold carry no multiplier (not unique upstream): compared are
    the shared controls, the particles on or above the threshold and the objective."""
    from pathlib import Path

    from oracle import lqp_oracle as orc
    from pmpc_amd import backend

    g = np.load(Path(__file__).parent / "golden" / "worstk_ties.npz")
    names = ["x0", "f", "fx", "fu", "X_prev", "U_prev", "Q", "R", "X_ref", "U_ref"]
    args = tuple(g[f"c{case}_{n}"] for n in names)
    kw = {k_[len(f"c{case}_kw_"):]: (float(g[k_]) if g[k_].ndim == 0 else g[k_]) for k_ in g.files if k_.startswith(f"c{case}_kw_")}
    Nc, k = (int(v) for v in g[f"c{case}_meta"])
    Xo, Uo = g[f"c{case}_X"], g[f"c{case}_U"]
    X, U = backend.lcone_solve(*abi_args(args, kw, Nc), smooth_alpha=float("nan"), solver="ecos", k=k)
    assert np.all(np.isfinite(U)), "solver failed"
    x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
    Jg, Jo = (orc.particle_costs_py(X_, U_, X_prev, U_prev, Q, R, X_ref, U_ref, reg_x=kw["reg_x"], reg_u=kw["reg_u"]) for X_, U_ in ((X, U), (Xo, Uo)))
    obj = lambda Jv: min((1 + 1e-3) * np.sum(np.maximum(Jv - t_, 0.0)) + (1 - 1e-3) * k * t_ for t_ in Jv)
    top = Jo >= np.sort(Jo)[::-1][k - 1] - 1e-9 * np.abs(Jo).max()
    Ncc = U.shape[1] if Nc < 0 else Nc
    assert abs(obj(Jg) - obj(Jo)) <= 1e-9 * abs(obj(Jo))
    assert _rel(U[:, :Ncc], Uo[:, :Ncc]) <= TOL and _rel(X[top], Xo[top]) <= TOL and _rel(U[top], Uo[top]) <= TOL


@pytest.mark.parametrize("Nc,slew,slew0", [(1, None, None), (-1, None, None), (0, None, None), (1, 0.5, None), (-1, 0.3, 0.4), (0, 0.7, 0.2), (2, 0.7, 0.2)])
def test_squareplus_with_one_particle_and_slew_matches_the_direct_program(co, Nc, slew, slew0):
    """M = 1 is the shape of nearly every reference example, and `lcone_solve` takes the squareplus branch for any M, with slew penalties
    (main.jl:265-279).  One particle: the epigraph row is degenerate and the solve is damped Newton on (1 - eps) J + the hinges; slew
    penalties through the increment form (control boxes and their hinges become boxes on the u-part of the state)."""
    from pmpc_amd import backend

    args, kw = rand_problem(np.random.default_rng(70 + abs(Nc) + (3 if slew else 0)), 1, 6, 4, 2, 0.4, None, slew, slew0)
    Xo, Uo = co.lcone_direct_py(*args, Nc=Nc, smooth_alpha=10.0, smooth_cstr="squareplus", smooth_beta=2.0, **kw)
    X, U = backend.lcone_solve(*abi_args(args, kw, Nc), smooth_alpha=10.0, solver="ecos", smooth_cstr="squareplus", smooth_beta=2.0)
    assert np.all(np.isfinite(X)) and np.all(np.isfinite(U))
    assert _rel(X, Xo) <= TOL and _rel(U, Uo) <= TOL, (_rel(X, Xo), _rel(U, Uo))


@pytest.mark.parametrize("wgt,slew", [(2.5, None), (0.3, 0.5)])
def test_squareplus_with_one_weighted_particle(co, wgt, slew):
    """`weights` (scale_probs_cost!, main.jl:96-112) scale the particle's cost, not the hinges: (1 - eps) w J + hinges."""
    import torch

    from pmpc_amd.device import DeviceSolver

    args, kw = rand_problem(np.random.default_rng(85), 1, 6, 4, 2, 0.4, None, slew, None)
    x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
    Xo, Uo = co.lcone_direct_py(x0, f, fx, fu, X_prev, U_prev, wgt * Q, wgt * R, X_ref, U_ref, Nc=1, smooth_alpha=10.0, smooth_cstr="squareplus", smooth_beta=2.0,
                                **{k: (wgt * v if k in ("reg_x", "reg_u", "slew_reg") else v) for k, v in kw.items()})
    dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda")
    T = lambda a: dev(np.swapaxes(a, -1, -2))
    s = DeviceSolver(0)
    skw = dict(slew_reg=dev(kw["slew_reg"])) if slew is not None else {}
    X, U, st = s.lcone_solve(smooth_alpha=10.0, smooth_cstr="squareplus", smooth_beta=2.0, f=dev(f), fx=T(fx), fu=T(fu), X_prev=dev(X_prev), U_prev=dev(U_prev), Q=T(Q),
                             R=T(R), X_ref=dev(X_ref), U_ref=dev(U_ref), reg_x=kw["reg_x"], reg_u=kw["reg_u"], Nc=1, x0=dev(x0), lu=dev(kw["u_l"]), uu=dev(kw["u_u"]),
                             symmetric_cost=True, weights=dev(np.array([wgt])), **skw)
    s.sync()
    assert st == 0
    assert _rel(X.cpu().numpy(), Xo) <= TOL and _rel(U.cpu().numpy(), Uo) <= TOL
    s.close()


def _device_cone_solve(args, kw, Nc, options, repeats=1):
    """One context with the given options, `repeats` solves of the same problem (the later ones see what the context learnt); no assertion on the status."""
    import torch

    from pmpc_amd.device import DeviceSolver

    x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
    dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda")
    T = lambda a: dev(np.swapaxes(a, -1, -2))
    s = DeviceSolver(0)
    for key, val in options.items():
        s.set_option(key, val)
    out = []
    for _ in range(repeats):
        X, U, status = s.lcone_solve(f=dev(f), fx=T(fx), fu=T(fu), X_prev=dev(X_prev), U_prev=dev(U_prev), Q=T(Q), R=T(R), X_ref=dev(X_ref), U_ref=dev(U_ref),
                                     reg_x=kw["reg_x"], reg_u=kw["reg_u"], Nc=Nc, symmetric_cost=True, lu=dev(kw["u_l"]), uu=dev(kw["u_u"]))
        s.sync()
        out.append((X.cpu().numpy(), U.cpu().numpy(), status, dict(s.last_info)))
    s.close()
    return out


@pytest.mark.parametrize("copies,others,Nc,bu,seed", [(2, 4, 1, 0.4, 1231), (5, 3, 1, 0.4, 507), (4, 3, 2, 0.4, 407), (4, 3, -1, 5.0, 1250), (2, 5, -1, 5.0, 1232)])
def test_every_body_of_the_cone_path_forced_on_the_same_problem(co, copies, others, Nc, bu, seed):
    """Which body of `lcone_body` answers a hard-box call depends, by default, on what the context learnt about the shape (`fp_key`): option
    `cone_path` takes that memory out — 1 free-particles body first, 2 epigraph path, 3 the rank-based iteration alone.  The same problem
    through each, twice on one context: every body that answers (status 0) gives the direct program's optimum; a body that cannot (the
    rank-based iteration with more than two costs on the threshold) says so with a non-zero status and NaN outputs, never with another
    answer; the automatic order and the forced ones agree."""
    args, kw = tied_problem(np.random.default_rng(seed), copies, others, 6, 4, 2, bu, Nc)
    Xo, Uo = co.lcone_direct_py(*args, Nc=Nc, **kw)
    answered = {}
    for path in (0, 1, 2, 3):
        for rep, (X, U, status, info) in enumerate(_device_cone_solve(args, kw, Nc, {"cone_path": path}, repeats=2)):
            if status == 0:
                assert _rel(X, Xo) <= TOL and _rel(U, Uo) <= TOL, (path, rep, _rel(X, Xo), _rel(U, Uo))
                answered[path] = answered.get(path, 0) + 1
            else:
                assert np.all(np.isnan(X)) and np.all(np.isnan(U)), (path, rep, status)
    assert answered.get(0) == 2 and answered.get(1) == 2 and answered.get(2) == 2, answered  # (only the rank-based iteration alone may decline)
    if copies <= 2:
        assert answered.get(3) == 2, answered
