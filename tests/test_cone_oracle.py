"""CPU tests of oracle/cone_oracle.py — the DIRECT restatement of the reference's cone program (PMPC.jl/src/main.jl:204-316,
cone_utils.jl:25-232) — against (a) the identities the reference's own comments state, (b) the derived oracle
(oracle/lqp_oracle.py: lcone_solve_py eliminates (y, t) and searches the threshold particle), which it must agree with wherever
that one applies, and (c) rows the reference's own Julia + ECOS stack printed (tests/golden/ref_*.npz)."""
import numpy as np
import pytest

from tests.support.problems import rand_problem


@pytest.fixture(scope="module")
def co():
    from oracle import cone_oracle, lqp_oracle

    lqp_oracle.build()
    return cone_oracle


@pytest.fixture(scope="module")
def orc():
    from oracle import lqp_oracle

    lqp_oracle.build()
    return lqp_oracle


def test_Pqr2Gh_rows_are_the_quadratic_epigraph(co):
    """cone_utils.jl:26-33: || (tau - bet ; L z - b) || <= tau - alf  <=>  1/2 z'Pz + q'z + r <= tau."""
    rng = np.random.default_rng(0)
    n = 7
    B = rng.standard_normal((n, n))
    P, q, r = B @ B.T + n * np.eye(n), rng.standard_normal(n), 0.37
    G_left, G_right, h = co.Pqr2Gh_py(P, q, r)
    assert G_left.shape == (n + 2, n) and G_right.shape == (n + 2, 1) and np.all(G_right[:2, 0] == 1.0) and np.all(G_right[2:] == 0.0)
    for _ in range(20):
        z = rng.standard_normal(n)
        Jz = 0.5 * z @ P @ z + q @ z + r
        for tau in (Jz - 0.3, Jz + 1e-9, Jz + 2.0):
            s = G_left @ z + G_right[:, 0] * tau - h
            d = s[0] ** 2 - s[1:] @ s[1:]
            assert abs(d - (tau - Jz)) <= 1e-9 * max(1.0, abs(s[0]) ** 2)  # s0^2 - |s_tail|^2 == tau - J(z)
            assert (s[0] >= np.linalg.norm(s[1:])) == (tau >= Jz)


def test_particle_quadratic_is_the_tracking_cost(co, orc):
    """qp_utils.jl:60-162: 1/2 z'Pz + q'z + resid is the particle's cost as the device and the derived oracle evaluate it
    (slew constants absent upstream, :140-160)."""
    rng = np.random.default_rng(1)
    args, kw = rand_problem(rng, 2, 5, 4, 2, 0.4)
    x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
    X, U = rng.standard_normal(X_prev.shape), rng.standard_normal(U_prev.shape)
    for slew in (dict(), dict(slew_reg=0.7), dict(slew_reg=0.7, slew_reg0=0.4, slew_um1=rng.standard_normal((2, 2)))):
        Jd = orc.particle_costs_py(X, U, X_prev, U_prev, Q, R, X_ref, U_ref, reg_x=kw["reg_x"], reg_u=kw["reg_u"], **slew)
        for i in range(2):
            um1 = slew.get("slew_um1")
            P, q, r = co.qp_repr_Pq_py(Q[i], R[i], X_prev[i], U_prev[i], X_ref[i], U_ref[i], kw["reg_x"], kw["reg_u"], slew.get("slew_reg", 0.0),
                                       slew.get("slew_reg0", 0.0), None if um1 is None else um1[i])
            z = np.concatenate([U[i].ravel(), X[i].ravel()])
            assert abs(0.5 * z @ P @ z + q @ z + r - Jd[i]) <= 1e-10 * max(1.0, abs(Jd[i]))


@pytest.mark.parametrize("smooth", ["", "logbarrier", "squareplus"])
def test_problem_shape_follows_main_jl(co, smooth):
    """Row classes and variable counts of main.jl:216-291: M y-rows (+ hard box rows) | M epigraph cones (+ 3-row cones per box side
    for squareplus) | 3 rows per box side for logbarrier; one new variable per smoothed row; cost (1+eps) on y, (1-eps) k on t."""
    rng = np.random.default_rng(2)
    M, N, x, u, Nc = 3, 4, 4, 2, 1
    args, kw = rand_problem(rng, M, N, x, u, 0.4)
    alpha = float("nan") if smooth == "" else 10.0
    prob = co.lcone_problem_py(*args, Nc=Nc, smooth_cstr=smooth, smooth_alpha=alpha, **kw)
    nbox = 2 * (Nc + M * (N - Nc)) * u  # both sides of every control variable (shared ones once)
    nz = Nc * u + M * ((N - Nc) * u + N * x)
    assert prob.nz == nz
    if smooth == "":
        assert (prob.l, prob.q, prob.e) == (M + nbox, [2 + N * (x + u)] * M, 0) and prob.c.size == nz + M + 1
    elif smooth == "logbarrier":
        assert (prob.l, prob.q, prob.e) == (M, [2 + N * (x + u)] * M, nbox) and prob.c.size == nz + M + 1 + nbox
        assert np.all(prob.c[nz + M + 1:] == 1.0)
    else:
        assert (prob.l, prob.q, prob.e) == (M, [2 + N * (x + u)] * M + [3] * nbox, 0) and prob.c.size == nz + M + 1 + nbox
    assert np.allclose(prob.c[nz:nz + M], 1 + co.COST_ANCHOR_EPS) and np.isclose(prob.c[nz + M], (1 - co.COST_ANCHOR_EPS) * M)
    assert prob.G.shape == (prob.l + sum(prob.q) + 3 * prob.e, prob.c.size) and prob.A.shape[1] == prob.c.size


@pytest.mark.parametrize("alpha", [float("nan"), 1e2, 1.0])
@pytest.mark.parametrize("Nc", [1, -1])
def test_direct_and_derived_oracles_agree(co, orc, alpha, Nc):
    rng = np.random.default_rng(3)
    args, kw = rand_problem(rng, 5, 6, 4, 2, 0.4)
    Xo, Uo = orc.lcone_solve_py(*args, Nc=Nc, smooth_alpha=alpha, **kw)
    X, U, info = co.lcone_direct_py(*args, Nc=Nc, smooth_alpha=alpha, return_info=True, **kw)
    assert np.abs(X - Xo).max() <= 2e-8 * max(1.0, np.abs(Xo).max()) and np.abs(U - Uo).max() <= 2e-8, info


def tied_problem(rng, copies, others, N=6, x=4, u=2, bu=0.4):
    """`copies` identical particles whose cost is the LOWEST + `others` different, costlier ones: at the optimum of the epigraph problem
    the identical ones sit together on the threshold t (a `copies`-way tie; M < 500: t = min J), whatever their multipliers do."""
    from oracle import lqp_oracle as orc

    args, kw = rand_problem(rng, copies + others, N, x, u, bu)
    X, U = orc.lqp_solve_py(*args, Nc=1, **kw)
    x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
    J = orc.particle_costs_py(X, U, X_prev, U_prev, Q, R, X_ref, U_ref, reg_x=kw["reg_x"], reg_u=kw["reg_u"])
    order = np.argsort(J)  # cheapest first: it becomes particles 0 .. copies-1, the costliest `others` follow
    keep = np.concatenate([[order[0]] * copies, order[-others:]])
    args = tuple(np.array(np.asarray(a)[keep], copy=True) for a in args)
    kw = {k: (np.array(np.asarray(v)[keep], copy=True) if isinstance(v, np.ndarray) and v.ndim and v.shape[0] == copies + others else v) for k, v in kw.items()}
    return args, kw


@pytest.mark.parametrize("copies,alpha", [(3, float("nan")), (5, float("nan")), (3, 10.0)])
def test_ties_among_identical_particles(co, copies, alpha):
    """More than two particle costs on the kink: the case the derived oracle's pair search does not cover.  Identical particles
    must come out with identical trajectories, and every one of them on the threshold (hard boxes: J_i = t exactly)."""
    rng = np.random.default_rng(10 + copies)
    args, kw = tied_problem(rng, copies, 3)
    X, U, info = co.lcone_direct_py(*args, Nc=1, smooth_alpha=alpha, return_info=True, **kw)
    for i in range(1, copies):
        assert np.abs(X[i] - X[0]).max() <= 1e-7 and np.abs(U[i] - U[0]).max() <= 1e-7
    from oracle import lqp_oracle

    x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
    J = lqp_oracle.particle_costs_py(X, U, X_prev, U_prev, Q, R, X_ref, U_ref, reg_x=kw["reg_x"], reg_u=kw["reg_u"])
    assert np.ptp(J[:copies]) <= 1e-7 * max(1.0, abs(J[0]))
    assert abs(J[0] - info["t"]) <= 1e-5 * max(1.0, abs(J[0])) and np.all(J[copies:] >= info["t"] - 1e-6)


def _direct_aff_solve(co, rows):
    def aff(f, fx, fu, x0, X_prev, U_prev, Q, R, X_ref, U_ref, reg_x, reg_u, slew_rate, u_slew, x_l, x_u, u_l, u_u, solver_settings=None, **_):
        s = solver_settings or {}
        rows.append(1)
        X, U = co.lcone_direct_py(x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref, reg_x=reg_x, reg_u=reg_u, Nc=s.get("Nc", -1), u_l=u_l, u_u=u_u,
                                  slew_reg=slew_rate if slew_rate else None, smooth_alpha=s.get("smooth_alpha", float("nan")), mu_final=1e-10)
        return np.concatenate([x0[:, None, :], X], 1), U, dict()

    return aff


@pytest.mark.parametrize("name,nrows", [("ref_root_testing_single", 8), ("ref_root_testing_consensus", 1)])
def test_direct_oracle_reproduces_rows_the_reference_printed(co, monkeypatch, name, nrows):
    """PIN: the first rows of two tables the reference's own Julia + ECOS stack printed (tests/root_testing.ipynb: M = 1 with slew and
    log-barrier smoothing; M = 20 consensus, Nc = 5, slew 1e2, smoothing alpha = 1), through the reference-shaped program itself —
    Pqr2Gh cones, exponential-cone rows of make_logbarrier_constraint in the "ecos" row order, read in ECOS's own convention.
    (All rows of both tables are pinned through the derived oracle in tests/test_host_logic.py; the conic solve of the M = 20
    program takes a minute per row, so only the leading rows are repeated here.)"""
    import pmpc_amd.scp_mpc as scp
    from tests.support import notebook_problem as nbp

    args, kw, settings, table = nbp.load_table(name)
    kw["max_it"] = nrows
    rows = []
    monkeypatch.setattr(scp, "aff_solve", _direct_aff_solve(co, rows))
    X, U, data = scp.scp_solve(*args, solver_settings=settings, **kw)
    got = np.array([[h["obj"], h["resid"]] for h in data["hist"]])
    want = table[:nrows, 1:3]
    _, obj_tol, res_tol, _ = nbp.TABLES[name]
    np.testing.assert_allclose(got[:, 0], want[:, 0], rtol=obj_tol)
    np.testing.assert_allclose(got[:, 1], want[:, 1], rtol=res_tol)
