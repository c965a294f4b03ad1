"""Linear rows of the reference's `extra_cstrs` that involve STATES (PMPC.jl/src/main.jl:293-316, rows `G z <= h` over
z = [U_cons; U_free; X], cone_solver.jl:163-166): obstacle half-spaces a'x_(t+1) <= h, rows coupling the state and the control of one
stage.  The host side restates each as an upper bound on an auxiliary state the dynamics produce (`pmpc_amd.extra_cstrs.
aux_state_problem`), the device solver holds it with its state-box rows.  Checked against the oracle's joint problem with the rows
as rows: the plain-sum QP (`solver = "osqp"`), the reference's cone program with the rows handed to `augment_cone_problem!`
(`solver = "ecos"`, hard boxes and log-barrier smoothing).  fp64, 1e-7 relative (north star 1e-6)."""
import numpy as np
import pytest
import scipy.sparse as sp

from tests.support.problems import rand_problem

pytestmark = pytest.mark.gpu
TOL = 1e-7


def _rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1.0)


def make_rows(rng, oracle, args, kw, Nc, spec, cut=0.25):
    """One tuple with a row per `spec` entry (particle, stage, form): form 0 couples X[i, t] with U[i, t], form 1 X[i, t-1] with U[i, t],
    "x" the state alone, "u" two controls of the stage alone.  Each row cuts `cut` off the value the unconstrained-in-the-rows optimum
    takes, so every row binds or nearly so."""
    x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
    M, N, x = f.shape
    u = fu.shape[-1]
    Ncc = N if Nc < 0 else Nc
    X0, U0 = oracle.lqp_solve_py(*args, Nc=Nc, **kw)
    ncu = Ncc * u + M * (N - Ncc) * u
    n = ncu + M * N * x
    xcol = lambda i, j, r: ncu + (i * N + j) * x + r
    ucol = lambda i, j, r: j * u + r if j < Ncc else Ncc * u + (i * (N - Ncc) + (j - Ncc)) * u + r
    G, h = np.zeros((len(spec), n)), np.zeros(len(spec))
    for k, (i, t, form) in enumerate(spec):
        a, b = rng.standard_normal(x), rng.standard_normal(u)
        if form == "x":
            b[:] = 0.0
        if form == "u":
            a[:] = 0.0
        jx = t - 1 if form == 1 else t
        for r in range(x):
            G[k, xcol(i, jx, r)] = a[r]
        for r in range(u):
            G[k, ucol(i, t, r)] = b[r]
        h[k] = a @ X0[i, jx] + b @ U0[i, t] - cut
    Gs = sp.csr_matrix(G)
    return (len(spec), [], 0, Gs, sp.csr_matrix((len(spec), 0)), h, np.zeros(n), np.zeros(0)), (Gs, h)


def solve_host(args, kw, Nc, tuples, **settings):
    from pmpc_amd import backend

    x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
    X, U, _ = backend.aff_solve(f, fx, fu, x0, X_prev, U_prev, Q, R, X_ref, U_ref, kw["reg_x"], kw["reg_u"], None, None, kw.get("x_l"), kw.get("x_u"),
                                kw.get("u_l"), kw.get("u_u"), solver_settings=dict(Nc=Nc, extra_cstrs=tuples, **settings))
    assert not np.isnan(U).any(), "solver failed"
    assert X.shape[-1] == f.shape[-1]
    return X[:, 1:], U


def z_of(X, U, Nc):
    Ncc = U.shape[1] if Nc < 0 else Nc
    return np.concatenate([U[0, :Ncc].reshape(-1), U[:, Ncc:].reshape(-1), X.reshape(-1)])


#        M, N, x, u, Nc, bu, spec
CASES = [
    (3, 6, 4, 2, 1, 1.0, [(0, 2, 0), (1, 3, 1), (2, 0, 0), (0, 2, 0), (1, 5, "x")]),      # two rows on one (particle, stage): two auxiliary states
    (4, 8, 4, 2, 2, 0.8, [(i, t, "x") for i in range(4) for t in (3, 6)]),                  # half-spaces on the states
    (2, 5, 3, 2, -1, None, [(0, 1, 0), (1, 4, 1), (0, 3, "u")]),                            # every control shared, no boxes; a control-only row rides along
    (3, 6, 12, 4, 1, 2.0, [(0, 2, "x"), (1, 4, 0), (2, 5, 1)]),                             # quadrotor-sized (13 states: the generic kernels)
    (5, 7, 5, 3, 0, 1.0, [(i, 3, 0) for i in range(5)]),                                    # no consensus
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"M{c[0]}N{c[1]}x{c[2]}u{c[3]}Nc{c[4]}")
def test_state_rows_on_the_qp_path_match_the_joint_qp_with_rows(case, oracle):
    M, N, x, u, Nc, bu, spec = case
    rng = np.random.default_rng(9100 + CASES.index(case))
    args, kw = rand_problem(rng, M, N, x, u, bu)
    tup, rows = make_rows(rng, oracle, args, kw, Nc, spec)
    Xo, Uo = oracle.lqp_solve_py(*args, Nc=Nc, rows=rows, **kw)
    X0, _ = oracle.lqp_solve_py(*args, Nc=Nc, **kw)
    assert _rel(Xo, X0) > 1e-3  # the rows move the answer
    X, U = solve_host(args, kw, Nc, [tup], solver="osqp")
    assert _rel(X, Xo) < TOL and _rel(U, Uo) < TOL, (_rel(X, Xo), _rel(U, Uo))
    assert np.max(rows[0] @ z_of(X, U, Nc) - rows[1]) < 1e-8
    assert np.sum(rows[0] @ z_of(X, U, Nc) - rows[1] > -1e-8) > 0  # some row binds


@pytest.mark.parametrize("case", [CASES[0], CASES[1], CASES[4]], ids=lambda c: f"M{c[0]}N{c[1]}x{c[2]}u{c[3]}Nc{c[4]}")
@pytest.mark.parametrize("alpha", [None, 8.0])
def test_state_rows_in_the_cone_program_match_the_restated_reference_program(case, alpha, oracle):
    """The reference's default solver path: the eps-anchored objective (main.jl:204-239) with the rows through augment_cone_problem!
    (main.jl:293-316) — as hard rows, and smoothed with the boxes (log barrier 1/alpha on each row, main.jl:298-312)."""
    from oracle import cone_oracle as co

    M, N, x, u, Nc, bu, spec = case
    rng = np.random.default_rng(9200 + CASES.index(case))
    args, kw = rand_problem(rng, M, N, x, u, bu)
    tup, rows = make_rows(rng, oracle, args, kw, Nc, spec)
    skw = {} if alpha is None else dict(smooth_alpha=alpha)
    Xo, Uo = co.lcone_direct_py(*args, Nc=Nc, extra_cstrs=[tup], **skw, **kw)
    X, U = solve_host(args, kw, Nc, [tup], solver="ecos", **skw)
    tol = 1e-6
    assert _rel(X, Xo) < tol and _rel(U, Uo) < tol, (_rel(X, Xo), _rel(U, Uo))
    assert np.max(rows[0] @ z_of(X, U, Nc) - rows[1]) < 1e-8


def test_rows_outside_the_supported_structure_are_refused_with_the_reason(oracle):
    from pmpc_amd import backend

    M, N, x, u, Nc = 2, 4, 3, 2, 1
    rng = np.random.default_rng(9300)
    args, kw = rand_problem(rng, M, N, x, u, 1.0)
    ncu = Nc * u + M * (N - Nc) * u
    n = ncu + M * N * x
    two_particles = np.zeros((1, n)); two_particles[0, ncu + 0] = 1.0; two_particles[0, ncu + N * x] = 1.0
    two_stages = np.zeros((1, n)); two_stages[0, ncu + 0] = 1.0; two_stages[0, ncu + 2 * x] = 1.0
    x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
    for G, why in ((two_particles, "several particles"), (two_stages, "several stages")):
        tup = (1, [], 0, sp.csr_matrix(G), sp.csr_matrix((1, 0)), np.ones(1), np.zeros(n), np.zeros(0))
        with pytest.raises(ValueError, match=why):
            backend.aff_solve(f, fx, fu, x0, X_prev, U_prev, Q, R, X_ref, U_ref, 1.0, 0.1, None, None, None, None, kw["u_l"], kw["u_u"],
                              solver_settings=dict(solver="osqp", Nc=Nc, extra_cstrs=[tup]))


@pytest.mark.parametrize("solver", ["osqp", "ecos"])
def test_host_loop_avoids_an_obstacle_through_extra_cstrs_fns(solver):
    """The use the reference documents for `extra_cstrs_fns` (README.md:219-239): constraints re-linearised about the previous iterate in
    every SCP iteration — here a disc obstacle, convexified to the half-space  n_k'(p_k - c) >= r,  n_k the direction from the centre to
    the previous iterate's position.  A double integrator (the dynamics are exact, the obstacle is the only non-convexity), 3 particles
    with different starts sharing their first control.  The converged trajectories stay outside the disc; without the rows they cross it."""
    import pmpc_amd

    M, N, dt = 3, 20, 0.2
    A = np.eye(4); A[0, 2] = A[1, 3] = dt
    B = np.zeros((4, 2)); B[0, 0] = B[1, 1] = 0.5 * dt * dt; B[2, 0] = B[3, 1] = dt

    def f_fx_fu(X, U):
        return X @ A.T + U @ B.T, np.broadcast_to(A, X.shape[:-1] + (4, 4)).copy(), np.broadcast_to(B, X.shape[:-1] + (4, 2)).copy()

    x0 = np.array([[2.0, 2.0, 0, 0], [2.0, 1.8, 0, 0], [1.8, 2.0, 0, 0]])
    ctr, rad = np.array([1.0, 1.05]), 0.45
    Q = np.tile(np.diag([1.0, 1.0, 0.1, 0.1]), (M, N, 1, 1))
    R = np.tile(0.05 * np.eye(2), (M, N, 1, 1))
    Nc = 1
    ncu = Nc * 2 + M * (N - Nc) * 2
    n = ncu + M * N * 4

    def cstrs(X_prev, U_prev, problems):
        P = X_prev[..., :2] - ctr
        nrm = P / np.maximum(np.linalg.norm(P, axis=-1, keepdims=True), 1e-9)
        rows, cols, vals, h = [], [], [], []
        r = 0
        for i in range(M):
            for k in range(N):
                for d in range(2):  # -n'p <= -r - n'c
                    rows.append(r); cols.append(ncu + (i * N + k) * 4 + d); vals.append(-nrm[i, k, d])
                h.append(-rad - nrm[i, k] @ ctr)
                r += 1
        return [(r, [], 0, sp.csr_matrix((vals, (rows, cols)), shape=(r, n)), sp.csr_matrix((r, 0)), np.array(h), np.zeros(n), np.zeros(0))]

    # initial guess: straight lines to the origin bowed to one side of the disc
    s = np.linspace(0, 1, N + 1)[1:]
    Xg = np.zeros((M, N, 4))
    Xg[..., :2] = x0[:, None, :2] * (1 - s)[None, :, None] + 0.6 * np.sin(np.pi * s)[None, :, None] * np.array([1.0, -1.0])
    common = dict(X_ref=np.zeros((M, N, 4)), U_ref=np.zeros((M, N, 2)), X_prev=Xg, U_prev=np.zeros((M, N, 2)), u_l=-3 * np.ones((M, N, 2)), u_u=3 * np.ones((M, N, 2)),
                  max_it=30, reg_x=1.0, reg_u=1.0, res_tol=1e-7, verbose=False)
    Xf, Uf, _ = pmpc_amd.solve(f_fx_fu, Q, R, x0, solver_settings=dict(solver=solver, Nc=Nc), **common)
    X, U, data = pmpc_amd.solve(f_fx_fu, Q, R, x0, solver_settings=dict(solver=solver, Nc=Nc), extra_cstrs_fns=cstrs, **common)
    assert X is not None and X.shape == (M, N + 1, 4)
    dist_free = np.linalg.norm(Xf[:, 1:, :2] - ctr, axis=-1).min()
    dist = np.linalg.norm(X[:, 1:, :2] - ctr, axis=-1).min()
    assert dist_free < rad - 0.05, dist_free          # the unconstrained paths cross the disc
    assert dist > rad - 1e-6, dist                    # the constrained ones do not (a half-space lies outside the disc)
    assert dist < rad + 1e-3                          # and they graze it
    assert np.all(U[:, 0] == U[0:1, 0])               # the shared control is shared


@pytest.mark.parametrize("smooth", [None, ("logbarrier", 8.0), ("squareplus", 8.0)])
def test_linear_cost_term_c_left_with_one_particle(smooth, oracle):
    """`c_left` (cone_utils.jl:152-154) is added to the cone program's cost vector outside the epigraph rows.  One particle: the program
    is min (1 - eps) J + c'z (+ smoothing) — the references shifted by Q^-1 c / (1 - eps), as the reference itself folds linear cost
    terms (pmpc/scp_mpc.py:171-185).  A tuple may carry the cost alone (no rows)."""
    from oracle import cone_oracle as co

    N, x, u, Nc = 6, 4, 2, 1
    rng = np.random.default_rng(9500)
    args, kw = rand_problem(rng, 1, N, x, u, 0.8)
    n = N * u + N * x
    c = 0.5 * rng.standard_normal(n)
    tup = (0, [], 0, sp.csr_matrix((0, n)), sp.csr_matrix((0, 0)), np.zeros(0), c, np.zeros(0))
    skw = {} if smooth is None else dict(smooth_cstr=smooth[0], smooth_alpha=smooth[1])
    Xo, Uo = co.lcone_direct_py(*args, Nc=Nc, extra_cstrs=[tup], **skw, **kw)
    X0, U0 = co.lcone_direct_py(*args, Nc=Nc, **skw, **kw)
    assert _rel(Xo, X0) > 1e-3  # the term moves the answer
    X, U = solve_host(args, kw, Nc, [tup], solver="ecos", **skw)
    assert _rel(X, Xo) < 1e-6 and _rel(U, Uo) < 1e-6, (_rel(X, Xo), _rel(U, Uo))


def test_cost_terms_and_cones_outside_the_supported_cases_are_refused_with_the_reason(oracle):
    from oracle import cone_oracle as co
    from pmpc_amd import backend

    M, N, x, u, Nc = 2, 4, 3, 2, 1
    rng = np.random.default_rng(9600)
    args, kw = rand_problem(rng, M, N, x, u, 1.0)
    x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
    ncu = Nc * u + M * (N - Nc) * u
    n = ncu + M * N * x
    call = lambda tup, **st: backend.aff_solve(f, fx, fu, x0, X_prev, U_prev, Q, R, X_ref, U_ref, 1.0, 0.1, None, None, None, None, kw["u_l"], kw["u_u"],
                                               solver_settings=dict(Nc=Nc, extra_cstrs=[tup], solver="ecos", **st))
    cost = (0, [], 0, sp.csr_matrix((0, n)), sp.csr_matrix((0, 0)), np.zeros(0), np.ones(n), np.zeros(0))
    with pytest.raises(ValueError, match="one particle"):
        call(cost)
    row = np.zeros((1, n)); row[0, 0] = 1.0
    Gl, Gr, hh = co.smoothen_linear_inequalities_py(sp.csr_matrix(row), np.ones(1), 5.0)
    etup = (0, [], 1, Gl, Gr, hh, np.zeros(n), np.ones(1))
    for st in ({}, dict(smooth_alpha=5.0), dict(smooth_alpha=5.0, smooth_cstr="squareplus")):
        with pytest.raises(ValueError, match="exponential cones are not supported"):
            call(etup, **st)
