"""General stage-local conic rows of the reference's `extra_cstrs` mechanism (PMPC.jl/src/main.jl:293-316; tuple format of
README.md:219-239) on the GPU — linear rows coupling several controls of one stage (a stage-wise polytope), several
second-order cones per stage, stage-dependent data, with and without control boxes and consensus stages — against the sparse
joint-KKT conic oracle (`oracle.lconic_solve_py`: log-barrier path following over the joint variable vector, independent of the
device code).  Tolerance: fp64, 1e-7 relative (north star: 1e-6)."""
import numpy as np
import pytest
import scipy.sparse as sp

from tests.support.problems import rand_problem

pytestmark = pytest.mark.gpu
TOL = 1e-7


def _rel(a, b, floor=1e-300):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), floor)


def make_tuples(rng, M, N, x, u, Nc, kinds):
    """`extra_cstrs` tuples over z = [U_cons; U_free; X] that keep u = 0 strictly feasible.  kinds: list of "lin2" (two linear rows
    coupling all controls of the stage), "soc" (a cone |B u + b| <= a'u + a0 with stage-dependent data)."""
    Ncc = N if Nc < 0 else Nc
    nblocks = Ncc + M * (N - Ncc)
    ncu = nblocks * u
    n = ncu + M * N * x
    tuples = []
    for kind in kinds:
        rows, cols, vals, h, q = [], [], [], [], []
        r = 0
        for b in range(nblocks):
            if kind == "lin2":
                for _ in range(2):
                    a = rng.standard_normal(u)
                    for k in range(u):
                        rows.append(r); cols.append(b * u + k); vals.append(a[k])
                    h.append(0.15 + 0.2 * rng.random())  # a'u <= h, u = 0 strictly inside
                    r += 1
            else:
                qk = min(3, u)  # rows of the cone: 1 + (qk - 1)
                a = 0.3 * rng.standard_normal(u)
                B = rng.standard_normal((qk - 1, u))
                for k in range(u):
                    rows.append(r); cols.append(b * u + k); vals.append(a[k])
                h.append(-(0.2 + 0.2 * rng.random()))  # row 0: a'u - h0 = a'u + a0
                for t in range(qk - 1):
                    for k in range(u):
                        rows.append(r + 1 + t); cols.append(b * u + k); vals.append(B[t, k])
                    h.append(0.02 * rng.standard_normal())
                q.append(qk)
                r += qk
        G = sp.csr_matrix((vals, (rows, cols)), shape=(r, n))
        l = r if kind == "lin2" else 0
        tuples.append((l, q, 0, G, sp.csr_matrix((r, 0)), np.array(h), np.zeros(n), np.zeros(0)))
    return tuples, ncu


def oracle_solve(oracle, args, kw, Nc, tuples, ncu, weights=None):
    lin_G, lin_h, socs = [], [], []
    for (l, q, e, G, Gr, h, cl, cr) in tuples:
        G = sp.csr_matrix(G)
        if l:
            lin_G.append(G[:l]); lin_h.append(h[:l])
        r = l
        for qk in q:
            socs.append((G[r:r + qk], h[r:r + qk]))
            r += qk
    lin = (sp.vstack(lin_G), np.concatenate(lin_h)) if lin_G else None
    return oracle.lconic_solve_py(*args, Nc=Nc, reg_x=kw["reg_x"], reg_u=kw["reg_u"], u_l=kw.get("u_l"), u_u=kw.get("u_u"), lin=lin, socs=socs,
                                  z0=np.zeros(ncu), weights=weights)


# (M, N, x, u, Nc, u-bound, kinds).  The rounds have no globalisation: several constraints of one stage that are active together
# with tight boxes on the same controls can make the set cycle, which ends as a FAILED solve (NaN outputs), never as a wrong answer
# (DESIGN.md); the cases below settle.
CASES = [
    (4, 8, 4, 2, 0, 0.6, ["lin2"]),
    (4, 6, 5, 3, 1, 2.0, ["soc"]),
    (4, 6, 5, 3, 2, 2.0, ["lin2"]),
    (4, 6, 5, 3, -1, None, ["soc"]),
    (4, 6, 5, 3, 1, None, ["soc", "lin2"]),
    (4, 6, 5, 3, -1, 2.0, ["soc", "lin2"]),
    (5, 7, 12, 4, 1, 2.0, ["lin2", "soc"]),
    (6, 9, 8, 4, 2, 2.0, ["soc", "soc"]),
    (3, 8, 12, 4, 0, None, ["lin2", "lin2"]),
]


@pytest.mark.parametrize("case", CASES, ids=str)
def test_general_stage_cones_through_the_host_path_match_the_conic_oracle(case, oracle):
    """`backend.aff_solve(..., solver_settings=dict(extra_cstrs=[tuples]))`: the tuples are recognised as stage-local rows / cones
    (pmpc_amd/extra_cstrs.py), handed to `pmpc_lsoc_solve_device` in its general form and solved by the active-set rounds."""
    from pmpc_amd import backend

    M, N, x, u, Nc, bu, kinds = case
    rng = np.random.default_rng(8200 + CASES.index(case))
    args, kw = rand_problem(rng, M, N, x, u, bu)
    tuples, ncu = make_tuples(rng, M, N, x, u, Nc, kinds)
    Xo, Uo = oracle_solve(oracle, args, kw, Nc, tuples, ncu)
    x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
    X, U, _ = backend.aff_solve(f, fx, fu, x0, X_prev, U_prev, Q, R, X_ref, U_ref, kw["reg_x"], kw["reg_u"], None, None, None, None,
                                kw.get("u_l"), kw.get("u_u"), solver_settings=dict(solver="osqp", Nc=Nc, extra_cstrs=tuples))
    assert not np.isnan(U).any(), "solver failed"
    assert _rel(X[:, 1:], Xo) < TOL and _rel(U, Uo, 1.0) < TOL, (_rel(X[:, 1:], Xo), _rel(U, Uo, 1.0))
    # some row / cone is active (else the test says nothing about them)
    act = 0
    for (l, q, e, G, Gr, h, cl, cr) in tuples:
        z = np.concatenate([U[0, :(N if Nc < 0 else Nc)].reshape(-1), U[:, (N if Nc < 0 else Nc):].reshape(-1)])
        s = sp.csr_matrix(G)[:, :ncu] @ z - h
        if l:
            act += int((s[:l] > -1e-7).sum())
        r = l
        for qk in q:
            act += int(s[r] - np.linalg.norm(s[r + 1:r + qk]) < 1e-7)
            r += qk
    assert act > 0


def test_weights_and_warm_start_with_general_cones(oracle):
    """Device API: per-particle cost weights next to the cones, and a second solve of a perturbed problem warm-started from the
    first one's set and multipliers."""
    import torch

    from pmpc_amd.device import DeviceSolver
    from pmpc_amd.extra_cstrs import stage_cones_from_extra_cstrs

    M, N, x, u, Nc = 5, 8, 6, 3, 1
    rng = np.random.default_rng(77)
    args, kw = rand_problem(rng, M, N, x, u, 2.0)
    tuples, ncu = make_tuples(rng, M, N, x, u, Nc, ["lin2", "soc"])
    wts = 0.3 + rng.random(M)
    cn = stage_cones_from_extra_cstrs(tuples, M, N, x, u, Nc)
    assert cn["sizes"] == [0, 0, 2] and cn["A"].shape == (M, N, 5, u)
    dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda")
    T = lambda a: dev(np.swapaxes(a, -1, -2))
    s = DeviceSolver(0)
    rounds = []
    for rep in range(2):
        if rep == 1:
            args = tuple(a + 0.01 * rng.standard_normal(a.shape) if k in (1, 4, 5) else a for k, a in enumerate(args))
        Xo, Uo = oracle_solve(oracle, args, kw, Nc, tuples, ncu, weights=wts)
        x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
        X, U, status = s.lsoc_solve(f=dev(f), fx=T(fx), fu=T(fu), X_prev=dev(X_prev), U_prev=dev(U_prev), Q=T(Q), R=T(R), X_ref=dev(X_ref),
                                    U_ref=dev(U_ref), reg_x=kw["reg_x"], reg_u=kw["reg_u"], Nc=Nc, symmetric_cost=True, lu=dev(kw["u_l"]),
                                    uu=dev(kw["u_u"]), weights=dev(wts), cones=dict(sizes=cn["sizes"], A=dev(cn["A"]), c=dev(cn["c"])))
        s.sync()
        assert status == 0
        assert _rel(X.cpu().numpy(), Xo) < TOL and _rel(U.cpu().numpy(), Uo, 1.0) < TOL
        rounds.append(s.last_info["active_set_rounds"])
        assert s.last_info["ipm_iters"] == 0
    assert rounds[1] <= rounds[0]
    s.close()
