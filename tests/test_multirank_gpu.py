"""The world > 1 code paths on ONE GPU: N contexts joined by the library's in-process stand-in communicator
(`pmpc_comm_init_mock`, a test hook — RCCL refuses two ranks on one device), one host thread per rank, particles sharded
in contiguous blocks exactly as bench.py shards them over RCCL.  What runs is everything but RCCL itself: the packed scalar
exchange (pack / all-reduce / unpack kernels), the consensus Hessian / gradient all-reduce and redundant dense solve, the
owner logic (global particle 0 carries the consensus boxes; its bounds are broadcast), warm starts across ranks."""
import threading

import numpy as np
import pytest

from tests.support.problems import rand_problem

pytestmark = pytest.mark.gpu
_group = [1000]


def _solve_sharded(args, kw, Nc, world, repeats=1, cone=False, soc=None, options=None, smooth=None):
    import torch

    from pmpc_amd import _lib
    from pmpc_amd.device import DeviceSolver

    x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
    M = f.shape[0]
    assert M % world == 0
    Ml = M // world
    _group[0] += 1
    group = _group[0]
    out, errs = [None] * world, []

    def run(rank):
        try:
            sl = slice(rank * Ml, (rank + 1) * Ml)
            dev = lambda a: torch.tensor(np.ascontiguousarray(a[sl]), dtype=torch.float64, device="cuda")
            T = lambda a: dev(np.swapaxes(a, -1, -2))
            s = DeviceSolver(0)
            if world > 1:
                assert s.lib.pmpc_comm_init_mock(s.h, rank, world, group) == 0
                s.rank, s.world = rank, world
            for key, val in (options or {}).items():
                s.set_option(key, val)
            opt = dict(smooth) if smooth else {}  # (smooth_alpha / smooth_cstr / smooth_beta of the cone objective)
            if "u_l" in kw:
                opt.update(lu=dev(kw["u_l"]), uu=dev(kw["u_u"]))
            if "x_l" in kw:
                opt.update(lx=dev(kw["x_l"]), ux=dev(kw["x_u"]))
            if "slew_reg" in kw:
                opt.update(slew_reg=dev(kw["slew_reg"]))
            if "slew_reg0" in kw:
                opt.update(slew_reg0=dev(kw["slew_reg0"]), slew_um1=dev(kw["slew_um1"]))
            for rep in range(repeats):  # repeats > 1: the second solve is warm-started on every rank, and reuses the
                opt["static_cons_bounds"] = rep > 0  # consensus bounds broadcast for the first (PMPC_STATIC_CONS_BOUNDS)
                if soc is not None:
                    full = lambda a: torch.tensor(np.asarray(a, dtype=np.float64), device="cuda")
                    opt.update(soc_W=full(soc["W"]), soc_w0=full(soc["w0"]), soc_v=full(soc["v"]), soc_v0=soc["v0"],
                               soc_u_interior=full(soc["u_interior"]))
                X, U, status = (s.lsoc_solve if soc is not None else (s.lcone_solve if cone else s.lqp_solve))(f=dev(f), fx=T(fx), fu=T(fu), X_prev=dev(X_prev), U_prev=dev(U_prev), Q=T(Q), R=T(R),
                                           X_ref=dev(X_ref), U_ref=dev(U_ref), reg_x=kw["reg_x"], reg_u=kw["reg_u"], Nc=Nc,
                                           symmetric_cost=True, **opt)
                s.sync()
            out[rank] = (X.cpu().numpy(), U.cpu().numpy(), status, dict(s.last_info))
            s.close()
        except Exception as e:  # surface failures of a rank thread in the main thread
            errs.append(e)

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errs, errs
    assert all(o is not None for o in out), "a rank did not finish (deadlock in a collective?)"
    assert all(o[2] == 0 for o in out)
    return np.concatenate([o[0] for o in out]), np.concatenate([o[1] for o in out]), [o[3] for o in out]


# (M, N, x, u, Nc, u-bound, x-bound, slew, slew0)
MR_CASES = [
    (8, 9, 12, 4, 1, 0.4, None, None, None),
    (8, 7, 4, 2, 3, 0.3, 6.0, None, None),
    (6, 8, 5, 3, -1, 0.4, None, None, None),
    (8, 6, 3, 2, 2, 0.3, 5.0, 0.5, 0.3),
    (8, 9, 12, 4, 0, 0.4, None, None, None),
    (12, 20, 2, 1, -1, 0.5, None, None, None),
]


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("case", MR_CASES, ids=[str(c) for c in MR_CASES])
def test_sharded_solve_matches_single_rank_and_oracle(case, world, oracle):
    M, N, x, u, Nc, bu, bx, sl, sl0 = case
    if M % world:
        pytest.skip("particles do not divide")
    rng = np.random.default_rng(9000 + MR_CASES.index(case))
    args, kw = rand_problem(rng, M, N, x, u, bu, bx, sl, sl0)
    if Nc != 0 and bu is not None:  # per-particle control boxes: the consensus stages must use GLOBAL particle 0's everywhere
        scale = 1.0 + 0.5 * rng.random((M, 1, 1))
        kw["u_l"], kw["u_u"] = kw["u_l"] * scale, kw["u_u"] * scale
    Xo, Uo = oracle.lqp_solve_py(*args, Nc=Nc, **kw)
    X1, U1, _ = _solve_sharded(args, kw, Nc, 1)
    Xw, Uw, infos = _solve_sharded(args, kw, Nc, world, repeats=2)
    rel = lambda a, b: np.linalg.norm(a - b) / max(np.linalg.norm(b), 1.0)
    assert rel(X1, Xo) < 1e-7 and rel(U1, Uo) < 1e-7
    assert rel(Xw, Xo) < 1e-7 and rel(Uw, Uo) < 1e-7
    k = N if Nc < 0 else Nc
    if k:
        assert np.all(Uw[:, :k] == Uw[0:1, :k])  # the shared controls are bit-identical on every rank
    assert len({(i["ipm_iters"], i["active_set_rounds"], i["structured_solves"]) for i in infos}) == 1  # every rank took the same decisions


@pytest.mark.parametrize("world,Nc,dims", [(2, 1, (8, 10, 4, 2)), (4, 1, (8, 8, 12, 4)), (2, 2, (6, 8, 4, 2))])
def test_sharded_binding_state_boxes_match_single_rank(world, Nc, dims, oracle):
    """State boxes that BIND (rows of the active-set rounds, kernels_xbox.hip) on sharded particles with a consensus horizon: the open
    rows of a round ride in the next round's merged exchange (tail[4]), so a round without status changes is not accepted while a held
    row is still off its bound.  Same answer as one rank to 1e-9, the oracle to 1e-7, cold and warm."""
    from tests.support.problems import xbox_problem

    M, N, x, u = dims
    args, kw = xbox_problem(np.random.default_rng(50 + world + Nc + x), oracle, M, N, x, u, Nc, 0.4, pull=0.8)
    Xo, Uo = oracle.lqp_solve_py(*args, Nc=Nc, **kw)
    assert np.sum((np.abs(Xo - kw["x_l"]) < 1e-8) | (np.abs(Xo - kw["x_u"]) < 1e-8)) > 0  # some state rows bind at the optimum
    X1, U1, _ = _solve_sharded(args, kw, Nc, 1, repeats=2)
    Xw, Uw, infos = _solve_sharded(args, kw, Nc, world, repeats=2)
    rel = lambda a, b: np.linalg.norm(a - b) / max(np.linalg.norm(b), 1.0)
    assert rel(X1, Xo) < 1e-7 and rel(U1, Uo) < 1e-7, (rel(X1, Xo), rel(U1, Uo))
    assert rel(Xw, X1) < 1e-9 and rel(Uw, U1) < 1e-9, (rel(Xw, X1), rel(Uw, U1))
    assert np.all(Uw[:, :Nc] == Uw[0:1, :Nc])


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_rounds_with_restart_and_elementwise_update(world, oracle):
    """The later rounds' work savers on sharded particles: factor sweeps restarted from checkpoints (as_ckpt, on by default) and settled
    particles updated elementwise from the forward sweep's sensitivity records (as_sens_min_m forced to 1: normally on from 3072 particles per
    rank) — the shared step every rank applies is the all-reduced one.  A cold solve whose rounds take several passes; against one rank with
    both options off (1e-9) and the oracle."""
    M, N, x, u, Nc = 8, 12, 12, 4, 1
    rng = np.random.default_rng(4242)
    args, kw = rand_problem(rng, M, N, x, u, 0.3)
    Xo, Uo = oracle.lqp_solve_py(*args, Nc=Nc, **kw)
    X0, U0, _ = _solve_sharded(args, kw, Nc, 1, options=dict(as_ckpt=0, as_sens_min_m=0))
    Xw, Uw, infos = _solve_sharded(args, kw, Nc, world, options=dict(as_sens_min_m=1))
    assert infos[0]["active_set_rounds"] >= 3, infos[0]  # (several rounds: later ones restart sweeps and update settled particles elementwise)
    rel = lambda a, b: np.linalg.norm(a - b) / max(np.linalg.norm(b), 1.0)
    assert rel(X0, Xo) < 1e-7 and rel(U0, Uo) < 1e-7
    assert rel(Xw, X0) < 1e-9 and rel(Uw, U0) < 1e-9, (rel(Xw, X0), rel(Uw, U0))
    assert np.all(Uw[:, :Nc] == Uw[0:1, :Nc])
    assert len({(i["ipm_iters"], i["active_set_rounds"], i["structured_solves"]) for i in infos}) == 1


def test_eight_ranks_quadrotor_shape(oracle):
    """bench.py's 8-GPU layout in miniature: 8 ranks x 4 particles, quadrotor dimensions, Nc = 1, control boxes."""
    from pmpc_amd import dynamics as dyn

    prob = dyn.make_quadrotor_problem(M=32, N=12)
    f, fx, fu = prob["f_fx_fu_fn"](np.concatenate([prob["x0"][:, None, :], prob["X_prev"][:, :-1, :]], 1), prob["U_prev"])
    args = (prob["x0"], f, fx, fu, prob["X_prev"], prob["U_prev"], prob["Q"], prob["R"], prob["X_ref"], prob["U_ref"])
    kw = dict(reg_x=prob["reg_x"], reg_u=prob["reg_u"], u_l=prob["u_l"], u_u=prob["u_u"])
    Xo, Uo = oracle.lqp_solve_py(*args, Nc=1, **kw)
    Xw, Uw, infos = _solve_sharded(args, kw, 1, 8, repeats=2)
    assert np.linalg.norm(Xw - Xo) / np.linalg.norm(Xo) < 1e-7 and np.linalg.norm(Uw - Uo) / np.linalg.norm(Uo) < 1e-7
    assert np.all(Uw[:, :1] == Uw[0:1, :1]) and len({(i["ipm_iters"], i["active_set_rounds"]) for i in infos}) == 1


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("kink", [False, True])
def test_sharded_cone_path_matches_cone_oracle(world, kink, oracle):
    """`pmpc_lcone_solve_device` on sharded particles: the particle costs are gathered so that every rank ranks them
    identically (threshold particle, kink bisection) — against the cone oracle on the whole ensemble."""
    from tests.test_cone_gpu import _make_kink

    M, N, x, u, Nc = 60, 6, 4, 2, 1
    args, kw = rand_problem(np.random.default_rng(4005), M, N, x, u, 1.0)
    if kink:
        args, kw = _make_kink(oracle, args, kw, Nc)
    Xo, Uo, info = oracle.lcone_solve_py(*args, Nc=Nc, return_info=True, **kw)
    assert bool(info.get("kink", False)) == kink
    Xw, Uw, infos = _solve_sharded(args, kw, Nc, world, cone=True)
    assert np.linalg.norm(Xw - Xo) / np.linalg.norm(Xo) < 1e-6 and np.linalg.norm(Uw - Uo) / np.linalg.norm(Uo) < 1e-6
    assert len({i["outer_solves"] for i in infos}) == 1 and infos[0]["outer_solves"] >= 2


@pytest.mark.parametrize("world,copies,others,Nc", [(2, 5, 3, 1), (4, 5, 3, 1), (2, 3, 3, 2)])
def test_sharded_cone_path_with_ties_matches_single_rank(world, copies, others, Nc):
    """The epigraph path on sharded particles: every rank holds all multipliers, checks the gathered costs and solves the same epigraph
    problem on the all-gathered condensed quadratics — `copies` identical cheapest particles (a `copies`-way tie on the threshold)
    spread over the ranks.  Same answer as one rank (1e-8) and as the direct cone program (1e-6)."""
    from oracle import cone_oracle as co
    from tests.test_cone_ties_gpu import tied_problem

    args, kw = tied_problem(np.random.default_rng(700 + world + copies + Nc), copies, others, 6, 4, 2, 0.4, Nc)
    Xo, Uo = co.lcone_direct_py(*args, Nc=Nc, **kw)
    X1, U1, _ = _solve_sharded(args, kw, Nc, 1, cone=True)
    Xw, Uw, infos = _solve_sharded(args, kw, Nc, world, repeats=2, cone=True)
    rel = lambda a, b: np.linalg.norm(a - b) / max(np.linalg.norm(b), 1.0)
    assert rel(X1, Xo) < 1e-6 and rel(U1, Uo) < 1e-6, (rel(X1, Xo), rel(U1, Uo))
    assert rel(Xw, X1) < 1e-8 and rel(Uw, U1) < 1e-8, (rel(Xw, X1), rel(Uw, U1))
    assert np.all(Uw[:, :Nc] == Uw[0:1, :Nc])


@pytest.mark.parametrize("world,copies,others,Nc,alpha,seed", [(2, 3, 3, 1, 10.0, 906), (4, 5, 3, 1, 10.0, 910), (2, 3, 3, 2, 10.0, 506), (4, 3, 5, 1, 100.0, 908)])
def test_sharded_smoothed_cone_objective_matches_single_rank_and_direct_program(world, copies, others, Nc, alpha, seed):
    """Log-barrier smoothing of the cone objective (main.jl:246-262) on sharded particles: the Newton iteration's host side runs on gathered
    per-particle scalars and condensed blocks, identically on every rank; `copies` identical cheapest particles (a tie on the threshold, each with
    a fractional multiplier) spread over the ranks.  Same answer as one rank (1e-9) and as the direct cone program (1e-6); a second, warm solve too."""
    from oracle import cone_oracle as co
    from tests.test_cone_ties_gpu import tied_problem

    args, kw = tied_problem(np.random.default_rng(seed), copies, others, 6, 4, 2, 0.4, Nc)
    Xo, Uo = co.lcone_direct_py(*args, Nc=Nc, smooth_alpha=alpha, **kw)
    sm = dict(smooth_alpha=alpha)
    X1, U1, _ = _solve_sharded(args, kw, Nc, 1, cone=True, smooth=sm)
    Xw, Uw, infos = _solve_sharded(args, kw, Nc, world, repeats=2, cone=True, smooth=sm)
    rel = lambda a, b: np.linalg.norm(a - b) / max(np.linalg.norm(b), 1.0)
    assert rel(X1, Xo) < 1e-6 and rel(U1, Uo) < 1e-6, (rel(X1, Xo), rel(U1, Uo))
    assert rel(Xw, X1) < 1e-9 and rel(Uw, U1) < 1e-9, (rel(Xw, X1), rel(Uw, U1))
    assert np.all(Uw[:, :Nc] == Uw[0:1, :Nc])
    assert len({i["ipm_iters"] for i in infos}) == 1 and all(i["fast_path"] == 1 for i in infos)


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_squareplus_cone_objective_matches_single_rank_and_direct_program(world):
    """smooth_cstr = "squareplus" (main.jl:265-279) on sharded particles: the same full-space Newton iteration with the hinge instead of the log."""
    from oracle import cone_oracle as co
    from tests.test_cone_ties_gpu import tied_problem

    args, kw = tied_problem(np.random.default_rng(930 + world), 3, 5, 6, 4, 2, 0.4, 1)
    Xo, Uo = co.lcone_direct_py(*args, Nc=1, smooth_alpha=10.0, smooth_cstr="squareplus", smooth_beta=5.0, **kw)
    sm = dict(smooth_alpha=10.0, smooth_cstr="squareplus", smooth_beta=5.0)
    X1, U1, _ = _solve_sharded(args, kw, 1, 1, cone=True, smooth=sm)
    Xw, Uw, _ = _solve_sharded(args, kw, 1, world, cone=True, smooth=sm)
    rel = lambda a, b: np.linalg.norm(a - b) / max(np.linalg.norm(b), 1.0)
    assert rel(X1, Xo) < 1e-6 and rel(U1, Uo) < 1e-6, (rel(X1, Xo), rel(U1, Uo))
    assert rel(Xw, X1) < 1e-9 and rel(Uw, U1) < 1e-9, (rel(Xw, X1), rel(Uw, U1))


@pytest.mark.parametrize("Nc", [0, 1, -1])
def test_sharded_stage_cones_match_oracle(Nc, oracle):
    """`pmpc_lsoc_solve_device` on 2 ranks: the complementarity sums, step lengths and failure flag cross ranks."""
    M, N, x, u = 6, 7, 5, 3
    args, kw = rand_problem(np.random.default_rng(77), M, N, x, u, 0.6)
    W = np.zeros((2, 3)); W[0, 1] = W[1, 2] = 1.0
    soc = dict(W=W, w0=np.zeros(2), v=np.array([0.5, 0, 0]), v0=0.05, u_interior=np.array([0.2, 0, 0]))
    Xo, Uo = oracle.lsoc_solve_py(*args, Nc=Nc, reg_x=kw["reg_x"], reg_u=kw["reg_u"], u_l=kw["u_l"], u_u=kw["u_u"], soc_W=W,
                                  soc_w0=soc["w0"], soc_v=soc["v"], soc_v0=soc["v0"], u_interior=soc["u_interior"])
    Xw, Uw, _ = _solve_sharded(args, kw, Nc, 2, repeats=2, soc=soc)
    assert np.linalg.norm(Xw - Xo) / np.linalg.norm(Xo) < 1e-6 and np.linalg.norm(Uw - Uo) / np.linalg.norm(Uo) < 1e-6


_RCCL_SINGLE_SCRIPT = r"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from pmpc_amd.device import DeviceSolver
from tests.support.problems import rand_problem

def make(comm):
    s = DeviceSolver(0)
    if comm:
        buf = ctypes.create_string_buffer(128)
        assert s.lib.pmpc_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p)) == 0
        assert s.lib.pmpc_comm_init(s.h, 0, 1, ctypes.cast(buf, ctypes.c_void_p)) == 0
    return s

plain, rccl = make(False), make(True)
dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda")
T = lambda a: dev(np.swapaxes(a, -1, -2))
worst = 0.0
for k, (M, N, x, u, Nc, kind) in enumerate([(8, 9, 12, 4, 1, "qp"), (6, 8, 5, 3, -1, "qp"), (8, 7, 4, 2, 3, "qp"), (8, 9, 12, 4, 0, "qp"),
                                            (40, 6, 4, 2, 1, "cone"), (6, 7, 5, 3, 1, "soc"), (12, 6, 4, 2, 1, "smooth"), (10, 6, 4, 2, 2, "smooth"),
                                            (8, 6, 4, 2, 1, "squareplus")]):
    args, kw = rand_problem(np.random.default_rng(500 + k), M, N, x, u, 0.4 if kind != "soc" else 0.6)
    x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref = args
    opt = dict(f=dev(f), fx=T(fx), fu=T(fu), X_prev=dev(X_prev), U_prev=dev(U_prev), Q=T(Q), R=T(R), X_ref=dev(X_ref),
               U_ref=dev(U_ref), reg_x=kw["reg_x"], reg_u=kw["reg_u"], Nc=Nc, symmetric_cost=True, lu=dev(kw["u_l"]), uu=dev(kw["u_u"]))
    if kind == "soc":
        W = np.zeros((2, 3)); W[0, 1] = W[1, 2] = 1.0
        opt.update(soc_W=dev(W), soc_w0=dev(np.zeros(2)), soc_v=dev(np.array([0.5, 0, 0])), soc_v0=0.05, soc_u_interior=dev(np.array([0.2, 0, 0])))
    if kind == "smooth":  # log-barrier smoothing of the cone objective: the gathers of lcone_smooth_body (all-reduce of zero-padded tables) and its broadcast
        opt.update(smooth_alpha=10.0)
    if kind == "squareplus":
        opt.update(smooth_alpha=10.0, smooth_cstr="squareplus", smooth_beta=5.0)
    res = []
    for s in (plain, rccl):
        for rep in range(2):  # the second solve is warm-started and reuses the broadcast consensus bounds
            X, U, status = getattr(s, {"qp": "lqp_solve", "cone": "lcone_solve", "soc": "lsoc_solve", "smooth": "lcone_solve", "squareplus": "lcone_solve"}[kind])(static_cons_bounds=rep > 0, **opt)
            s.sync()
            assert status == 0, (kind, status)
        res.append((X.cpu().numpy(), U.cpu().numpy()))
    e = max(np.abs(res[0][0] - res[1][0]).max(), np.abs(res[0][1] - res[1][1]).max())
    if kind == "soc":  # (the cone rounds stop at a relative step of 1e-6, i.e. ~1e-10 from the optimum; the two contexts differ in their warm-start history)
        assert e < 1e-7, e
        e = 0.0
    worst = max(worst, e)
    print(kind, (M, N, x, u, Nc), "max abs diff", e, flush=True)
assert worst < 1e-9, worst
plain.close(); rccl.close()
print("RCCL_SINGLE_OK")
"""


def test_real_rccl_calls_on_a_one_rank_communicator():
    """RCCL itself, as far as a one-GPU box allows: with PMPC_RCCL_SINGLE=1 (test hook) `pmpc_comm_init` builds a REAL 1-rank
    RCCL communicator (ncclGetUniqueId / ncclCommInitRank through the library's own dlopen) and every solve goes through the
    multi-rank code paths — ncclAllReduce (sum / min / max, fp64 and int32, in place, on the solver's non-blocking stream,
    next to the host's sequence-number polling) and ncclBroadcast.  Own process: the mock communicator of the tests above
    replaces the collectives process-wide."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    env = dict(os.environ, PMPC_RCCL_SINGLE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _RCCL_SINGLE_SCRIPT], cwd=str(Path(__file__).resolve().parents[1]), env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_SINGLE_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_bench_calling_pattern_sharded_over_mock_ranks():
    """bench.py's loop (outputs fed back as X_prev / U_prev, PMPC_STATIC_CONS_BOUNDS + PMPC_PREV_IS_LAST_SOLUTION from the second
    iteration) on 2 / 4 / 8 in-process ranks against one rank: tools/debug/sharded_scp_loop.py, in its own process."""
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    r = subprocess.run([sys.executable, "tools/debug/sharded_scp_loop.py", "64"], cwd=str(root), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "SHARDED_SCP_OK" in r.stdout and "SHARDED_NCN_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
