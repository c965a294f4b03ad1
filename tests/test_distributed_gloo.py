"""world_size-2 `gloo` test (CPU) of the particle-sharded solve: each rank holds a contiguous shard of the
particles and the ONLY data exchanged are the ones the GPU path all-reduces over RCCL — the condensed
consensus Hessian/gradient (sum), the IPM complementarity sums (sum), the step-length ratios (min) and the
residual norms (max) — SURVEY.md §8(e).  The sharded result must equal the oracle's joint solve."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, case, out_dir, active_set=False, state_rows=None):
    sys.path.insert(0, str(ROOT))
    import torch
    import torch.distributed as dist

    from tests.support import structured_np as snp
    from tests.support.problems import rand_problem

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    M, N, x, u, Nc, bu, bx = case
    if state_rows is not None:  # binding state boxes, feasible by construction (seed, pull)
        from oracle import lqp_oracle
        from tests.support.problems import xbox_problem

        args, kw = xbox_problem(np.random.default_rng(state_rows[0]), lqp_oracle, M, N, x, u, Nc, bu, pull=state_rows[1], margin=0.05)
    else:
        args, kw = rand_problem(np.random.default_rng(7), M, N, x, u, bu, bx)  # every rank draws the same batch
    ML = M // world
    sl = slice(rank * ML, (rank + 1) * ML)
    largs = tuple(a[sl] for a in args)
    lkw = {k: (v[sl] if isinstance(v, np.ndarray) and v.shape[:1] == (M,) else v) for k, v in kw.items()}
    k = N if Nc < 0 else Nc
    if "u_l" in lkw and k > 0:  # consensus-control bounds are global particle 0's (lqp_utils.jl:329-330)
        lkw["u_l"], lkw["u_u"] = lkw["u_l"].copy(), lkw["u_u"].copy()
        lkw["u_l"][:, :k], lkw["u_u"][:, :k] = kw["u_l"][0:1, :k], kw["u_u"][0:1, :k]

    def red(op):
        def f(a):
            scalar = np.isscalar(a) or np.ndim(a) == 0
            t = torch.as_tensor(np.atleast_1d(np.asarray(a, dtype=np.float64)).copy())
            dist.all_reduce(t, op=op)
            return float(t[0]) if scalar else t.numpy()
        return f

    p = snp.Problem(*largs[1:], Nc=Nc, **lkw)
    p.owns_consensus = rank == 0
    if state_rows is not None:  # + the state rows: one more pair {changes, open rows} in the per-round sum
        X, U, info = snp.active_set_solve_xb(p, allreduce=red(dist.ReduceOp.SUM))
        info["iters"] = info["rounds"] + 100 * info["phase1_rounds"]
    elif active_set:  # the primal-dual active-set iteration: [Hc | gc] sums + one change-counter sum per round
        X, U, info = snp.active_set_solve(p, allreduce=red(dist.ReduceOp.SUM))
        info["iters"] = info["rounds"]
    else:
        X, U, info = snp.ipm_solve(p, allreduce=red(dist.ReduceOp.SUM), allreduce_min=red(dist.ReduceOp.MIN),
                                   allreduce_max=red(dist.ReduceOp.MAX))
    np.savez(Path(out_dir) / f"rank{rank}.npz", X=X, U=U, iters=info["iters"])
    dist.destroy_process_group()


@pytest.mark.parametrize("case", [(6, 7, 3, 2, 1, 0.3, None), (6, 7, 3, 2, 3, 0.3, 6.0), (4, 6, 3, 2, -1, None, None), (8, 6, 4, 2, 0, 0.3, None)],
                         ids=["Nc1-ubox", "Nc3-ubox-xbox", "NcN-free", "Nc0-ubox"])
def test_sharded_solve_equals_joint_solve(case, oracle, tmp_path):
    _run_sharded(case, oracle, tmp_path, False)


@pytest.mark.parametrize("case", [(6, 7, 3, 2, 1, 0.3, None), (6, 7, 3, 2, 3, 0.2, None), (4, 6, 3, 2, -1, 0.05, None), (8, 6, 4, 2, 0, 0.3, None)],
                         ids=["Nc1-ubox", "Nc3-ubox", "NcN-ubox", "Nc0-ubox"])
def test_sharded_active_set_iteration_equals_joint_solve(case, oracle, tmp_path):
    """The active-set rounds of the GPU path (solver.hip `active_set_solve`) on 2 gloo ranks: the consensus controls' status
    is decided identically on both ranks from all-reduced quantities only."""
    _run_sharded(case, oracle, tmp_path, True)


def _run_sharded(case, oracle, tmp_path, active_set):
    import torch.multiprocessing as mp

    from tests.support.problems import rand_problem

    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, case, str(tmp_path), active_set), nprocs=world, join=True)
    M, N, x, u, Nc, bu, bx = case
    args, kw = rand_problem(np.random.default_rng(7), M, N, x, u, bu, bx)
    Xo, Uo = oracle.lqp_solve_py(*args, Nc=Nc, **kw)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    X, U = np.concatenate([p["X"] for p in parts]), np.concatenate([p["U"] for p in parts])
    assert parts[0]["iters"] == parts[1]["iters"]  # identical all-reduced scalars => identical control flow
    tol = 1e-10 if active_set else 1e-7  # the accepted active-set point is the vertex itself
    assert np.linalg.norm(X - Xo) / np.linalg.norm(Xo) < tol
    assert np.linalg.norm(U - Uo) / max(np.linalg.norm(Uo), 1.0) < tol
    k = N if Nc < 0 else Nc
    if k:
        assert np.all(U[:, :k] == U[0:1, :k])  # consensus across ranks, bitwise


# (M, N, x, u, Nc, u-bound, seed, pull): state boxes that bind in a few entries (tests/support/problems.py::xbox_problem)
XB_CASES = [(6, 10, 4, 2, 1, 0.4, 5300, 0.95), (6, 10, 4, 2, 1, None, 5301, 0.95), (4, 9, 6, 3, 2, 0.5, 5302, 0.95), (8, 8, 3, 2, 0, 0.4, 5303, 0.95)]


@pytest.mark.parametrize("case", XB_CASES, ids=lambda c: f"M{c[0]}N{c[1]}x{c[2]}u{c[3]}Nc{c[4]}")
def test_state_rows_model_matches_the_oracle(case, oracle):
    """The algorithm of pmpc_amd/csrc/kernels_xbox.hip restated in numpy (tests/support/structured_np.py::active_set_solve_xb):
    two-phase cold start, binding state boxes as semismooth-Newton rows with penalty + multiplier estimate, hysteresis, partial
    activation — lands on the oracle's optimum to round-off.  (CPU check of the METHOD; the HIP path itself is
    tests/test_xbox_gpu.py.)"""
    from tests.support import structured_np as snp
    from tests.support.problems import xbox_problem

    M, N, x, u, Nc, bu, seed, pull = case
    args, kw = xbox_problem(np.random.default_rng(seed), oracle, M, N, x, u, Nc, bu, pull=pull, margin=0.05)
    Xo, Uo = oracle.lqp_solve_py(*args, Nc=Nc, **kw)
    assert np.sum((Xo <= kw["x_l"] + 1e-9) | (Xo >= kw["x_u"] - 1e-9)) > 0
    X, U, info = snp.active_set_solve_xb(snp.Problem(*args[1:], Nc=Nc, **kw))
    assert info["held"] > 0 and info["rounds"] <= 8
    assert np.linalg.norm(X - Xo) / np.linalg.norm(Xo) < 1e-9 and np.linalg.norm(U - Uo) / max(np.linalg.norm(Uo), 1.0) < 1e-9


@pytest.mark.parametrize("case", [XB_CASES[0], XB_CASES[2]], ids=["Nc1", "Nc2"])
def test_sharded_state_rows_equal_joint_solve(case, oracle, tmp_path):
    """The same rounds on 2 gloo ranks: statuses of the shared controls and the accept decision come from all-reduced sums only
    (the consensus system and {changes, open rows}), so both ranks take the same number of rounds and land on the joint optimum."""
    import torch.multiprocessing as mp

    from tests.support.problems import xbox_problem

    M, N, x, u, Nc, bu, seed, pull = case
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, (M, N, x, u, Nc, bu, None), str(tmp_path), True, (seed, pull)), nprocs=world, join=True)
    args, kw = xbox_problem(np.random.default_rng(seed), oracle, M, N, x, u, Nc, bu, pull=pull, margin=0.05)
    Xo, Uo = oracle.lqp_solve_py(*args, Nc=Nc, **kw)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    X, U = np.concatenate([p["X"] for p in parts]), np.concatenate([p["U"] for p in parts])
    assert parts[0]["iters"] == parts[1]["iters"]
    assert np.linalg.norm(X - Xo) / np.linalg.norm(Xo) < 1e-9 and np.linalg.norm(U - Uo) / max(np.linalg.norm(Uo), 1.0) < 1e-9
    assert np.all(U[:, :Nc] == U[0:1, :Nc])
