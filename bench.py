#!/usr/bin/env python3
"""bench.py — SCP iterations/sec of the particle SCP-MPC hot path on MI355X.

One "step" = one full SCP iteration on synthetic particle batches, everything resident in HBM:
    linearise dynamics on device  ->  convex sub-problem (c_lqp_solve semantics, device API)
    ->  SCP residual max(|dX|_2, |dU|_2)  ->  X_prev, U_prev <- X, U
Workload (BASELINE.json north_star / BASELINE.md config D): synthetic quadrotor xdim=12 udim=4,
M=4096 particles, N=50, Nc=1 consensus, box constraints on the controls, fp64.  With --gpus N the
4096 particles are sharded N ways (strong scaling), one process per GPU, RCCL all-reduce of the
consensus Hessian/gradient and of the active-set change counters (interior-point scalars on the fallback path) only.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel,
HIP-event timed on the solver's own stream) and `cpu_baseline` (oracle timed on a bounded sample).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

os.environ.setdefault("OMP_WAIT_POLICY", "passive")  # CPU baselines: no spinning OpenMP barriers under a cgroup CPU quota

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
ELEMS_PER_UNIT = 420   # elements per (particle, stage) at x12,u4 with control bounds
FP64_PEAK_TFLOPS = 78.6  # AMD MI355X data sheet: fp64 matrix = fp64 vector = 78.6 TFLOP/s (256 CUs x 4 SIMDs x 32 flop/clk x 2.4 GHz: one
#                          v_mfma_f64_16x16x4 = 2048 flop per 64 cycles per SIMD); the CDNA4 guide in this image quotes no fp64 figure


def riccati_flops_per_unit(x, u):
    """matrix Riccati work per (particle, stage), 4x^3 + 6x^2u + 4xu^2 + u^3/3 (11.2 kflop at x12, u4)."""
    return 4 * x ** 3 + 6 * x * x * u + 4 * x * u * u + u ** 3 / 3.0


def pmc_sq_evidence(kernel_prefix):
    """{counter: share of SQ_WAVE_CYCLES} of one kernel from the committed counter summary (tools/debug/pmc_sq_summary.py output), or None."""
    f = ROOT / "profiles" / "r05_pmc_sq_D.txt"
    if not f.exists():
        return None
    lines = f.read_text().splitlines()
    for k, ln in enumerate(lines):
        if kernel_prefix in ln:
            for nxt in lines[k + 1:k + 4]:
                if "of wave cycles:" in nxt:
                    out = {}
                    for part in nxt.split("of wave cycles:")[1].split("  "):
                        part = part.strip()
                        if part:
                            name, val = part.rsplit(" ", 1)
                            out[name] = float(val.rstrip("%")) / 100.0
                    return out
    return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--M", type=int, default=4096)
    ap.add_argument("--N", type=int, default=50)
    ap.add_argument("--model", default="quadrotor", choices=["quadrotor", "unicycle"])
    ap.add_argument("--Nc", type=int, default=1, help="consensus horizon (-1 = N, the reference default)")
    ap.add_argument("--weak", action="store_true", help="weak scaling: --M particles PER GPU instead of in total")
    ap.add_argument("--soc", action="store_true", help="quadrotor only: add the thrust cone ||(tau_x,tau_y)|| <= 0.3 T per stage "
                    "(config E's constraint set, fp64; pmpc_lsoc_solve_device)")
    ap.add_argument("--cone", action="store_true", help="the reference's DEFAULT solver path (solver=\"ecos\" -> c_lcone_solve, pmpc/static_backend.py:242-253): "
                    "the eps-anchored epigraph objective of PMPC.jl/src/main.jl:204-238 through pmpc_lcone_solve_device, Python-driven SCP loop")
    ap.add_argument("--smooth-alpha", type=float, default=float("nan"), help="with --cone: log-barrier smoothing of the boxes (main.jl:246-262); NaN = hard boxes")
    ap.add_argument("--fp32", action="store_true", help="fp32-STORAGE mode (BASELINE config E's dtype; the reference is fp64-only): fx, fu, Q, R and the "
                    "factor records in float32 — half the HBM bytes of the dominant arrays —, arithmetic fp64 (include/pmpc_abi.h PMPC_F32_MATRICES)")
    ap.add_argument("--repeats", type=int, default=4, help="extra repeats of the timed window from a fresh SCP start (spread; outside `value`)")
    ap.add_argument("--thrust-min", type=float, default=0.0, help="quadrotor: lower box of the thrust as a fraction of the hover thrust "
                    "(default 0: the thrust cone's apex is feasible; > 0 keeps every cone away from its apex)")
    ap.add_argument("--vmax", type=float, default=0.0, help="quadrotor: state boxes |v| <= VMAX m/s on the three velocity components "
                    "(the other states unbounded); 0 = no state boxes (BASELINE config D).  With a limit that binds the sub-problems "
                    "exercise the state rows of the active-set rounds (kernels_xbox.hip; PMPC_XBOX_AS=0: the interior-point iteration)")
    ap.add_argument("--trace-steps", action="store_true", help="print (interior-point iterations, active-set rounds, factorisations) of every "
                    "SCP iteration of the first window to stderr")
    ap.add_argument("--python-loop", action="store_true", help="drive the SCP loop from Python (one linearise / solve / residual call per "
                    "iteration) instead of pmpc_scp_loop_device")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-generic", action="store_true")
    ap.add_argument("--verbose", nargs="?", const=1, default=0, type=int)
    ap.add_argument("--profile-all", action="store_true", help="HIP events around every launch class, not only the dominant kernel")
    ap.add_argument("--ignore-status", action="store_true", help="timing experiments with deliberately wrong kernels")
    return ap.parse_args()


def cpu_baseline(model, M_total, N, Nc, budget_s=20.0):
    """The reference-shaped CPU path (oracle: single-threaded CSC assembly as lqp_utils.jl + restated OSQP
    ADMM with a fresh sparse factorisation per call, eps=1e-3 = OSQP default) timed on a bounded
    sample of the SAME workload: the first M_s particles of the same seeded batch, one SCP
    iteration.  The joint QP is linear in M apart from the shared Nc*udim consensus columns, so the
    per-iteration time is scaled by M_total / M_s to the metric's unit."""
    from oracle import lqp_oracle as orc
    from pmpc_amd import dynamics as dyn

    def one(Ms):
        prob = dyn.make_quadrotor_problem(M=Ms, N=N, Nc=Nc) if model == "quadrotor" else dyn.make_unicycle_problem(M=Ms, N=N, Nc=Nc)
        X_ = np.concatenate([prob["x0"][:, None, :], prob["X_prev"][:, :-1]], 1)
        f, fx, fu = prob["f_fx_fu_fn"](X_, prob["U_prev"])
        M, _, x = f.shape
        u = fu.shape[-1]
        nanx = np.full((M, N, x), np.nan)
        nanv = np.full(M, np.nan)
        args = (x, u, N, M, Nc, prob["x0"], f, orc.to_abi_mat(fx), orc.to_abi_mat(fu), prob["X_prev"], prob["U_prev"],
                orc.to_abi_mat(prob["Q"]), orc.to_abi_mat(prob["R"]), prob["X_ref"], prob["U_ref"], nanx, nanx,
                prob["u_l"], prob["u_u"], prob["reg_x"], prob["reg_u"], nanv, nanv, np.full((M, u), np.nan))
        t = time.perf_counter()
        _, _, tm = orc.reference_shaped_solve_abi(*args[:5], *args[6:])
        return time.perf_counter() - t, tm

    Ms = 8
    t, tm = one(Ms)
    while t < budget_s / 4 and Ms < M_total:
        Ms = min(M_total, Ms * 2)
        t, tm = one(Ms)
    scaled = t * (M_total / Ms)
    return dict(value=1.0 / scaled, unit="SCP iterations/s", cores=1, kind="port",
                sample=(f"{model} M_s={Ms} of {M_total} particles, N={N}, one SCP sub-problem: CSC assembly "
                        f"{tm['assemble_s']:.2f}s + restated OSQP (eps=1e-3, {tm['iters']} ADMM its, "
                        f"{tm['factorizations']} sparse LU) {tm['solve_s']:.2f}s = {t:.2f}s; scaled x{M_total / Ms:g}"))


def host_cores():
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (the GPU boxes expose 256 hardware
    threads under a 16-CPU quota; 256 spinning OpenMP threads there run 50x slower than 16)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline_structured(model, M_total, N, Nc, budget_s=6.0):
    """Second, stronger CPU line: the SAME structured algorithm as the HIP path (Riccati + condensing
    + Mehrotra on the boxes) in plain C with OpenMP over particles (oracle/structured_cpu.c), all host cores, on the first
    M_s particles of the same seeded batch; per-particle work is independent, so the time scales by M_total / M_s."""
    from oracle import lqp_oracle as orc
    from pmpc_amd import dynamics as dyn

    cores = host_cores()

    def one(Ms):
        prob = dyn.make_quadrotor_problem(M=Ms, N=N, Nc=Nc) if model == "quadrotor" else dyn.make_unicycle_problem(M=Ms, N=N, Nc=Nc)
        X_ = np.concatenate([prob["x0"][:, None, :], prob["X_prev"][:, :-1]], 1)
        f, fx, fu = prob["f_fx_fu_fn"](X_, prob["U_prev"])
        _, _, info = orc.structured_cpu_solve_py(prob["x0"], f, fx, fu, prob["X_prev"], prob["U_prev"], prob["Q"], prob["R"],
                                                 prob["X_ref"], prob["U_ref"], prob["reg_x"], prob["reg_u"], Nc=Nc,
                                                 u_l=prob["u_l"], u_u=prob["u_u"], threads=cores)
        assert info["status"] == 0, info
        return info

    Ms = min(M_total, 32 * cores)
    info = one(Ms)  # also warms the thread pool up
    info = one(Ms)
    while info["solve_s"] < budget_s / 4 and Ms < M_total:
        Ms = min(M_total, Ms * 2)
        info = one(Ms)
    scaled = info["solve_s"] * (M_total / Ms)
    return dict(value=1.0 / scaled, unit="SCP iterations/s", cores=cores, kind="port",
                sample=(f"{model} M_s={Ms} of {M_total} particles, N={N}, one SCP sub-problem by the structured algorithm in C + "
                        f"OpenMP (oracle/structured_cpu.c, {info['iters']} interior-point iterations, cold start) "
                        f"{info['solve_s']:.2f}s; scaled x{M_total / Ms:g}"))


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU) through
    torch.distributed.run and relay their output (rank 0 prints the JSON line).  Runs before this process touches torch or
    the GPU; the children are ordinary subprocesses (no exec of a GPU-initialised process)."""
    import socket
    import subprocess

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')} ranks were launched")
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # nccl == RCCL on ROCm

    from pmpc_amd import dynamics as dyn
    from pmpc_amd.device import MODEL_QUADROTOR, MODEL_UNICYCLE, DeviceSolver, to_device_problem

    M_total, N, Nc = (args.M * world if args.weak else args.M), args.N, (args.N if args.Nc < 0 else args.Nc)
    assert M_total % world == 0
    M_loc = M_total // world
    if args.model == "quadrotor":
        prob = dyn.make_quadrotor_problem(M=M_total, N=N, Nc=Nc)
        if args.thrust_min > 0.0:  # hover thrust = half the upper box (make_quadrotor_problem: T <= 2 m g)
            prob["u_l"][..., 0] = args.thrust_min * 0.5 * prob["u_u"][..., 0]
        model, x, u = MODEL_QUADROTOR, 12, 4
    else:
        prob = dyn.make_unicycle_problem(M=M_total, N=N, Nc=Nc)
        model, x, u = MODEL_UNICYCLE, 4, 2
    sl = slice(rank * M_loc, (rank + 1) * M_loc)  # contiguous particle shard
    shard = {k: (v[sl] if isinstance(v, np.ndarray) and v.shape[:1] == (M_total,) else v) for k, v in prob.items()}
    d = to_device_problem(shard, dev)
    solver = DeviceSolver(local_rank)
    solver.init_comm()

    X0, U0 = d["X_prev"].clone(), d["U_prev"].clone()  # the SCP loop's cold start: X_prev = hover at x0, U_prev = U_ref
    Xa, Ua = X0.clone(), U0.clone()
    Xb, Ub = torch.empty_like(Xa), torch.empty_like(Ua)
    f = torch.empty((M_loc, N, x), dtype=torch.float64, device=dev)
    mdt = torch.float32 if args.fp32 else torch.float64  # storage type of the matrix stacks
    fx = torch.empty((M_loc, N, x, x), dtype=mdt, device=dev)
    fu = torch.empty((M_loc, N, u, x), dtype=mdt, device=dev)
    if args.fp32:
        d["Q"], d["R"] = d["Q"].to(torch.float32).contiguous(), d["R"].to(torch.float32).contiguous()
    hist = []

    soc_kw = {}
    if args.soc:
        assert args.model == "quadrotor"
        Wc = torch.zeros((2, 4), dtype=torch.float64, device=dev)
        Wc[0, 1] = Wc[1, 2] = 1.0
        soc_kw = dict(soc_W=Wc, soc_w0=torch.zeros(2, dtype=torch.float64, device=dev),
                      soc_v=torch.tensor([0.3, 0.0, 0.0, 0.0], dtype=torch.float64, device=dev), soc_v0=0.0,
                      soc_u_interior=torch.tensor([9.81, 0.0, 0.0, 0.0], dtype=torch.float64, device=dev))
    xb_kw = {}
    if args.vmax > 0.0:
        assert args.model == "quadrotor" and not args.soc
        lx = torch.full((M_loc, N, x), -float("inf"), dtype=torch.float64, device=dev)
        lx[..., 3:6] = -args.vmax
        xb_kw = dict(lx=lx, ux=-lx)
    solve_fn = solver.lsoc_solve if args.soc else solver.lqp_solve
    if args.cone:
        assert not args.soc
        import functools

        solve_fn = functools.partial(solver.lcone_solve, smooth_alpha=args.smooth_alpha)
    # the library loop drives the cone objective itself (PMPC_CONE_OBJECTIVE; smoothing: barrier_mu = 1 / alpha)
    cone_kw = dict(cone_objective=True, barrier_mu=(1.0 / args.smooth_alpha if args.smooth_alpha == args.smooth_alpha else 0.0)) if args.cone else {}
    solve_events = []  # (start, end) HIP events on the solver's stream around the convex sub-problem (repeat windows only)

    def step(Xp, Up, Xo, Uo, first, time_solve=False):
        solver.linearize(model, d["x0"], Xp, Up, d["params"], f, fx, fu, wait_current_stream=False)
        if time_solve:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record(solver.stream)
        _, _, status = solve_fn(**soc_kw, **xb_kw, f=f, fx=fx, fu=fu, X_prev=Xp, U_prev=Up, Q=d["Q"], R=d["R"], X_ref=d["X_ref"],
                                        U_ref=d["U_ref"], reg_x=prob["reg_x"], reg_u=prob["reg_u"], Nc=Nc, x0=d["x0"],
                                        lu=d.get("lu"), uu=d.get("uu"), X_out=Xo, U_out=Uo, verbose=args.verbose,
                                        force_generic=args.force_generic, symmetric_cost=True, wait_current_stream=False,
                                        static_cons_bounds=True, prev_is_last_solution=not first)  # (an SCP loop: same boxes,
                                        # and X_prev / U_prev are the previous iteration's solution)
        if time_solve:
            ev[1].record(solver.stream)
            solve_events.append(ev)
        if status != 0 and not args.ignore_status:
            raise SystemExit(f"solver failed with status {status}")
        res = solver.scp_residual(Xo, Xp, Uo, Up, wait_current_stream=False)  # SCP residual of pmpc/scp_mpc.py:397-403: one fused pass on the solver's stream
        if world > 1:
            with torch.cuda.stream(solver.stream):
                dist.all_reduce(res, op=dist.ReduceOp.MAX)
        return res

    f2, fx2, fu2 = torch.empty_like(f), torch.empty_like(fx), torch.empty_like(fu)

    def run(k, time_solve=False):
        nonlocal Xa, Ua, Xb, Ub
        if args.python_loop or time_solve or k == 0:
            for _ in range(k):
                res = step(Xa, Ua, Xb, Ub, first=len(hist) == 0, time_solve=time_solve)
                hist.append((res, dict(solver.last_info)))
                Xa, Xb, Ua, Ub = Xb, Xa, Ub, Ua
            return
        # the SCP loop inside the library (pmpc_scp_loop_device): same kernels, same sequence, no host work between iterations
        res, infos, last_in_out, done = solver.scp_loop(
            model, d["params"], k, f2=f2, fx2=fx2, fu2=fu2, first_cold=len(hist) == 0, **soc_kw, **xb_kw, f=f, fx=fx, fu=fu, X_prev=Xa, U_prev=Ua,
            Q=d["Q"], R=d["R"], X_ref=d["X_ref"], U_ref=d["U_ref"], reg_x=prob["reg_x"], reg_u=prob["reg_u"], Nc=Nc, x0=d["x0"],
            lu=d.get("lu"), uu=d.get("uu"), X_out=Xb, U_out=Ub, force_generic=args.force_generic, symmetric_cost=True,
            wait_current_stream=False, **cone_kw)
        if done != k and not args.ignore_status:
            raise SystemExit(f"solver failed with status {infos[-1]['status']} in SCP iteration {len(hist) + done + 1}")
        for i in range(done):
            hist.append((res[i:i + 1], infos[i]))
        if last_in_out:
            Xa, Xb, Ua, Ub = Xb, Xa, Ub, Ua

    def window(profile_level, time_solve=False):
        """W untimed + K timed SCP iterations from the SCP loop's cold start; returns (seconds, max over ranks)."""
        nonlocal Xa, Ua
        Xa.copy_(X0)
        Ua.copy_(U0)
        hist.clear()
        torch.cuda.synchronize()
        run(args.warmup)
        solver.sync()
        solver.profile(profile_level)
        solver.profile_read()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(args.steps, time_solve)
        solver.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    # ---- the contract's timed region: W warm-up + exactly K timed steps (HIP events around the dominant kernel only) ----------
    elapsed = window(2 if args.profile_all else 1)
    prof = solver.profile_read()
    timed = list(hist[args.warmup:])
    if args.trace_steps and rank == 0:
        print("per SCP iteration (ipm iterations, active-set rounds, factorisations):",
              [(h[1]["ipm_iters"], h[1]["active_set_rounds"], h[1]["structured_solves"]) for h in hist], file=sys.stderr)
    # ---- repeats of the same window from a fresh SCP start: spread of the measurement; the last one carries HIP events around
    #      every launch class and around the sub-problem (aff_solve) as a whole — outside the reported region ------------------
    rep_s = [elapsed]
    for r in range(args.repeats):
        lastrep = r == args.repeats - 1
        rep_s.append(window(2 if lastrep else 1, time_solve=lastrep))
    prof_all = solver.profile_read() if args.repeats > 0 else prof
    solve_ms = [a_.elapsed_time(b_) for a_, b_ in solve_events]
    # outside the timed region: the convex sub-problem alone (SURVEY.md section 8d asks for the aff_solve-only rate next
    # to the full-iteration rate) — same linearisation re-solved from a COLD start (re-solving an identical problem
    # warm would flatter the number), no dynamics / residual kernels
    solver.profile(1)  # (the cold solves' factor sweeps are the PLAIN full sweeps: equality-only solve + first active-set round)
    solver.profile_read()
    k2 = max(1, min(args.steps, 10))
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(k2):
        solve_fn(**soc_kw, f=f, fx=fx, fu=fu, X_prev=Xb, U_prev=Ub, Q=d["Q"], R=d["R"], X_ref=d["X_ref"], U_ref=d["U_ref"],
                         reg_x=prob["reg_x"], reg_u=prob["reg_u"], Nc=Nc, x0=d["x0"], lu=d.get("lu"), uu=d.get("uu"), X_out=Xa, U_out=Ua,
                         force_generic=args.force_generic, symmetric_cost=True, wait_current_stream=False, cold_start=True)
    solver.sync()
    aff_only = k2 / (time.perf_counter() - t1)
    prof_cold = solver.profile_read()
    solver.profile(False)
    ipm_its = [h[1]["ipm_iters"] for h in timed]
    solves = [h[1]["structured_solves"] for h in timed]
    as_rounds = [h[1]["active_set_rounds"] for h in timed]

    if rank == 0:
        value = args.steps / elapsed
        ms_f, n_f = prof["bwd_factor"]
        unit_bytes = ELEMS_PER_UNIT * 8 if (x, u) == (12, 4) else (2 * x * x + x * u + u * u + 3 * x + 2 * u + 2 * u + x + u) * 8
        if args.fp32:  # the matrix stacks (2 x^2 + x u + u^2 elements of the unit) are 4-byte entries
            unit_bytes -= 4 * (2 * x * x + x * u + u * u)
        alg_bytes = unit_bytes * M_loc * N  # per launch: every (particle, stage) of this rank's shard
        avg_s = (ms_f / max(n_f, 1)) * 1e-3
        achieved = alg_bytes / avg_s / 1e9 if n_f else 0.0
        ms_c, n_c = prof_cold["bwd_factor"]
        plain = None
        if n_c:
            avg_c = ms_c / n_c * 1e-3
            plain = {"avg_launch_ms": 1e3 * avg_c, "achieved": alg_bytes / avg_c / 1e9, "frac": alg_bytes / avg_c / 1e9 / HBM_PEAK_GBS,
                     "launches": int(n_c)}
        # PMC traffic of the timed kernel: measured once per round by tools/profile_gpu.sh + tools/summarize_profile.py (separate
        # --pmc passes, corrections as the MI355X guide prescribes) AT CONFIG D, fp64 — attached only to that workload
        traffic = traffic_kernel = traffic_profile = None
        tf = ROOT / "profiles" / "traffic.json"
        at_config_d = args.model == "quadrotor" and M_loc == 4096 and N == 50 and Nc == 1 and not (args.fp32 or args.soc or args.cone or args.force_generic or args.vmax > 0.0)
        # other workloads with a PMC pass of their own (tools/profile_gpu.sh): keyed by a tag in profiles/traffic.json["workloads"]
        wl_tag = None
        if args.model == "quadrotor" and M_loc == 4096 and Nc == 1 and not (args.cone or args.force_generic or args.vmax > 0.0):
            if args.soc and N == 100:
                wl_tag = "E_soc_fp32" if args.fp32 else "E_soc"
            elif args.fp32 and N == 50 and not args.soc:
                wl_tag = "D_fp32"
        if tf.exists() and wl_tag:
            try:
                tw = json.loads(tf.read_text()).get("workloads", {}).get(wl_tag)
                if tw:
                    traffic, traffic_kernel, traffic_profile = tw.get("bytes_per_launch"), tw.get("kernels"), tw.get("profile")
            except Exception:
                pass
        if tf.exists() and at_config_d:
            try:
                tj = json.loads(tf.read_text())
                # timed region = DEFECT instantiation once the SCP loop runs (second step on); plain instantiation otherwise
                key = "bwd_factor_defect" if "bwd_factor_defect_bytes_per_launch" in tj else "bwd_factor"
                traffic = tj.get(f"{key}_bytes_per_launch")
                traffic_kernel = tj.get(key, {}).get("kernels")
                traffic_profile = tj.get("profile")
                if plain is not None:
                    plain["traffic"] = tj.get("bwd_factor_as_plain_bytes_per_launch", tj.get("bwd_factor_bytes_per_launch"))
                    plain["traffic_kernel"] = tj.get("bwd_factor_as_plain" if "bwd_factor_as_plain" in tj else "bwd_factor", {}).get("kernels")
            except Exception:
                traffic = None
        per_solve = None
        if solve_ms:
            sm = float(np.mean(solve_ms)) * 1e-3
            per_solve = {"aff_solve_ms": 1e3 * sm, "achieved": alg_bytes / sm / 1e9, "frac": alg_bytes / sm / 1e9 / HBM_PEAK_GBS,
                         "note": "BASELINE.md section 3's definition: algorithmic bytes of ONE sub-problem / seconds per aff_solve (HIP events on the "
                                 "solver's stream around the whole sub-problem, last repeat window)"}
        # ---- what the dominant kernel is actually limited by: bytes AND flops fractions, and the A/B that names the limiter ----------
        flop_unit = riccati_flops_per_unit(x, u)
        flops_launch = flop_unit * M_loc * N
        ach_tf = flops_launch / avg_s / 1e12 if n_f else 0.0
        flops = {"flop_per_unit": flop_unit, "flop_per_launch": flops_launch, "achieved": ach_tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                 "frac": ach_tf / FP64_PEAK_TFLOPS,
                 "note": "useful flops of the matrix Riccati recursion (SURVEY.md section 8d) / launch time; the 16x16x4 fp64 MFMA tiles are "
                         "padded 12 -> 16 rows, so the matrix pipe is busier than this figure by (16/12)^2 on the G and H products"}
        # what bounds the dominant kernel: SQ counters of the profiled run (profiles/r05_pmc_sq_D.txt, tools/debug/pmc_sq.sh: three --pmc
        # passes at config D) are READ here, not restated; what this run measures itself is attached next to them
        sq = pmc_sq_evidence("k_bwd_as<12, 4, 1, false, true, 0, double>")
        sq_fwd = pmc_sq_evidence("k_fwd_as<12, 4, false, true, false, double>")
        rst = solver.restart_stats(reset=False)
        limiter = {"name": "vector-instruction issue: four waves per SIMD keep its issue port busy (fp64 VALU and 64-bit lane moves at a quarter of the fp32 rate)",
                   "evidence": {"profile": "profiles/r05_pmc_sq_D.txt (rocprofv3 --pmc, SQ_WAVE_CYCLES / SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_* per kernel, config D)",
                                "factor_sweep_share_of_wave_cycles": sq, "forward_sweep_share_of_wave_cycles": sq_fwd,
                                "reading": "parked at a memory wait for under a quarter of a wave's life, stalled at issue for over half; executing x 4 waves per SIMD ~ the whole "
                                           "issue port: neither HBM latency nor bytes in flight bound the sweeps at this size, instructions per stage and swept stages do",
                                "this_run": {"factor_sweep_avg_launch_ms": 1e3 * avg_s, "bytes_frac_of_peak": achieved / HBM_PEAK_GBS, "flops_frac_of_peak": ach_tf / FP64_PEAK_TFLOPS,
                                             "later_rounds_factor_sweeps": rst,
                                             "note": "later_rounds_factor_sweeps: sweeps of unsettled particles since the context was created — those restarted from a "
                                                     "checkpoint and the stages they ran (of N per full sweep), those from the terminal cost"}},
                   "bytes_frac_of_peak": achieved / HBM_PEAK_GBS, "flops_frac_of_peak": ach_tf / FP64_PEAK_TFLOPS}
        # ---- Nc > 1 (the reference's default is Nc = N): the consensus launch class (condensing kernel + particle reduction + dense
        #      solve) is a roofline object of its own.  SURVEY.md section 8(d) expected it MFMA-bound; measured it is HBM-bound on the
        #      per-particle condensed Hessians (M x (Nc u)^2 doubles written by the condensing kernel and read back by the reduction).
        cons_roof = None
        if Nc > 1 and "consensus" in prof_all and prof_all["consensus"][1] > 0:
            ncv = Nc * u
            ms_cons = prof_all["consensus"][0] / args.steps  # per step (every round's consensus launches)
            rounds_ps = max(1.0, float(np.mean(as_rounds))) if np.mean(as_rounds) > 0 else 1.0
            solves_ps = max(rounds_ps, float(np.mean(solves)))
            cond_flops = (x * x + x * u) * u * Nc * Nc * M_loc  # off-diagonal blocks: Y_j Gamma_{j-1,l} walked forward, 2 (x^2 + x u) flops per column and stage pair
            # bytes of the class as built (r04): the condensing kernel walks the consensus stages once per group of 4 column tiles (64 consensus
            # variables), reading fx, fu and the 64-double factor record of every stage from the tile group's first stage on; it writes ONE slab
            # of upper-triangle sums per 8 particles (k_cond_fast_grouped), which the reduction reads once.  (r03: every particle's (Nc u)^2
            # block went out and came back — 2.6 GB at config D with Nc = N; the same workload now moves 0.8 GB.)
            ntile = (ncv + 15) // 16
            stage_bytes = (x * x + x * u + 64) * 8
            walk = sum(max(0, Nc - (16 * t0) // u) for t0 in range(0, ntile, 4))
            hbytes = float(M_loc * walk * stage_bytes + 2 * ((M_loc + 7) // 8) * (ncv * (ncv + 1) // 2) * 8)
            t_one = ms_cons * 1e-3 / solves_ps
            cons_roof = {"bound": "hbm", "kernel": "consensus launch class: k_cond_fast_grouped (off-diagonal blocks of the condensed Hessians, summed over groups of 8 particles "
                                                   "before they leave the chip) + reduction of the group slabs + dense Cholesky (one workgroup, the matrix register-resident as fp64 MFMA blocks)",
                         "avg_ms_per_solve": 1e3 * t_one, "solves_per_step": solves_ps, "bytes_per_solve": hbytes, "achieved": hbytes / t_one / 1e9, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": hbytes / t_one / 1e9 / HBM_PEAK_GBS,
                         "flops": {"flop_per_solve": cond_flops, "achieved": cond_flops / t_one / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                   "frac": cond_flops / t_one / 1e12 / FP64_PEAK_TFLOPS},
                         "note": "SURVEY.md section 8(d) prices full-consensus condensing as a dense (N x) x (N u) contraction per particle against the fp64 MFMA "
                                 "peak; the structured form walks the stages (O(Nc^2) small tile products, ~8 GFLOP at config D).  Neither roof binds: the "
                                 "condensing kernel is a dependent chain of 3 fp64 MFMAs per tile and stage behind a per-stage barrier (0.30 ms at config D; tiles per "
                                 "wave, operand prefetch and an LDS-only barrier measured: CHANGELOG.md section 5.7), the (Nc u)^2 Cholesky is a chain of Nc u / 16 "
                                 "panels on ONE workgroup (0.12 ms at config D, profiles/r05_cons_solve_micro.txt)"}
        rates = [args.steps / t_ for t_ in rep_s]
        w0, w1 = args.warmup + 1, args.warmup + args.steps
        out = {
            "metric": "SCP iterations/sec (M particles x N horizon)", "value": value, "unit": "SCP iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak" if args.weak else "strong", "vs_baseline": None,
            "dtype": "f32 storage (fx, fu, Q, R, factor records) / f64 arithmetic" if args.fp32 else "f64", "data": "synthetic",
            "config": {"workload": (f"{args.model} x{x} u{u} M={M_total} N={N} Nc={Nc} box-u, full SCP iteration "
                                    "(on-device linearise + c_lqp_solve-equivalent + residual), BASELINE config D"
                                    if args.model == "quadrotor" and M_total == 4096 and N == 50 and Nc == 1 and not args.cone and not args.soc and not args.vmax > 0.0 else
                                    f"{args.model} x{x} u{u} M={M_total} N={N} Nc={Nc} box-u" + (" + thrust cone per stage (config E constraints, fp64)" if args.soc else "")
                                    + ((" through the reference's default cone path (c_lcone_solve semantics: eps-anchored epigraph objective"
                                        + (f", log-barrier smoothing alpha = {args.smooth_alpha:g})" if args.smooth_alpha == args.smooth_alpha else ", hard boxes)")) if args.cone else "")
                                    + (f", thrust >= {args.thrust_min} x hover" if args.thrust_min > 0.0 else "")
                                    + (f", state boxes |v| <= {args.vmax} m/s" if args.vmax > 0.0 else ""))
                                   + f"; timed window = SCP iterations {w0}..{w1} from the cold start X_prev = x0, U_prev = U_ref "
                                     "(the active-set round count falls as the SCP loop converges, so the rate depends on the window)",
                       "particles_per_gpu": M_loc, "parallelism": f"particle-shard x{world}",
                       "scp_loop": "python (one call per linearise / solve / residual)" if args.python_loop else "library (pmpc_scp_loop_device)",
                       "ipm_iters_per_step": float(np.mean(ipm_its)), "riccati_factorisations_per_step": float(np.mean(solves)),
                       "active_set_rounds_per_step": float(np.mean(as_rounds)),
                       "weighted_qps_per_step": float(np.mean([h[1].get("outer_solves", 0) for h in timed])),
                       "fast_path": bool(timed[-1][1]["fast_path"]), "final_scp_residual": float(timed[-1][0][0].item()),
                       "aff_solve_only_cold_per_s": aff_only,
                       "ipm_warm_start": os.environ.get("PMPC_WARM_START", "1") != "0"},
            "repeats": {"windows": len(rates), "values": rates, "median": float(np.median(rates)), "min": float(np.min(rates)),
                        "max": float(np.max(rates)),
                        "note": "`value` is the FIRST window (the contract's W + K steps); the others repeat it from a fresh SCP start in the same process"},
            "roofline": {"bound": "hbm", "kernel": "backward Riccati factor sweep", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_kernel": traffic_kernel, "traffic_profile": traffic_profile,
                         "note": "full factor sweeps of the timed region only (every particle x stage): later active-set rounds skip the "
                                 "settled particles and are timed in a class of their own (bwd_factor_partial). Inside the "
                                 "SCP loop the full sweep is the DEFECT instantiation (it also carries the base point's dynamics defect, "
                                 "which replaces a separate rollout kernel); plain_full_sweep = the "
                                 "same figure for the plain instantiation, measured over the cold solves after the timed region",
                         "plain_full_sweep": plain, "per_solve": per_solve,
                         "alg_bytes_per_launch": alg_bytes, "avg_launch_ms": 1e3 * avg_s, "launches": int(n_f),
                         "flops": flops, "limiter": limiter, "consensus_class": cons_roof,
                         "kernel_ms_per_step": {k: v[0] / args.steps for k, v in prof_all.items() if v[1] > 0},
                         "kernel_ms_per_step_note": "every launch class, HIP events, last repeat window (not the reported one)"},
        }
        if not args.no_cpu_baseline and world == 1:
            import shutil

            out["cpu_baseline"] = cpu_baseline(args.model, M_total, N, Nc)
            # north_star asks for the Julia back end on the same box: probed, never installed here (no julia in the image)
            out["cpu_baseline"]["julia"] = shutil.which("julia") or "absent (probed with shutil.which): the port below is what can run"
            out["cpu_baseline_structured"] = cpu_baseline_structured(args.model, M_total, N, Nc)
            # same algorithm, cold start on both sides (the SCP loop's warm-started rate is NOT comparable with a cold CPU solve)
            out["cpu_baseline_structured"]["gpu_cold_over_cpu_cold"] = aff_only / out["cpu_baseline_structured"]["value"]
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
