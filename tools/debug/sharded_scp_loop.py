"""bench.py's calling pattern on ONE GPU through the in-process communicator: an SCP-like loop (re-linearised dynamics, X_prev /
U_prev = the previous outputs, PMPC_STATIC_CONS_BOUNDS + PMPC_PREV_IS_LAST_SOLUTION from the second iteration) on 2 / 4 / 8
ranks against the same loop on one rank.  usage: sharded_scp_loop.py [M]"""
import sys, threading, numpy as np, torch
sys.path.insert(0, ".")
from pmpc_amd import dynamics as dyn
from pmpc_amd.device import DeviceSolver

M = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N, ITERS = 50, 5
prob = dyn.make_quadrotor_problem(M=M, N=N)
rng = np.random.default_rng(3)
noise = [(0.02 * rng.standard_normal((M, N, 12)), 1 + 0.02 * rng.standard_normal((M, N, 12, 12)), 1 + 0.02 * rng.standard_normal((M, N, 12, 4)))
         for _ in range(ITERS)]
_group = [5000]


def run_world(world):
    Ml = M // world
    _group[0] += 1
    group, out, errs = _group[0], [None] * world, []

    def rank_fn(rank):
        try:
            sl = slice(rank * Ml, (rank + 1) * Ml)
            dev = lambda a: torch.tensor(np.ascontiguousarray(a[sl]), dtype=torch.float64, device="cuda")
            T = lambda a: dev(np.swapaxes(a, -1, -2))
            s = DeviceSolver(0)
            if world > 1:
                assert s.lib.pmpc_comm_init_mock(s.h, rank, world, group) == 0
                s.rank, s.world = rank, world
            Xp, Up = prob["X_prev"].copy(), prob["U_prev"].copy()  # full arrays; every rank linearises its own shard of them
            infos = []
            for t in range(ITERS):
                X_lin = np.concatenate([prob["x0"][:, None, :], Xp[:, :-1, :]], 1)
                f, fx, fu = prob["f_fx_fu_fn"](X_lin, Up)
                f, fx, fu = f + noise[t][0], fx * noise[t][1], fu * noise[t][2]
                X, U, status = s.lqp_solve(f=dev(f), fx=T(fx), fu=T(fu), X_prev=dev(Xp), U_prev=dev(Up), Q=T(prob["Q"]), R=T(prob["R"]),
                                           X_ref=dev(prob["X_ref"]), U_ref=dev(prob["U_ref"]), reg_x=prob["reg_x"], reg_u=prob["reg_u"], Nc=1,
                                           symmetric_cost=True, lu=dev(prob["u_l"]), uu=dev(prob["u_u"]), static_cons_bounds=t > 0,
                                           prev_is_last_solution=t > 0)
                s.sync()
                assert status == 0
                infos.append((s.last_info["ipm_iters"], s.last_info["active_set_rounds"]))
                # the next linearisation point = this solution; the other shards' parts are not needed by this rank's solve,
                # but its own shard must be exactly its last output
                Xp[sl], Up[sl] = X.cpu().numpy(), U.cpu().numpy()
            out[rank] = (Xp[sl].copy(), Up[sl].copy(), infos)
            s.close()
        except Exception as e:
            errs.append(e)

    th = [threading.Thread(target=rank_fn, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(timeout=300) for t in th]
    assert not errs, errs
    assert all(o is not None for o in out), "a rank did not finish"
    return np.concatenate([o[0] for o in out]), np.concatenate([o[1] for o in out]), [o[2] for o in out]


X1, U1, i1 = run_world(1)
print("single rank (ipm iterations, active-set rounds) per solve:", i1[0])
for world in (2, 4, 8):
    Xw, Uw, infos = run_world(world)
    ex, eu = np.linalg.norm(Xw - X1) / np.linalg.norm(X1), np.linalg.norm(Uw - U1) / np.linalg.norm(U1)
    print(f"world {world}: rel diff X {ex:.2e} U {eu:.2e}; per-solve (ipm, rounds) {infos[0]}; ranks agree: {len({tuple(i) for i in infos}) == 1}")
    assert ex < 1e-9 and eu < 1e-9 and len({tuple(i) for i in infos}) == 1
    assert all(i[0] == 0 for i in infos[0][1:]), "a later solve fell back to the interior-point path"


# ---- the same through pmpc_scp_loop_device (the loop body inside the library: what bench.py times), built-in quadrotor dynamics ----
from pmpc_amd.device import MODEL_QUADROTOR, to_device_problem  # noqa: E402


def run_world_lib(world):
    Ml = M // world
    _group[0] += 1
    group, out, errs = _group[0], [None] * world, []

    def rank_fn(rank):
        try:
            sl = slice(rank * Ml, (rank + 1) * Ml)
            shard = {k: (v[sl] if isinstance(v, np.ndarray) and v.shape[:1] == (M,) else v) for k, v in prob.items()}
            d = to_device_problem(shard)
            s = DeviceSolver(0)
            if world > 1:
                assert s.lib.pmpc_comm_init_mock(s.h, rank, world, group) == 0
                s.rank, s.world = rank, world
            mk = lambda *shape: torch.empty(shape, dtype=torch.float64, device="cuda")
            Xa, Ua = d["X_prev"].clone(), d["U_prev"].clone()
            Xb, Ub = torch.empty_like(Xa), torch.empty_like(Ua)
            res, infos, last_in_out, done = s.scp_loop(
                MODEL_QUADROTOR, d["params"], ITERS, f=mk(Ml, N, 12), fx=mk(Ml, N, 12, 12), fu=mk(Ml, N, 4, 12), f2=mk(Ml, N, 12),
                fx2=mk(Ml, N, 12, 12), fu2=mk(Ml, N, 4, 12), X_prev=Xa, U_prev=Ua, X_out=Xb, U_out=Ub, Q=d["Q"], R=d["R"], X_ref=d["X_ref"],
                U_ref=d["U_ref"], reg_x=prob["reg_x"], reg_u=prob["reg_u"], Nc=1, x0=d["x0"], lu=d["lu"], uu=d["uu"], symmetric_cost=True)
            s.sync()
            assert done == ITERS and all(i["status"] == 0 for i in infos), infos
            X, U = (Xb, Ub) if last_in_out else (Xa, Ua)
            out[rank] = (X.cpu().numpy(), U.cpu().numpy(), res.cpu().numpy(), [(i["ipm_iters"], i["active_set_rounds"]) for i in infos])
            s.close()
        except Exception as e:
            errs.append(e)

    th = [threading.Thread(target=rank_fn, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(timeout=300) for t in th]
    assert not errs, errs
    assert all(o is not None for o in out), "a rank did not finish"
    assert all(np.array_equal(o[2], out[0][2]) for o in out), "ranks disagree on the SCP residuals"
    return np.concatenate([o[0] for o in out]), np.concatenate([o[1] for o in out]), out[0][2], [o[3] for o in out]


X1, U1, r1, j1 = run_world_lib(1)
print("library loop, single rank: residuals", r1, "(ipm iterations, rounds)", j1[0])
for world in (2, 4):
    Xw, Uw, rw, infos = run_world_lib(world)
    ex, eu = np.linalg.norm(Xw - X1) / np.linalg.norm(X1), np.linalg.norm(Uw - U1) / np.linalg.norm(U1)
    print(f"library loop, world {world}: rel diff X {ex:.2e} U {eu:.2e}; residual diff {np.max(np.abs(rw / r1 - 1)):.1e}; ranks agree: {len({tuple(i) for i in infos}) == 1}")
    assert ex < 1e-9 and eu < 1e-9 and np.max(np.abs(rw / r1 - 1)) < 1e-9 and len({tuple(i) for i in infos}) == 1
print("SHARDED_SCP_OK")


# ---- full consensus (Nc = N) inside an SCP loop with the promise flags: every later solve must be warm (rounds only) on
#      one rank and on mock ranks alike, and the sharded loop must equal the single-rank one ----
def run_world_ncn(world, Mn=64, Nn=12):
    from tests.support.problems import rand_problem
    rng = np.random.default_rng(5)
    args, kw = rand_problem(rng, Mn, Nn, 4, 2, 0.25)
    x0, f0, fx0, fu0, X_prev, U_prev, Q, R, X_ref, U_ref = args
    pert = [(0.03 * rng.standard_normal(f0.shape), 1 + 0.03 * rng.standard_normal(fx0.shape), 1 + 0.03 * rng.standard_normal(fu0.shape)) for _ in range(4)]
    Ml = Mn // world
    _group[0] += 1
    group, out, errs = _group[0], [None] * world, []

    def rank_fn(rank):
        try:
            sl = slice(rank * Ml, (rank + 1) * Ml)
            dev = lambda a: torch.tensor(np.ascontiguousarray(a[sl]), dtype=torch.float64, device="cuda")
            T = lambda a: dev(np.swapaxes(a, -1, -2))
            s = DeviceSolver(0)
            if world > 1:
                assert s.lib.pmpc_comm_init_mock(s.h, rank, world, group) == 0
                s.rank, s.world = rank, world
            Xp, Up, infos = X_prev.copy(), U_prev.copy(), []
            for t in range(4):
                f, fx, fu = f0 + pert[t][0], fx0 * pert[t][1], fu0 * pert[t][2]
                X, U, status = s.lqp_solve(f=dev(f), fx=T(fx), fu=T(fu), X_prev=dev(Xp), U_prev=dev(Up), Q=T(Q), R=T(R), X_ref=dev(X_ref),
                                           U_ref=dev(U_ref), reg_x=kw["reg_x"], reg_u=kw["reg_u"], Nc=-1, symmetric_cost=True, lu=dev(kw["u_l"]),
                                           uu=dev(kw["u_u"]), static_cons_bounds=t > 0, prev_is_last_solution=t > 0)
                s.sync()
                assert status == 0
                i = dict(s.last_info)
                infos.append((i["ipm_iters"], i["active_set_rounds"], i["structured_solves"]))
                Xp[sl], Up[sl] = X.cpu().numpy(), U.cpu().numpy()
            out[rank] = (Xp[sl].copy(), Up[sl].copy(), infos)
            s.close()
        except Exception as e:
            errs.append(e)

    th = [threading.Thread(target=rank_fn, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(timeout=300) for t in th]
    assert not errs, errs
    return np.concatenate([o[0] for o in out]), np.concatenate([o[1] for o in out]), [o[2] for o in out]


Xn, Un, in1 = run_world_ncn(1)
print("full consensus, single rank (ipm, rounds, factorisations):", in1[0])
assert all(i[0] == 0 and i[1] == i[2] for i in in1[0][1:]), "a later solve was not warm-started"
for world in (2, 4):
    Xw, Uw, infos = run_world_ncn(world)
    eu = np.linalg.norm(Uw - Un) / np.linalg.norm(Un)
    print(f"full consensus, world {world}: rel diff U {eu:.2e}; {infos[0]}; ranks agree: {len({tuple(i) for i in infos}) == 1}")
    assert eu < 1e-9 and len({tuple(i) for i in infos}) == 1
    assert all(i[0] == 0 and i[1] == i[2] for i in infos[0][1:]), "a later sharded solve was not warm-started"
print("SHARDED_NCN_OK")
