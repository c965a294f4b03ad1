"""Checkpointed restart of the later rounds' factor sweeps (kernels_as.hip, option `as_ckpt`): an unsettled particle's sweep
starts at the lowest checkpoint at or above its highest changed stage instead of at the terminal cost.  The answers must not
depend on it: the same SCP loop with the option on and off, iteration by iteration, and the restarts must actually happen —
also from the higher rungs of the ladder (a reference that jumps in mid-horizon saturates controls there)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _loop(model, prob, steps, ck, Nc, xb=None):
    import torch

    from pmpc_amd.device import MODEL_QUADROTOR, MODEL_UNICYCLE, DeviceSolver, to_device_problem

    d = to_device_problem(prob)
    s = DeviceSolver(0)
    s.set_option("as_ckpt", ck)
    mid = MODEL_QUADROTOR if model == "quadrotor" else MODEL_UNICYCLE
    Xa, Ua = d["X_prev"].clone(), d["U_prev"].clone()
    Xb, Ub = torch.empty_like(Xa), torch.empty_like(Ua)
    out, rounds = [], 0
    kw = {} if xb is None else dict(lx=-xb, ux=xb)
    for it in range(steps):
        f, fx, fu = s.linearize(mid, d["x0"], Xa, Ua, d["params"])
        _, _, status = s.lqp_solve(f=f, fx=fx, fu=fu, X_prev=Xa, U_prev=Ua, Q=d["Q"], R=d["R"], X_ref=d["X_ref"], U_ref=d["U_ref"],
                                   reg_x=prob["reg_x"], reg_u=prob["reg_u"], Nc=Nc, x0=d["x0"], lu=d.get("lu"), uu=d.get("uu"), X_out=Xb, U_out=Ub,
                                   symmetric_cost=True, static_cons_bounds=True, prev_is_last_solution=it > 0, **kw)
        s.sync()
        assert status == 0
        rounds += s.last_info["active_set_rounds"]
        out.append((Xb.cpu().numpy().copy(), Ub.cpu().numpy().copy()))
        Xa, Xb, Ua, Ub = Xb, Xa, Ub, Ua
    st = s.restart_stats()
    s.close()
    return out, rounds, st


def _compare(a, b):
    worst = 0.0
    for (Xa, Ua), (Xb, Ub) in zip(a, b):
        worst = max(worst, np.abs(Xa - Xb).max() / max(1.0, np.abs(Xa).max()), np.abs(Ua - Ub).max() / max(1.0, np.abs(Ua).max()))
    return worst


@pytest.mark.parametrize("N,Nc,jump", [(50, 1, False), (50, 1, True), (20, 1, False), (100, 2, True), (40, 0, True)])
def test_restart_leaves_the_answers_alone_quadrotor(N, Nc, jump):
    from pmpc_amd import dynamics as dyn

    prob = dyn.make_quadrotor_problem(M=192, N=N, Nc=Nc)
    if jump:  # the position reference jumps at two thirds of the horizon: torques saturate around that stage
        prob["X_ref"][:, (2 * N) // 3:, 0] += 1.5
        prob["u_u"][..., 1:] = 0.2
        prob["u_l"][..., 1:] = -0.2
    off, r0, s0 = _loop("quadrotor", prob, 5, 0, Nc)
    on, r1, s1 = _loop("quadrotor", prob, 5, 1, Nc)
    assert r0 == r1 and r0 > 5, (r0, r1)  # several rounds per solve: there is something to restart
    assert s0["restarted"] == 0
    assert s1["restarted"] > 0, s1
    if jump:  # changes in mid-horizon: restarts from the higher rungs (more than the 5 stages of the first one on average), or none possible
        assert s1["restarted_stages"] > 5 * s1["restarted"] or s1["full"] > 0, s1
    assert _compare(off, on) < 1e-10


def test_restart_leaves_the_answers_alone_unicycle():
    """x4 u2 (one MFMA k-step per tile).  The unicycle's closed-form Jacobians amplify last-bit differences of the iterate by many
    orders (tests/test_device_gpu.py: n / u2^3 cancellations), so two independent loops drift apart whatever the solver does: here
    both contexts solve the SAME sub-problems in lock-step (the iterate of the context without restarts is handed to both)."""
    import torch

    from pmpc_amd import dynamics as dyn
    from pmpc_amd.device import MODEL_UNICYCLE, DeviceSolver, to_device_problem

    prob = dyn.make_unicycle_problem(M=64, N=30, Nc=1)
    d = to_device_problem(prob)
    sa, sb = DeviceSolver(0), DeviceSolver(0)
    sa.set_option("as_ckpt", 0)
    sb.set_option("as_ckpt", 1)
    Xp, Up = d["X_prev"].clone(), d["U_prev"].clone()
    worst, rounds = 0.0, 0
    for it in range(6):
        f, fx, fu = sa.linearize(MODEL_UNICYCLE, d["x0"], Xp, Up, d["params"])
        sa.sync()
        outs = []
        for s in (sa, sb):
            X, U, status = s.lqp_solve(f=f, fx=fx, fu=fu, X_prev=Xp, U_prev=Up, Q=d["Q"], R=d["R"], X_ref=d["X_ref"], U_ref=d["U_ref"],
                                       reg_x=prob["reg_x"], reg_u=prob["reg_u"], Nc=1, x0=d["x0"], lu=d.get("lu"), uu=d.get("uu"),
                                       symmetric_cost=True, static_cons_bounds=True, prev_is_last_solution=it > 0)
            s.sync()
            assert status == 0
            outs.append((X.cpu().numpy(), U.cpu().numpy()))
        assert sa.last_info["active_set_rounds"] == sb.last_info["active_set_rounds"]
        rounds += sb.last_info["active_set_rounds"]
        worst = max(worst, _compare([outs[0]], [outs[1]]))
        Xp, Up = X_a, U_a = torch.as_tensor(outs[0][0], device=Xp.device), torch.as_tensor(outs[0][1], device=Up.device)
    st = sb.restart_stats()
    sa.close()
    sb.close()
    assert rounds > 6 and st["restarted"] > 0, (rounds, st)
    assert worst < 1e-12, worst


@pytest.mark.parametrize("model,M,N,f32", [("quadrotor", 192, 50, False), ("quadrotor", 96, 20, True), ("unicycle", 64, 30, False)])
def test_elementwise_update_of_settled_particles_leaves_the_answers_alone(model, M, N, f32):
    """Option as_sens_min_m (k_fwd_as<.., SENS>): settled particles of the later rounds take the shared-control step elementwise from the
    forward sweep's sensitivity records.  Lock-step on identical sub-problems (see the unicycle test above), with the records on for
    any particle count against off; fp32-storage variant included."""
    import torch

    from pmpc_amd import dynamics as dyn
    from pmpc_amd.device import MODEL_QUADROTOR, MODEL_UNICYCLE, DeviceSolver, to_device_problem

    prob = dyn.make_quadrotor_problem(M=M, N=N, Nc=1) if model == "quadrotor" else dyn.make_unicycle_problem(M=M, N=N, Nc=1)
    mid = MODEL_QUADROTOR if model == "quadrotor" else MODEL_UNICYCLE
    d = to_device_problem(prob)
    sa, sb = DeviceSolver(0), DeviceSolver(0)
    sa.set_option("as_sens_min_m", 0)
    sb.set_option("as_sens_min_m", 1)
    Q, R = (d["Q"].float().contiguous(), d["R"].float().contiguous()) if f32 else (d["Q"], d["R"])
    Xp, Up = d["X_prev"].clone(), d["U_prev"].clone()
    worst, rounds = 0.0, 0
    for it in range(6):
        f, fx, fu = sa.linearize(mid, d["x0"], Xp, Up, d["params"])
        if f32:
            fx, fu = fx.float().contiguous(), fu.float().contiguous()
        sa.sync()
        outs = []
        for s in (sa, sb):
            X, U, status = s.lqp_solve(f=f, fx=fx, fu=fu, X_prev=Xp, U_prev=Up, Q=Q, R=R, X_ref=d["X_ref"], U_ref=d["U_ref"],
                                       reg_x=prob["reg_x"], reg_u=prob["reg_u"], Nc=1, x0=d["x0"], lu=d.get("lu"), uu=d.get("uu"),
                                       symmetric_cost=True, static_cons_bounds=True, prev_is_last_solution=it > 0)
            s.sync()
            assert status == 0
            outs.append((X.cpu().numpy(), U.cpu().numpy()))
        rounds += sb.last_info["active_set_rounds"]
        worst = max(worst, _compare([outs[0]], [outs[1]]))
        Xp, Up = torch.as_tensor(outs[0][0], device=Xp.device), torch.as_tensor(outs[0][1], device=Up.device)
    sa.close()
    sb.close()
    assert rounds > 6, rounds  # later rounds ran: settled particles went through the elementwise path
    assert worst < (1e-9 if f32 else 1e-12), worst


@pytest.mark.parametrize("soc", [False, True])
def test_unsettled_first_launch_order_and_cone_elementwise_path(soc):
    """(1) A later round's launches take the unsettled particles first (option as_perm_min_m, k_as_perm; only with the elementwise path, where
    the settled particles' waves are short): the workgroup -> particle map must not change a bit of the answer.  (2) The elementwise path
    with stage cones (the raw Newton steps the cone pass needs are written elementwise too) against sweeps, lock-step on identical
    sub-problems."""
    import torch

    from pmpc_amd import dynamics as dyn
    from pmpc_amd.device import MODEL_QUADROTOR, DeviceSolver, to_device_problem

    M, N = 160, 40
    prob = dyn.make_quadrotor_problem(M=M, N=N, Nc=1)
    d = to_device_problem(prob)
    dev = lambda a: torch.tensor(np.asarray(a, dtype=np.float64), device="cuda")
    kw = {}
    if soc:
        W = np.zeros((2, 4))
        W[0, 1] = W[1, 2] = 1.0
        kw = dict(soc_W=dev(W), soc_w0=dev(np.zeros(2)), soc_v=dev([0.3, 0, 0, 0]), soc_v0=0.0, soc_u_interior=dev([9.81, 0, 0, 0]))
    solvers = [DeviceSolver(0) for _ in range(3)]
    for s, (sens, perm) in zip(solvers, ((0, 0), (1, 0), (1, 1))):
        s.set_option("as_sens_min_m", sens)
        s.set_option("as_perm_min_m", perm)
    Xp, Up = d["X_prev"].clone(), d["U_prev"].clone()
    rounds = 0
    for it in range(5):
        f, fx, fu = solvers[0].linearize(MODEL_QUADROTOR, d["x0"], Xp, Up, d["params"])
        solvers[0].sync()
        outs = []
        for s in solvers:
            fn = s.lsoc_solve if soc else s.lqp_solve
            X, U, status = fn(f=f, fx=fx, fu=fu, X_prev=Xp, U_prev=Up, Q=d["Q"], R=d["R"], X_ref=d["X_ref"], U_ref=d["U_ref"], reg_x=prob["reg_x"], reg_u=prob["reg_u"],
                              Nc=1, x0=d["x0"], lu=d.get("lu"), uu=d.get("uu"), symmetric_cost=True, static_cons_bounds=True, prev_is_last_solution=it > 0, **kw)
            s.sync()
            assert status == 0
            outs.append((X.cpu().numpy(), U.cpu().numpy()))
        rounds += solvers[2].last_info["active_set_rounds"]
        assert np.array_equal(outs[1][0], outs[2][0]) and np.array_equal(outs[1][1], outs[2][1])  # launch order: bit for bit
        assert _compare([outs[0]], [outs[1]]) < 1e-11                                            # elementwise path against sweeps
        Xp, Up = torch.as_tensor(outs[0][0], device=Xp.device), torch.as_tensor(outs[0][1], device=Up.device)
    for s in solvers:
        s.close()
    assert rounds > 5
