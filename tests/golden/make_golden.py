#!/usr/bin/env python3
"""Generates tests/golden/*.npz.  Runs ONLY in the build container (needs /root/reference).

What is pinned and by what:
  * cone-path fixtures (`cone_*.npz`): the same problem definitions, minimiser of the reference's
    epsilon-anchored epigraph objective (oracle.lcone_solve_py) with the particle weights it implies.
  * sub-problem fixtures (`qp_*.npz`): the reference's own example / test problem definitions
    (restated here from the cited lines, numpy-seeded where the reference draws random numbers),
    solved by the oracle (oracle/lqp_oracle.py) whose KKT certificate is stored alongside.
  * SCP fixtures (`scp_*.npz`): the REFERENCE's own Python SCP loop, imported from
    /root/reference (pmpc/scp_mpc.py:205-442) with its torch unicycle (tests/dubins_car.py),
    run with `pmpc.scp_mpc.aff_solve` replaced by the oracle (the loop looks it up as a module
    global, scp_mpc.py:370; Julia itself cannot run here).  Stored: final X, U and every `hist` row.

  * `ref_notebook_cpu_table.npz`: the only numeric output of the reference's own solver stack on this path — the
    50-row (obj, resid) table stored in examples/gpu_solver.ipynb (Julia + ECOS run by the authors, 4 digits).

Apart from that table the reference holds no numeric golden vectors for this path: the other
files pin the oracle + host loop against the reference's loop, not against Julia/OSQP output.
"""
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
REF = Path("/root/reference")

from oracle import lqp_oracle as orc  # noqa: E402


def save(name, **kw):
    np.savez_compressed(HERE / name, **kw)
    print("wrote", name, {k: np.shape(v) for k, v in kw.items()})


def solve_and_pack(args, kw, Nc):
    X, U, info = orc.lqp_solve_py(*args, Nc=Nc, return_info=True, **kw)
    names = ["x0", "f", "fx", "fu", "X_prev", "U_prev", "Q", "R", "X_ref", "U_ref"]
    pack = dict(zip(names, args))
    pack.update({k: np.asarray(v, float) for k, v in kw.items()})
    pack.update(Nc=np.array(Nc), X=X, U=U, cert=np.array([info["cert"][k] for k in ("stationarity", "equality", "bound_violation", "complementarity")]))
    return pack


# ---- 1. canonical ABI example: tests/pmpcjl_test.py:164-219 (and the u in +-1 variant of
#         PMPC.jl/src/c_precompile.jl:7-49) --------------------------------------------------------------
def double_integrator(u_limit):
    M, N, xdim, udim, Nc = 1, 30, 2, 1, 3
    x0 = np.array([5.0, 5.0])
    A = np.array([[1.0, 0.1], [0.0, 1.0]])
    fx = np.tile(A, (M, N, 1, 1))
    fu = np.tile(np.array([[0.0], [1.0]]), (M, N, 1, 1))
    f = np.zeros((M, N, xdim))
    f[:, 0] = A @ x0
    zx, zu = np.zeros((M, N, xdim)), np.zeros((M, N, udim))
    Q = np.tile(np.eye(2), (M, N, 1, 1))
    R = np.tile(np.eye(1), (M, N, 1, 1))
    kw = dict(reg_x=1.0, reg_u=0.1, u_l=-u_limit * np.ones((M, N, udim)), u_u=u_limit * np.ones((M, N, udim)),
              x_l=-20.0 * np.ones((M, N, xdim)), x_u=20.0 * np.ones((M, N, xdim)), slew_reg=np.ones(M), slew_reg0=np.zeros(M),
              slew_um1=np.zeros((M, udim)))
    args = (x0[None], f, fx, fu, zx, zu, Q, R, zx.copy(), zu.copy())
    return args, kw, Nc


# ---- 3. chain problem of PMPC.jl/test/test.jl:334-377 (Julia's RNG stream restated with numpy) ----------
def chain(rng, M=100, N=30):
    xdim, udim = 4, 2
    x0 = np.full(4, 5.0)
    fx, fu, f = np.zeros((M, N, 4, 4)), np.zeros((M, N, 4, 2)), np.zeros((M, N, 4))
    for i in range(M):
        r1, r2 = rng.choice(np.linspace(0.1, 0.9, 10), 2)
        A = np.array([[1, r1, 0, 0], [0, 1, 0, 0], [0, 0, 1, r2], [0, 0, 0, 1.0]])
        fx[i], fu[i] = A, np.array([[0, 0], [1, 0], [0, 0], [0, 1.0]])
        f[i, 0] = A @ x0
    zx, zu = np.zeros((M, N, xdim)), np.zeros((M, N, udim))
    Q, R = np.tile(np.eye(4), (M, N, 1, 1)), np.tile(np.eye(2), (M, N, 1, 1))
    kw = dict(reg_x=0.0, reg_u=0.0, u_l=-np.ones((M, N, udim)), u_u=np.ones((M, N, udim)))
    return (np.tile(x0, (M, 1)), f, fx, fu, zx, zu, Q, R, zx.copy(), zu.copy()), kw


# ---- 4. random SPD problem of PMPC.jl/test/runtests.jl:6-27 ----------------------------------------------
def random_spd(rng):
    M, N, xdim, udim = 3, 11, 4, 2
    g = rng.standard_normal
    x0, f = g((M, xdim)), g((M, N, xdim))
    fx, fu = g((M, N, xdim, xdim)), g((M, N, xdim, udim))
    X_prev, U_prev = g((M, N, xdim)), g((M, N, udim))
    Qh, Rh = g((M, N, xdim, xdim)), g((M, N, udim, udim))
    Q, R = np.swapaxes(Qh, -1, -2) @ Qh, np.swapaxes(Rh, -1, -2) @ Rh
    X_ref, U_ref = g((M, N, xdim)), g((M, N, udim))
    # runtests.jl:24-27 also sets |x| <= 100, which this random unstable fx makes infeasible (the
    # reference only asserts "does not throw" there); the goldens keep the solvable variants:
    # unconstrained (runtests.jl:33-35) and |u| <= 0.2 only.
    kw = dict(reg_x=0.0, reg_u=0.0, u_l=-0.2 * np.ones((M, N, udim)), u_u=0.2 * np.ones((M, N, udim)))
    return (x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref), kw


def cone_pack(args, kw, Nc):
    """cone-path fixture: the same problem data, minimiser of the epsilon-anchored epigraph objective
    (PMPC.jl/src/main.jl:194-354 through the C ABI, hard boxes) by oracle.lcone_solve_py"""
    X, U, info = orc.lcone_solve_py(*args, Nc=Nc, return_info=True, **kw)
    names = ["x0", "f", "fx", "fu", "X_prev", "U_prev", "Q", "R", "X_ref", "U_ref"]
    pack = dict(zip(names, args))
    pack.update({k: np.asarray(v, float) for k, v in kw.items()})
    pack.update(Nc=np.array(Nc), X=X, U=U, weights=info["weights"], J=info["J"],
                cert=np.array([info["qp"]["cert"][k] for k in ("stationarity", "equality", "bound_violation", "complementarity")]))
    return pack


def scp_reference_run(N, reg_x, reg_u, max_it):
    """tests/simple.py:20-29 (N=25, default regs) / tests/remote.py:20-39 (N=30, reg 1/1) through the
    REFERENCE scp_solve, oracle as aff_solve."""
    sys.path.insert(0, str(REF))
    sys.path.insert(0, str(REF / "tests"))
    import torch

    torch.set_default_dtype(torch.float64)
    import pmpc.scp_mpc as ref_scp
    from dubins_car import f_np, fu_np, fx_np

    ref_scp.aff_solve = orc.aff_solve
    xdim, udim = 4, 2

    def f_fx_fu_fn(X, U):
        p = np.array([1.0, 1.0, 0.3])
        return f_np(X, U, p), fx_np(X, U, p), fu_np(X, U, p)

    Q = np.tile(np.eye(xdim), (N, 1, 1))
    R = np.tile(1e-2 * np.eye(udim), (N, 1, 1))
    x0 = np.ones(xdim)
    X_ref, U_ref = np.zeros((N, xdim)), np.zeros((N, udim))
    u_l, u_u = -np.ones((N, udim)), np.ones((N, udim))
    kw = dict(u_l=u_l, u_u=u_u, max_it=max_it, solver_settings=dict(solver="osqp"))
    if reg_x is not None:
        kw.update(reg_x=reg_x, reg_u=reg_u)
    X, U, data = ref_scp.scp_solve(f_fx_fu_fn, Q, R, x0, X_ref, U_ref, X_ref.copy(), U_ref.copy(), **kw)
    hist = np.array([[h["it"], h["obj"], h["resid"], h["reg_x"], h["reg_u"]] for h in data["hist"]])
    return dict(X=X, U=U, hist=hist, N=np.array(N), reg_x=np.array(hist[0, 3]), reg_u=np.array(hist[0, 4]), max_it=np.array(max_it))


def notebook_table():
    """The one numeric output of the REFERENCE ITSELF that this path has: the table
    printed by `pmpc.solve` in examples/gpu_solver.ipynb, cell "CPU version for a quick check" (50 SCP iterations,
    4 significant digits): unicycle with eps = 1e-3 (the notebook's `car`), N = 20, M = 1, Q = I, R = 1e-2 I, x0 = 1,
    refs 0, |u| <= 1, reg_x = 3, reg_u = 1, default solver "ecos" => the cone path.  The rows are data transcribed by
    this function from the notebook's stored output; columns it, obj, resid, reg_x, reg_u."""
    import json
    import re

    nb = json.load(open(REF / "examples" / "gpu_solver.ipynb"))
    cell = next(c for c in nb["cells"] if c["cell_type"] == "code" and "X2, U2, data = pmpc.solve(**cpu_args, **cpu_opts)" in "".join(c["source"]))
    text = "".join("".join(o.get("text", [])) for o in cell["outputs"] if "text" in o)
    rows = [[float(v) for v in line.strip("| \n").split("|")] for line in text.splitlines() if re.match(r"\|\s*\d{4}", line)]
    rows = np.array(rows)[:, [0, 2, 3, 4, 5]]
    assert rows.shape == (50, 5)
    return dict(table=rows, N=np.array(20), reg_x=np.array(3.0), reg_u=np.array(1.0), car_eps=np.array(1e-3), u_lim=np.array(1.0),
                R_diag=np.array(1e-2), params=np.array([1.0, 1.0, 0.3]))


def _cpu_rows(nb_name, needle, which=0):
    """(it, obj, resid, reg_x, reg_u) rows of the table `pmpc.solve(verbose=True)` left in a committed notebook (the CPU
    solver prints 6 columns; the experimental JAX solver's 7-column tables in the same cells are skipped)."""
    import json
    import re

    nb = json.load(open(REF / nb_name))
    cell = [c for c in nb["cells"] if c["cell_type"] == "code" and needle in "".join(c["source"])][which]
    text = "".join("".join(o.get("text", [])) for o in cell["outputs"] if "text" in o)
    rows = [[float(v) for v in line.strip("| \n").split("|")] for line in text.splitlines() if re.match(r"\|\s*\d{4}", line)]
    return np.array([r for r in rows if len(r) == 6])[:, [0, 2, 3, 4, 5]]


def more_notebook_tables():
    """Further outputs of the reference's own solver stack (Julia + ECOS / JuMP, run by the authors) stored in committed
    notebooks, with the problem each one was printed for (restated from the cited cells; all: unicycle `car`, eps = 1e-6,
    x0 = 1, Q = I, X_prev = U_prev = 0, U_ref = 0).  `params` is (v_scale, w_scale, T) as the dynamics saw them:
      * root_testing single: `P = ones(N)` is indexed `p[..., 0..2]` -> (1, 1, 1);
      * root_testing consensus: `P` has shape (M, N, 1) and JAX clamps the out-of-range indices 1, 2 -> all three = P_i;
      * logbarrier_tests: `np.array([0.3, 1.0, 1.0])` handed to a dynamics module that is not in the repository
        (goal_oriented_driving); read in the order of tests/dubins_car.py:60."""
    M = 20
    Pc = np.linspace(0.7, 1.0, M)
    common = dict(car_eps=np.array(1e-6), x_ref=np.array(0.0), lin_cost_xref=np.array(np.nan), Nc=np.array(-1), M=np.array(1),
                  slew_rate=np.array(0.0))
    return {
        # tests/root_testing.ipynb cells 3-4: M = 1, slew_rate 1e2, solver "ecos", smooth_alpha 1e-1, max_it 20
        "ref_root_testing_single.npz": dict(common, table=_cpu_rows("tests/root_testing.ipynb", "X, U, _ = solve(**dict(problem"),
                                            N=np.array(20), R_diag=np.array(1e-2), u_lim=np.array(1.0), reg_x=np.array(1.0),
                                            reg_u=np.array(1.0), slew_rate=np.array(1e2), smooth_alpha=np.array(1e-1),
                                            params=np.array([[1.0, 1.0, 1.0]])),
        # tests/root_testing.ipynb cells 10-11 ("Test consensus optmization"): M = 20, Nc = 5, slew_rate 1e2, smooth_alpha 1
        "ref_root_testing_consensus.npz": dict(common, table=_cpu_rows("tests/root_testing.ipynb", "X, U, data = solve(**problem)"),
                                               N=np.array(20), M=np.array(M), Nc=np.array(5), R_diag=np.array(1e-2), u_lim=np.array(1.0),
                                               reg_x=np.array(1.0), reg_u=np.array(1.0), slew_rate=np.array(1e2),
                                               smooth_alpha=np.array(1.0), params=np.stack([Pc, Pc, Pc], -1)),
        # tests/logbarrier_tests.ipynb cell 3: M = 1, N = 30, R = I, |u| <= 0.2, reg 1e-1 / 1e-2, logbarrier alpha 1e-1
        "ref_logbarrier_tests.npz": dict(common, table=_cpu_rows("tests/logbarrier_tests.ipynb", "X, U, data = pmpc.solve(*args, max_it=100, **opts)"),
                                         N=np.array(30), R_diag=np.array(1.0), u_lim=np.array(0.2), reg_x=np.array(1e-1),
                                         reg_u=np.array(1e-2), smooth_alpha=np.array(1e-1), params=np.array([[0.3, 1.0, 1.0]])),
        # tests/experimental.ipynb cell 11 ("CPU version"): warm-started from a JAX-solver result that is not stored, so only
        # the FIXED POINT the table converges to is reproducible (obj 1.617; the hard-constrained one prints 1.616)
        "ref_experimental_cpu.npz": dict(common, table=_cpu_rows("tests/experimental.ipynb", "X2, U2, data = pmpc.solve("),
                                         N=np.array(20), R_diag=np.array(1e-2), u_lim=np.array(1.0), reg_x=np.array(10.0),
                                         reg_u=np.array(1.0), smooth_alpha=np.array(1e3), params=np.array([[1.0, 1.0, 0.3]])),
        # tests/demo_cost_jax.ipynb cell 3 (solve_cpu leg): R = 0, X_ref = 0.4, default regs 1 / 1e-2, logbarrier alpha 1e3
        "ref_demo_cost_convex.npz": dict(common, table=_cpu_rows("tests/demo_cost_jax.ipynb", "## Convex Cost"),
                                         N=np.array(30), R_diag=np.array(0.0), u_lim=np.array(1.0), reg_x=np.array(1.0),
                                         reg_u=np.array(1e-2), smooth_alpha=np.array(1e3), x_ref=np.array(0.4),
                                         params=np.array([[1.0, 1.0, 0.3]])),
        # tests/demo_cost_jax.ipynb cell 13 (solve_cpu leg): the same cost through lin_cost_fn (float32 gradient X - 0.4),
        # X_ref = 0, regs 3 / 1, smooth_alpha 1e1
        "ref_demo_cost_external.npz": dict(common, table=_cpu_rows("tests/demo_cost_jax.ipynb", "## External Cost"),
                                           N=np.array(30), R_diag=np.array(0.0), u_lim=np.array(1.0), reg_x=np.array(3.0),
                                           reg_u=np.array(1.0), smooth_alpha=np.array(1e1), lin_cost_xref=np.array(0.4),
                                           params=np.array([[1.0, 1.0, 0.3]])),
    }


if __name__ == "__main__":
    orc.build()
    save("ref_notebook_cpu_table.npz", **notebook_table())
    for name, pack in more_notebook_tables().items():
        save(name, **pack)
    if "--tables-only" in sys.argv:
        raise SystemExit(0)
    for name, ul in (("qp_double_integrator_u04.npz", 0.4), ("qp_double_integrator_u1.npz", 1.0)):
        args, kw, Nc = double_integrator(ul)
        save(name, **solve_and_pack(args, kw, Nc))
    rng = np.random.default_rng(2020)  # cf. Random.seed!(2020), PMPC.jl/test/test.jl:4
    args, kw = chain(rng, M=24, N=30)
    for Nc in (0, 1, 3, -1):
        save(f"qp_chain_Nc{Nc if Nc >= 0 else 'N'}.npz", **solve_and_pack(args, kw, Nc))
    for Nc in (1, -1):
        save(f"cone_chain_Nc{Nc if Nc >= 0 else 'N'}.npz", **cone_pack(args, kw, Nc))
    args1, kw1, Nc1 = double_integrator(0.4)
    save("cone_double_integrator_u04.npz", **cone_pack(args1, kw1, Nc1))
    args, kw = random_spd(np.random.default_rng(2021))
    save("qp_random_spd_ubox.npz", **solve_and_pack(args, kw, -1))
    save("qp_random_spd.npz", **solve_and_pack(args, dict(reg_x=0.0, reg_u=0.0), -1))
    # state boxes that BIND (no reference test or notebook uses binding state boxes: pinned by the oracle's KKT certificate only)
    from tests.support.problems import xbox_problem

    args, kw = xbox_problem(np.random.default_rng(2022), orc, 5, 9, 4, 2, 1, 0.4, pull=0.9, margin=0.02)
    save("qp_xbox_binding.npz", **solve_and_pack(args, kw, 1))
    save("scp_unicycle_simple.npz", **scp_reference_run(25, None, None, 40))
    save("scp_unicycle_remote.npz", **scp_reference_run(30, 1.0, 1.0, 40))
