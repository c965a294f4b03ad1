#!/usr/bin/env python3
"""Exploration (build container only): which reading of the reference reproduces the (obj, resid) tables stored in
/root/reference/tests/root_testing.ipynb etc.  Runs this repository's host SCP loop over the oracle."""
import json
import re
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import lqp_oracle as orc  # noqa: E402
from pmpc_amd import dynamics as dyn  # noqa: E402
import pmpc_amd.scp_mpc as scp  # noqa: E402

REF = Path("/root/reference")


def table_of(nb_path, needle, which=0):
    nb = json.load(open(nb_path))
    cells = [c for c in nb["cells"] if c["cell_type"] == "code" and needle in "".join(c["source"])]
    cell = cells[which]
    text = "".join("".join(o.get("text", [])) for o in cell["outputs"] if "text" in o)
    rows = [[float(v) for v in line.strip("| \n").split("|")] for line in text.splitlines() if re.match(r"\|\s*\d{4}", line)]
    rows = [r for r in rows if len(r) == 6]  # the CPU solver's table (the experimental JAX solver prints a 7th column)
    return np.array(rows)[:, [0, 2, 3, 4, 5]]


def run(args, kw, solve_one):
    def aff(f, fx, fu, x0, X_prev, U_prev, Q, R, X_ref, U_ref, reg_x, reg_u, slew_rate, u_slew, x_l, x_u, u_l, u_u, solver_settings=None, **_):
        X, U = solve_one(x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref, reg_x, reg_u, slew_rate, u_l, u_u)
        return np.concatenate([x0[:, None, :], X], 1), U, dict()
    scp.aff_solve = aff
    X, U, data = scp.scp_solve(*args, **kw)
    return np.array([[h["it"], h["obj"], h["resid"]] for h in data["hist"]])


def consensus_problem():
    M, N, xdim, udim = 20, 20, 4, 2
    t = lambda z: np.tile(z, (M,) + (1,) * z.ndim)
    Q, R = t(np.tile(np.eye(xdim), (N, 1, 1))), t(np.tile(1e-2 * np.eye(udim), (N, 1, 1)))
    x0 = t(np.ones(xdim))
    zx, zu = np.zeros((M, N, xdim)), np.zeros((M, N, udim))
    P = np.linspace(0.7, 1.0, M)
    P = (np.ones((1, N)) * P[:, None])[..., None]
    params = np.concatenate([P, P, P], -1)  # JAX clamps the out-of-range indices p[..., 1], p[..., 2] to 0

    def f_fx_fu_fn(X, U):
        return dyn.unicycle(X, U, params, eps=1e-6)

    args = (f_fx_fu_fn, Q, R, x0, zx, zu, zx.copy(), zu.copy())
    kw = dict(u_l=-np.ones((M, N, udim)), u_u=np.ones((M, N, udim)), reg_x=1.0, reg_u=1.0, max_it=23, res_tol=0.0, verbose=False, slew_rate=1e2)
    return args, kw


def show(name, got, table):
    n = min(len(got), len(table))
    eo = np.max(np.abs(got[:n, 1] / table[:n, 1] - 1))
    er = np.max(np.abs(got[:n, 2] / table[:n, 2] - 1))
    print(f"{name:50s} obj max rel dev {eo:.2e}  resid max rel dev {er:.2e}   first rows got obj {got[:3,1]} resid {got[:3,2]}")


if __name__ == "__main__":
    orc.build()
    tab = table_of(REF / "tests" / "root_testing.ipynb", "X, U, data = solve(**problem)")
    print(tab[:3])
    args, kw = consensus_problem()
    for slew in (1e2, 0.0):
        for name, fn in {
            "qp hard": lambda x0, f, fx, fu, Xp, Up, Q, R, Xr, Ur, rx, ru, s, ul, uu: orc.lqp_solve_py(x0, f, fx, fu, Xp, Up, Q, R, Xr, Ur, reg_x=rx, reg_u=ru, Nc=5, u_l=ul, u_u=uu, slew_reg=s if s else None),
            "qp barrier mu=1": lambda x0, f, fx, fu, Xp, Up, Q, R, Xr, Ur, rx, ru, s, ul, uu: orc.lqp_solve_py(x0, f, fx, fu, Xp, Up, Q, R, Xr, Ur, reg_x=rx, reg_u=ru, Nc=5, u_l=ul, u_u=uu, slew_reg=s if s else None, barrier_mu=1.0),
            "cone hard": lambda x0, f, fx, fu, Xp, Up, Q, R, Xr, Ur, rx, ru, s, ul, uu: orc.lcone_solve_py(x0, f, fx, fu, Xp, Up, Q, R, Xr, Ur, reg_x=rx, reg_u=ru, Nc=5, u_l=ul, u_u=uu, slew_reg=s if s else None),
            "cone barrier alpha=1": lambda x0, f, fx, fu, Xp, Up, Q, R, Xr, Ur, rx, ru, s, ul, uu: orc.lcone_solve_py(x0, f, fx, fu, Xp, Up, Q, R, Xr, Ur, reg_x=rx, reg_u=ru, Nc=5, u_l=ul, u_u=uu, slew_reg=s if s else None, smooth_alpha=1.0),
        }.items():
            kw2 = dict(kw, slew_rate=slew)
            try:
                got = run(args, kw2, fn)
                show(f"slew={slew:g} {name}", got, tab)
            except Exception as e:  # noqa: BLE001
                print(f"slew={slew:g} {name}: FAILED {type(e).__name__}: {e}")


def detail(name, got, table):
    n = min(len(got), len(table))
    print(name)
    for k in range(n):
        print(f"  {int(table[k,0]):3d} obj {table[k,1]:.3e} got {got[k,1]:.5e} ({got[k,1]/table[k,1]-1:+.1e})   resid {table[k,2]:.3e} got {got[k,2]:.5e} ({got[k,2]/table[k,2]-1:+.1e})")
