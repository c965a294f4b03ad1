#!/usr/bin/env python3
"""Compact summary of bench.py logs: python tools/bench_line.py LOG [LOG ...]"""
import json
import sys

for f in sys.argv[1:]:
    try:
        line = [l for l in open(f).read().splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
    except Exception as e:  # noqa: BLE001
        tail = open(f).read()[-600:] if True else ""
        print(f, "NO JSON LINE:", repr(e), tail.replace("\n", " | ")[-400:])
        continue
    c, r = d["config"], d["roofline"]
    print(f, f"{d['value']:.1f} it/s", f"{d['ms_per_step']:.3f} ms", "rep", [round(v) for v in d["repeats"]["values"]],
          "rounds", c.get("active_set_rounds_per_step"), "ipm", c.get("ipm_iters_per_step"), "qps", c.get("weighted_qps_per_step"),
          "frac", round(r["frac"], 3), "per_solve", None if not r.get("per_solve") else round(r["per_solve"]["frac"], 3),
          {k: round(v, 3) for k, v in r["kernel_ms_per_step"].items()}, "resid", c.get("final_scp_residual"))
