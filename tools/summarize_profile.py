#!/usr/bin/env python3
"""Summarise a tools/profile_gpu.sh run into profiles/: kernel stats table, PMC HBM traffic per launch of
the dominant kernel (gfx950 corrections per /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE
are in KiB; FETCH_SIZE under-reports wide coalesced reads, so the read-side factor is CALIBRATED in the same
run on k_axpy, an 8-B-per-lane streaming kernel with an exactly known byte count)."""
import csv
import glob
import json
import sys
from collections import defaultdict
from pathlib import Path

tag = sys.argv[1]
src = Path("gpurun_out") / f"prof_{tag}"
dst = Path("profiles")
dst.mkdir(exist_ok=True)


def pmc(kind):
    f = glob.glob(str(src / f"pmc_{kind}" / "*" / "*counter_collection.csv"))[0]
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


fetch, write = pmc("fetch"), pmc("write")
M, N, x, u = 4096, 50, 12, 4
nx = M * N * x
key = lambda d, s: next(k for k in d if s in k)
ax = key(fetch, "k_axpy(")  # X += dX after the equality-only solve: 2 streams in, 1 out, nx doubles each
nu = M * N * u
ax_read_true = 2 * 8 * (nx + nu) / 2  # y and x streams; k_axpy runs once on X (nx) and once on U (nu) per solve
ax_fetch = sum(fetch[ax]) / len(fetch[ax]) * 1024
read_factor = ax_read_true / ax_fetch
ax_write = sum(write[key(write, "k_axpy(")]) / len(write[key(write, "k_axpy(")]) * 1024
out = {"profile": f"profiles/{tag}_kernel_stats.csv / rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE passes of tools/profile_gpu.sh {tag}",
       "calibration": {"kernel": "k_axpy", "true_read_bytes": ax_read_true, "FETCH_SIZE_bytes": ax_fetch,
                       "read_factor": read_factor, "true_write_bytes": 8 * (nx + nu) / 2, "WRITE_SIZE_bytes": ax_write}}
# kernels_as.hip: k_bwd_as<x, u, MODE, SKIP, DEFECT> (MODE 0 lean / 1 deep / 2 deep2; SKIP launches process a subset of the particles:
# never part of the roofline figure), k_fwd_as<x, u, DEFECT, PF2>; kernels_fast.hip: k_bwd_fast<x, u, FACTOR, HXB, HUB, DEEP>
# (since r03 the names end in the EX / storage-type arguments: <.., 0, double> = no cone / state-box terms, fp64 storage)
# The instantiations bench.py's timed region runs at config D: the first round's factor sweep is k_bwd_as<12, 4, 1, false, true>
# (deep, DEFECT) — `bwd_factor_defect`, the kernel `roofline.achieved` is measured on —, the forward sweeps k_fwd_as<12, 4, *, true>
# (PF2).  Every match of a prefix is pooled (the MODE / PF2 arguments depend on the particle count).
def pooled(d, prefixes):
    vals = []
    names = []
    for k in d:
        if any(p in k for p in prefixes):
            vals += d[k]
            names.append(k)
    if not vals:
        raise StopIteration
    return vals, names


for prefixes, label in ((("k_bwd_fast<12, 4, true, false, true, false>",), "bwd_factor"),
                        (("k_bwd_as<12, 4, 1, false, true, 0, double>", "k_bwd_as<12, 4, 0, false, true, 0, double>"), "bwd_factor_defect"),
                        (("k_bwd_as<12, 4, 1, false, false, 0, double>", "k_bwd_as<12, 4, 0, false, false, 0, double>"), "bwd_factor_as_plain"),
                        (("k_bwd_fast<12, 4, false",), "bwd_vec"), (("k_fwd_fast<12, 4, false>",), "fwd"),
                        (("k_fwd_as<12, 4, false, true, false, double>", "k_fwd_as<12, 4, false, false, false, double>"), "fwd_active_set"),
                        (("k_fwd_as<12, 4, true, true, false, double>", "k_fwd_as<12, 4, true, false, false, double>"), "fwd_active_set_defect")):
    try:
        fv, fnames = pooled(fetch, prefixes)
        wv, _ = pooled(write, prefixes)
    except StopIteration:
        continue
    fb = sum(fv) / len(fv) * 1024
    wb = sum(wv) / len(wv) * 1024
    out[label] = {"kernels": fnames, "FETCH_SIZE_bytes_raw": fb, "WRITE_SIZE_bytes": wb, "read_bytes_calibrated": fb * read_factor,
                  "launches_sampled": len(fv)}
    out[f"{label}_bytes_per_launch"] = fb * read_factor + wb
# r04: workloads with PMC passes of their own (tools/profile_gpu.sh): dominant kernel = the first round's factor sweep of the timed region
workloads = {}
for name, prefixes in (("E_soc", ("k_bwd_as<12, 4, 1, false, true, 1, double>", "k_bwd_as<12, 4, 0, false, true, 1, double>")),
                       ("D_fp32", ("k_bwd_as<12, 4, 1, false, true, 0, float>",)),
                       ("E_soc_fp32", ("k_bwd_as<12, 4, 1, false, true, 1, float>",))):
    try:
        ff = glob.glob(str(src / f"pmc_fetch_{name}" / "*" / "*counter_collection.csv"))[0]
        fw = glob.glob(str(src / f"pmc_write_{name}" / "*" / "*counter_collection.csv"))[0]
    except IndexError:
        continue
    accf, accw = defaultdict(list), defaultdict(list)
    for r in csv.DictReader(open(ff)):
        accf[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for r in csv.DictReader(open(fw)):
        accw[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    try:
        fv, fnames = pooled(accf, prefixes)
        wv, _ = pooled(accw, prefixes)
    except StopIteration:
        continue
    fb, wb = sum(fv) / len(fv) * 1024, sum(wv) / len(wv) * 1024
    workloads[name] = {"kernels": fnames, "FETCH_SIZE_bytes_raw": fb, "WRITE_SIZE_bytes": wb, "read_bytes_calibrated": fb * read_factor,
                       "bytes_per_launch": fb * read_factor + wb, "launches_sampled": len(fv),
                       "profile": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/profile_gpu.sh {tag} on this workload; read factor calibrated on k_axpy in the config-D pass"}
out["workloads"] = workloads
(dst / "traffic.json").write_text(json.dumps(out, indent=1))
(dst / f"{tag}_traffic.json").write_text(json.dumps(out, indent=1))
print(json.dumps(out, indent=1))

stats = glob.glob(str(src / "stats" / "*" / "*kernel_stats.csv"))[0]
(dst / f"{tag}_kernel_stats.csv").write_text(open(stats).read())
for log, name in (("bench_full.log", f"{tag}_bench.json"), ("bench_stats.log", f"{tag}_bench_under_rocprof.json"),
                  ("bench_C.log", f"{tag}_bench_C.json"), ("bench_B.log", f"{tag}_bench_B.json"),
                  ("bench_B_NcN.log", f"{tag}_bench_B_NcN.json"), ("bench_D_NcN.log", f"{tag}_bench_D_NcN.json"),
                  ("bench_shard_512.log", f"{tag}_bench_shard_512.json"), ("bench_shard_2048.log", f"{tag}_bench_shard_2048.json"),
                  ("bench_E_soc.log", f"{tag}_bench_E_soc.json"), ("bench_E_soc_fp32.log", f"{tag}_bench_E_soc_fp32.json"),
                  ("bench_D_fp32.log", f"{tag}_bench_D_fp32.json"), ("bench_D_cone.log", f"{tag}_bench_D_cone.json"),
                  ("bench_B_cone.log", f"{tag}_bench_B_cone.json"), ("bench_B_cone_smooth.log", f"{tag}_bench_B_cone_smooth.json"),
                  ("bench_D_vmax3.0.log", f"{tag}_bench_D_vmax3.json"), ("bench_D_vmax2.0.log", f"{tag}_bench_D_vmax2.json"),
                  ("bench_D_vmax3.0_ipm.log", f"{tag}_bench_D_vmax3_ipm.json"), ("bench_D_cone_smooth.log", f"{tag}_bench_D_cone_smooth.json")):
    if not (src / log).exists():
        continue
    lines = [l for l in open(src / log) if l.startswith("{")]
    if lines:
        (dst / name).write_text(lines[-1])

for extra in ("slew_paths.txt", "xbox_summary.txt", "stage_timeline.txt"):
    if (src / extra).exists():
        (dst / f"{tag}_{extra}").write_text(open(src / extra).read())
soc_stats = glob.glob(str(src / "stats_E_soc" / "*" / "*kernel_stats.csv"))
if soc_stats:
    (dst / f"{tag}_kernel_stats_E_soc.csv").write_text(open(soc_stats[0]).read())
