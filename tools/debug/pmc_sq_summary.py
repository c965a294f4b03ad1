"""Per-kernel means of the counters collected by tools/debug/pmc_sq.sh (the sweeps and the streaming kernels only)."""
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if not any(s in k for s in ("k_bwd_as", "k_fwd_as", "k_linearize", "k_scp_residual")):
        continue
    d = {c: sum(v) / len(v) for c, v in acc[k].items()}
    n = len(next(iter(acc[k].values())))
    print(f"{k[:110]}  ({n} launches)")
    print("   " + "  ".join(f"{c}={v:.4g}" for c, v in sorted(d.items())))
    wc = d.get("SQ_WAVE_CYCLES")
    if wc:
        print("   of wave cycles: " + "  ".join(f"{c[3:]} {100 * d[c] / wc:.1f}%" for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA") if c in d))
