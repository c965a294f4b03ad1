#!/usr/bin/env python3
"""Where the cycles of ONE stage of the two sweeps go (VERDICT r03, next #2a).

Runs bench.py's SCP loop on the DIAGNOSTIC build (make -C pmpc_amd/csrc VARIANT=tl EXTRA="-DPMPC_STAGE_TIMELINE
-DPMPC_DIAG_DIMS_12_4" -> libs_tmp/libpmpc_hip_tl.so), whose k_bwd_as / k_fwd_as stamp s_memtime at their phase boundaries for the
wave in the middle of the grid, and prints the mean cycles per phase over the branch-free main stages of the LAST launch of each
sweep.  One process per particle count (512: one wave per two SIMDs — the latency regime; 4096: four waves per SIMD).

    python tools/micro/stage_timeline.py [M ...]      (on the GPU box; writes gpurun_out/stage_timeline.txt as well)
"""
import ctypes
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
STAMPS = 10
BWD = ["entry -> control word decoded, operands landed", "issue next stage's loads", "gradient h = F'(s + S r), control rows read out",
       "H init + 6 MFMAs issued", "Cholesky of Huu (first read waits for the MFMA chain)", "gather + substitution (gains)",
       "S' MFMA issued, record + feed-forward stored", "next gradient s (2 reductions + lane shuffles)"]
FWD = ["entry -> 3 MFMAs issued (waits for this stage's loads)", "control word gathered", "decisions (first use waits for the MFMA chain)",
       "control / status stores", "B du MFMA, new state stored"]


def child(M):
    os.environ["PMPC_HIP_LIB"] = str(ROOT / "libs_tmp" / "libpmpc_hip_tl.so")
    sys.path.insert(0, str(ROOT))
    sys.argv = ["bench.py", "--steps", "8", "--warmup", "3", "--repeats", "0", "--no-cpu-baseline", "--M", str(M)]
    import io
    import contextlib

    import bench

    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    import json

    line = [l for l in buf.getvalue().splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    from pmpc_amd import _lib

    lib = _lib.load()
    raw = (ctypes.c_ulonglong * (3 * 128 * STAMPS))()
    lib.pmpc_debug_timeline_read.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
    rc = lib.pmpc_debug_timeline_read(raw)
    assert rc == 0, rc
    import numpy as np

    t = np.array(raw, dtype=np.uint64).reshape(3, 128, STAMPS).astype(np.int64)
    N = 50
    out = [f"== M = {M} particles ({M / 1024:g} waves per SIMD), instrumented build: {d['value']:.0f} it/s, {d['ms_per_step']:.3f} ms/step, "
           f"kernel ms/step {({k: round(v, 3) for k, v in d['roofline']['kernel_ms_per_step'].items()})}"]
    for kind, name, labels, nst in ((0, "k_bwd_as (full factor sweep, last launch)", BWD, 9), (1, "k_bwd_as SKIP (partial factor sweep, last launch)", BWD, 9),
                                    (2, "k_fwd_as (last launch)", FWD, 6)):
        a = t[kind]
        js = [j for j in range(3, N - 3) if a[j, 0] > 0 and a[j, nst - 1] > 0]
        if not js:
            out.append(f"-- {name}: no stamps (the wave in the middle of the grid did not run this variant)")
            continue
        seg = np.array([[a[j, k + 1] - a[j, k] for k in range(nst - 1)] for j in js], dtype=np.float64)
        # stage period: entry to entry of consecutive stages (bwd runs j downwards, fwd upwards)
        if kind == 2:
            per = np.array([a[j + 1, 0] - a[j, 0] for j in js if a[j + 1, 0] > 0], dtype=np.float64)
        else:
            per = np.array([a[j - 1, 0] - a[j, 0] for j in js if a[j - 1, 0] > 0], dtype=np.float64)
        out.append(f"-- {name}: {len(js)} main stages; stage period {per.mean():.0f} cycles (min {per.min():.0f}, max {per.max():.0f}); whole sweep "
                   f"{(a[:N, :nst].max() - a[:N, :nst][a[:N, :nst] > 0].min())} cycles")
        for k in range(nst - 1):
            out.append(f"   {seg[:, k].mean():8.0f} cyc ({100 * seg[:, k].mean() / per.mean():4.1f} %)  {labels[k]}")
        out.append(f"   {per.mean() - seg.sum(1).mean():8.0f} cyc ({100 * (per.mean() - seg.sum(1).mean()) / per.mean():4.1f} %)  between stages (loop overhead, register rotation, the stamps' own stores)")
    print("\n".join(out))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(int(sys.argv[2]))
        sys.exit(0)
    Ms = [int(v) for v in sys.argv[1:]] or [512, 4096]
    text = []
    for M in Ms:
        r = subprocess.run([sys.executable, __file__, "--child", str(M)], capture_output=True, text=True)
        text.append(r.stdout if r.returncode == 0 else f"M={M}: failed\n{r.stdout}\n{r.stderr[-3000:]}")
    s = "\n".join(text)
    print(s)
    od = ROOT / "gpurun_out"
    od.mkdir(exist_ok=True)
    (od / "stage_timeline.txt").write_text(s)
