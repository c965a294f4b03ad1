"""Smoothed cone objective at config D on a diagnostic (12,4)-only build: verbose first solves (debugging aid)."""
import os, sys
os.environ.setdefault("PMPC_HIP_LIB", "libs_tmp/libpmpc_hip_d.so")
sys.path.insert(0, ".")
sys.argv = ["bench.py", "--cone", "--smooth-alpha", sys.argv[1] if len(sys.argv) > 1 else "10", "--no-cpu-baseline", "--repeats", "0", "--steps", sys.argv[2] if len(sys.argv) > 2 else "2",
            "--warmup", "0", "--python-loop", "--verbose", "1", "--ignore-status"] + sys.argv[3:]
import bench
bench.main()
