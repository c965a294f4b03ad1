#!/bin/bash
# A/B different builds of libpmpc_hip.so on the same box (same device, same process environment): libs_tmp/*.so in turn, through
# PMPC_HIP_LIB (pmpc_amd/_lib.py).  AB_M: particle counts, AB_REPS: passes, AB_ARGS: further bench.py flags.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
for rep in $(seq 1 ${AB_REPS:-2}); do
for f in libs_tmp/*.so; do
  for m in ${AB_M:-512 4096}; do
    tag=$(basename $f .so)_M${m}_r${rep}
    PMPC_HIP_LIB=$f python bench.py --steps 20 --warmup 3 --repeats 1 --no-cpu-baseline --M $m ${AB_ARGS} > gpurun_out/ab/$tag.log 2>&1
    python tools/bench_line.py gpurun_out/ab/$tag.log | cut -c1-420
  done
done
done
