#!/bin/bash
# A/B different builds of libpmpc_hip.so on the same box (rule 24: same device, same process environment)
cd $GRAFT_REPO_ROOT
cp pmpc_amd/libpmpc_hip.so /tmp/orig.so
for rep in 1 2 3; do
for f in libs_tmp/*.so; do
  cp $f pmpc_amd/libpmpc_hip.so
  for m in ${AB_M:-256 4096}; do
    python bench.py --steps 20 --warmup 3 --no-cpu-baseline --ignore-status --M $m 2>&1 | grep -E "^\{" | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('$f', 'M=$m', 'it/s', round(d['value'],1), 'factor us', round(1e3*d['roofline']['avg_launch_ms'],1), {k: round(v,2) for k,v in d['roofline']['kernel_ms_per_step'].items()})"
  done
done
done
cp /tmp/orig.so pmpc_amd/libpmpc_hip.so
