#!/bin/bash
# A/B different builds of libpmpc_hip.so on the same box (same device, same process environment): libs_tmp/*.so in turn
cd $GRAFT_REPO_ROOT
cp pmpc_amd/libpmpc_hip.so /tmp/orig.so
for rep in $(seq 1 ${AB_REPS:-2}); do
for f in libs_tmp/*.so; do
  cp $f pmpc_amd/libpmpc_hip.so
  for m in ${AB_M:-256 4096}; do
    python bench.py --steps 20 --warmup 3 --repeats 1 --no-cpu-baseline --ignore-status --M $m ${AB_ARGS} 2>&1 | grep -E "^\{" | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('$f', 'M=$m', 'it/s', round(d['value'],1), [round(v,1) for v in d['repeats']['values']], 'factor us', round(1e3*d['roofline']['avg_launch_ms'],1), {k: round(v,3) for k,v in d['roofline']['kernel_ms_per_step'].items()}, 'rounds', d['config']['active_set_rounds_per_step'])"
  done
done
done
cp /tmp/orig.so pmpc_amd/libpmpc_hip.so
