#!/bin/bash
# Every fuzzer once, with the seeds given (default: fresh ones, not those of tests/test_fuzz_gpu.py); one summary line each.
# usage (on the GPU box):  bash tools/fuzz/run_all.sh [SEED_BASE] > gpurun_out/fuzz_summary.txt
b=${1:-500}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
run() { echo "== $*"; timeout -k 10 600 python "$@" 2>&1 | grep -v "^pmpc_hip: note\|amdgpu.ids" | tail -2 | cut -c1-240; }
run tools/fuzz/fuzz_parity.py $((b+1)) 400
run tools/fuzz/fuzz_soc.py $((b+2)) 150
run tools/fuzz/fuzz_warm_as.py $((b+3)) 80 5
run tools/fuzz/fuzz_xbox.py 80 $((b+4))
run tools/fuzz/fuzz_xbox.py 80 $((b+5)) cone
run tools/fuzz/fuzz_state_rows.py $((b+6)) 150
run tools/fuzz/fuzz_state_rows_cone.py $((b+7)) 40
run tools/fuzz/fuzz_cone.py $((b+8)) 100 24 6
run tools/fuzz/fuzz_cone.py $((b+9)) 60 300 5
run tools/fuzz/fuzz_sharded.py $((b+10)) 400
run tools/fuzz/fuzz_sharded.py $((b+14)) 200 smooth
run tools/fuzz/fuzz_sequence.py $((b+11)) 400
run tools/fuzz/fuzz_scp_loop.py $((b+12)) 300
run tools/fuzz/fuzz_freeze.py $((b+13)) 100 5
run tools/fuzz/fuzz_dense_cons.py $((b+15)) 150
