#!/bin/bash
# Issue / wait accounting of the sweeps (run ON THE GPU BOX through gpurun):  bash tools/debug/pmc_sq.sh <tag> [bench flags]
# Two --pmc passes of SQ counters (8 slots each, no trace domains beside them), summarised per kernel by tools/debug/pmc_sq_summary.py
tag=${1:-sq}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_$tag
mkdir -p $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $out/p1 -- python3 bench.py --steps 2 --warmup 1 --repeats 0 --no-cpu-baseline "$@" > $out/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $out/p2 -- python3 bench.py --steps 2 --warmup 1 --repeats 0 --no-cpu-baseline "$@" > $out/p2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_BUSY_sum --output-format csv -d $out/p3 -- python3 bench.py --steps 2 --warmup 1 --repeats 0 --no-cpu-baseline "$@" > $out/p3.log 2>&1
python3 tools/debug/pmc_sq_summary.py $out | tee $out/summary.txt
