#!/bin/bash
# Run ON THE GPU BOX through gpurun:  gpurun -- 'bash tools/profile_gpu.sh r01_c'
# 1) kernel-trace + stats, 2) PMC FETCH_SIZE, 3) PMC WRITE_SIZE (separate passes, as the MI355X guide prescribes).
tag=${1:-run}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 5 --warmup 2 --repeats 0 --no-cpu-baseline > $out/bench_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --repeats 0 --no-cpu-baseline > $out/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py --steps 2 --warmup 1 --repeats 0 --no-cpu-baseline > $out/bench_write.log 2>&1
python3 bench.py --steps 20 --warmup 3 > $out/bench_full.log 2>&1
grep '^{' $out/bench_full.log | cut -c1-400
find $out -name '*.csv' | head -20
# other BASELINE configs (no CPU baseline): C = quadrotor M=1024, B = unicycle M=256 N=30
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --M 1024 > $out/bench_C.log 2>&1
for m in 512 2048; do python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --M $m > $out/bench_shard_$m.log 2>&1; done
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --M 256 --N 30 --model unicycle > $out/bench_B.log 2>&1
grep -h '^{' $out/bench_C.log $out/bench_B.log | cut -c1-200
# secondary: the reference's default consensus horizon Nc = N (full consensus) on B and on the quadrotor
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --M 256 --N 30 --model unicycle --Nc -1 > $out/bench_B_NcN.log 2>&1
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --M 4096 --Nc -1 > $out/bench_D_NcN.log 2>&1
grep -h '^{' $out/bench_B_NcN.log $out/bench_D_NcN.log | cut -c1-200
# r03: config E's constraints (thrust cone per stage, N = 100) in fp64 and in fp32-storage mode, config D in fp32-storage mode, the
# reference's DEFAULT solver path (c_lcone_solve) on D and B, and the kernel split of the cone rounds
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --soc --N 100 > $out/bench_E_soc.log 2>&1
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --soc --N 100 --fp32 > $out/bench_E_soc_fp32.log 2>&1
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --fp32 > $out/bench_D_fp32.log 2>&1
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --cone > $out/bench_D_cone.log 2>&1
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --cone --M 256 --N 30 --model unicycle > $out/bench_B_cone.log 2>&1
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --cone --smooth-alpha 10 --M 256 --N 30 --model unicycle > $out/bench_B_cone_smooth.log 2>&1
grep -h '^{' $out/bench_E_soc.log $out/bench_E_soc_fp32.log $out/bench_D_fp32.log $out/bench_D_cone.log $out/bench_B_cone.log $out/bench_B_cone_smooth.log | cut -c1-160
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_E_soc -- python3 bench.py --steps 5 --warmup 2 --repeats 0 --no-cpu-baseline --soc --N 100 > $out/bench_stats_E_soc.log 2>&1
# r03: boxed slew problems (increment form + state-box rounds against the generic kernels) and binding state boxes (rounds on / off)
python3 tools/debug/slew_paths.py > $out/slew_paths.txt 2>&1
V=1 python3 tools/debug/xbox_check.py > $out/xbox_on.log 2>&1
PMPC_XBOX_AS=0 V=1 python3 tools/debug/xbox_check.py > $out/xbox_off.log 2>&1
for f in on off; do echo "state-box rounds $f: solves, interior-point iterations, factorisations"; grep "status " $out/xbox_$f.log | awk '{n++; it+=$6; ss+=$9} END {print n, it, ss}'; done > $out/xbox_summary.txt
cat $out/xbox_summary.txt
# r03: config D with a velocity limit that binds (state rows of the active-set rounds; PMPC_XBOX_AS=0 = interior-point iteration, the r02 path)
for v in 3.0 2.0; do python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --repeats 1 --vmax $v > $out/bench_D_vmax$v.log 2>&1; done
PMPC_XBOX_AS=0 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --repeats 1 --vmax 3.0 > $out/bench_D_vmax3.0_ipm.log 2>&1
grep -h '^{' $out/bench_D_vmax3.0.log $out/bench_D_vmax2.0.log $out/bench_D_vmax3.0_ipm.log | cut -c1-160

# r04: PMC traffic of the dominant kernel for the workloads whose bench lines carry their own alg_bytes_per_launch (config E's cone
# sweeps in fp64 and fp32 storage, config D in fp32 storage): separate FETCH_SIZE / WRITE_SIZE passes each, as for config D above
for wl in "E_soc:--soc --N 100" "D_fp32:--fp32" "E_soc_fp32:--soc --N 100 --fp32"; do
  name=${wl%%:*}; flags=${wl#*:}
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_$name -- python3 bench.py --steps 2 --warmup 1 --repeats 0 --no-cpu-baseline $flags > $out/bench_fetch_$name.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_$name -- python3 bench.py --steps 2 --warmup 1 --repeats 0 --no-cpu-baseline $flags > $out/bench_write_$name.log 2>&1
done
# r04: the reference's default path with log-barrier smoothing at config D (full-space Newton, any tie pattern) and squareplus at B;
# stage timeline of the two sweeps (diagnostic build, tools/micro/stage_timeline.py)
python3 bench.py --steps 10 --warmup 2 --repeats 0 --no-cpu-baseline --cone --smooth-alpha 10 > $out/bench_D_cone_smooth.log 2>&1
grep -h '^{' $out/bench_D_cone_smooth.log | cut -c1-160
if [ -f libs_tmp/libpmpc_hip_tl.so ]; then python3 tools/micro/stage_timeline.py 512 4096 > $out/stage_timeline.txt 2>&1; fi
