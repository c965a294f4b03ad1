# usage: bash tools/debug/prof_any.sh <tag> <bench flags...>   -> top kernels of one profiled bench run
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$out -- python3 /root/repo/bench.py --steps 10 --warmup 2 --repeats 0 --no-cpu-baseline "$@" > /root/repo/$out/out.log 2>&1 || exit 1
cd /root/repo
python - <<PY
import csv,glob
f=glob.glob("$out/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per step (12 steps): %.3f"%(tot/12e6))
for r in rows[:18]:
    print("%-95s %5s calls %9.1f us avg %6.1f us/step %5s %%"%(r["Name"][:95], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/12e3, r["Percentage"][:5]))
PY
