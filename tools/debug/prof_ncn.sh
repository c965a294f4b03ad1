out=gpurun_out/r4p/prof2
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$out -- python3 /root/repo/bench.py --Nc -1 --steps 10 --warmup 2 --repeats 0 --no-cpu-baseline $EXTRA > /root/repo/$out/out.log 2>&1 || exit 1
cd /root/repo
python - <<PY
import csv,glob
f=glob.glob("$out/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r["Name"][:100], r["Calls"], r["AverageNs"], r["Percentage"])
PY
