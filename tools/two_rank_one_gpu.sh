cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
# two ranks on ONE GPU (both LOCAL_RANK -> device 0) just to exercise the RCCL code path
cat > /tmp/b2.py <<'PY'
import os, sys
os.environ["LOCAL_RANK"]="0"
sys.argv=["bench.py","--gpus","2","--steps","2","--warmup","1","--M","256","--no-cpu-baseline"]
exec(open("bench.py").read())
PY
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 /tmp/b2.py 2>&1 | grep -v "^s*$" | head -60
