#!/bin/bash
# The multi-GPU legs of the bench in one go, on a node with N GPUs (the driver's 8-GPU node; 1 GPU works too):
#     bash tools/scale.sh [max_gpus=8] [steps=20]
# For N in 1 2 4 8 (<= max_gpus and <= visible GPUs): `bench.py --gpus N` weak and `--scaling strong`, with the gradient
# all-reduce through torch.distributed (NM_DIST_ALLREDUCE=torch) and through the library's own RCCL communicator
# (nmhip), one rank per GPU over RCCL exactly as the driver launches it; every line must carry dp.ranks_seen == N (an
# RCCL all-gather of the ranks inside the run).  N = 1 runs with NM_DIST_FORCE=1: a process group of one, every
# collective issued (each is the identity) -- the same code path with nobody to talk to.  Then the RCCL tests.
# Lines land in gpurun_out/scale/N<n>_<scaling>_<allreduce>.json; a summary table is printed at the end.
set -u
MAXN=${1:-8}
STEPS=${2:-20}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/scale
mkdir -p $OUT
cd $ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
NGPU=$(python -c 'import torch; print(torch.cuda.device_count())')
PORT=29611
fail=0
for N in 1 2 4 8; do
    [ $N -le $MAXN ] && [ $N -le $NGPU ] || continue
    for scaling in weak strong; do
        for ar in torch nmhip; do
            tag=N${N}_${scaling}_${ar}
            PORT=$((PORT + 1))
            extra=""
            [ $N -eq 1 ] && extra="NM_DIST_FORCE=1"
            env NM_DIST_ALLREDUCE=$ar $extra timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N \
                --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus $N --steps $STEPS --warmup 3 \
                --scaling $scaling --beam-batches 0 --no-feed-legs --no-configs --no-cpu-baseline \
                > $OUT/$tag.json 2> $OUT/$tag.err
            rc=$?
            python - $OUT/$tag.json $N $rc <<'PY' || fail=1
import json, sys
path, n, rc = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
try:
    line = json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])
except Exception as exc:
    print("FAIL {}: rc {} and no JSON line ({})".format(path, rc, exc)); sys.exit(1)
dp = line.get("dp") or {}
ok = rc == 0 and line["n_gpus"] == n and dp.get("ranks_seen") == n
print("{} {:<28} {:>10.0f} tok/s {:>8.3f} ms/step  ranks_seen {}  exposed all-reduce {} ms".format(
    "ok  " if ok else "FAIL", path.split("/")[-1], line["value"], line["ms_per_step"], dp.get("ranks_seen"),
    dp.get("allreduce_exposed_ms")))
sys.exit(0 if ok else 1)
PY
        done
    done
done
timeout 900 python -m pytest tests/test_dp_gpu.py -q -m gpu -k "rccl or communicator" > $OUT/rccl_tests.txt 2>&1 || fail=1
tail -2 $OUT/rccl_tests.txt
[ $fail -eq 0 ] && echo "scale.sh: every line ok" || echo "scale.sh: FAILURES above"
exit $fail
