export TMPDIR=/tmp
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 NM_DIST_FORCE=1
for mode in 1 0 1 0; do
echo "== NM_DP_SHARDED=$mode"
MASTER_PORT=2955$mode NM_DP_SHARDED=$mode timeout 600 python bench.py --gpus 1 --steps 20 --warmup 3 --no-configs --no-feed-legs --no-cpu-baseline --beam-batches 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], {k:d['dp'][k] for k in ('optimizer','optimizer_ms','allreduce_exposed_ms')})"
done
