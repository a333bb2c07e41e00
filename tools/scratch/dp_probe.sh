#!/bin/bash
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
cd $ROOT
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 NM_DIST_FORCE=1
for e in "NM_DP_EARLY_ISSUE=now" "NM_DP_EARLY_ISSUE=deferred" "NM_DP_EARLY_ISSUE=deferred NM_DIST_ALLREDUCE=nmhip" "NM_DP_OVERLAP=0"; do
  PORT=$((29700 + RANDOM % 200))
  echo "== $e"
  env $e MASTER_PORT=$PORT timeout 300 python bench.py --steps 20 --warmup 3 --beam-batches 0 --no-feed-legs --no-configs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['ms_per_step'], {k: v for k, v in d['dp'].items() if k != 'how'})"
done
