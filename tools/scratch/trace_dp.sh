#!/bin/bash
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/gpurun_out/r05_dp
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in 1 0; do
  rm -rf /tmp/tr_dp$c
  RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29688 NM_CLUSTER_LOOPS=$c NM_DIST_FORCE=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_dp$c -- python $ROOT/bench.py --steps 8 --warmup 3 --beam-batches 0 --no-feed-legs --no-configs --no-cpu-baseline > /tmp/tr_dp$c.log 2>&1; tail -5 /tmp/tr_dp$c.log
  t=$(ls /tmp/tr_dp$c/*/*_kernel_trace.csv | head -1)
  python $ROOT/tools/trace_timeline.py $t 5 25 > $OUT/timeline_dp_clu$c.txt 2>&1
  grep -o '"ms_per_step": [0-9.]*' /tmp/tr_dp$c.log | head -1
done
