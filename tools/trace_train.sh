#!/bin/bash
# rocprofv3 kernel trace + per-queue timeline (tools/trace_timeline.py) of one training step:
#     bash tools/trace_train.sh <tag> [ENV=VALUE ...]     -> gpurun_out/r05_clu/timeline_<tag>.txt
tag=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r05_clu
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr_$tag
env "$@" rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$tag -- python $ROOT/tools/train_profile.py --steps 8 $TRAIN_ARGS > /tmp/tr_$tag.log 2>&1
t=$(ls /tmp/tr_$tag/*/*_kernel_trace.csv | head -1)
python $ROOT/tools/trace_timeline.py $t 5 25 > $OUT/timeline_$tag.txt 2>&1
grep "ms/step" /tmp/tr_$tag.log
