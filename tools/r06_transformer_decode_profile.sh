#!/bin/bash
# Transformer decode (configs[4]) kernel statistics, greedy and beam-5 separately (on the GPU box):
#   gpurun -- 'bash tools/r06_transformer_decode_profile.sh r06'
set -u
R=${1:-r06}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/profiles_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
stats() {   # tag, command...
    local tag=$1; shift
    rm -rf /tmp/rs_$tag
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rs_$tag -- "$@" > /tmp/rs_$tag.log 2>&1
    local f=$(ls /tmp/rs_$tag/*/*_kernel_stats.csv 2>/dev/null | head -1)
    [ -n "$f" ] && head -60 "$f" > $OUT/${R}_${tag}_kernel_stats.csv
    grep -E "ms/batch|ms/step|parameters" /tmp/rs_$tag.log > $OUT/${R}_${tag}_wall.txt
}
stats transformer_greedy python $ROOT/tools/transformer_bench.py --greedy-only
stats transformer_beam python $ROOT/tools/transformer_bench.py --beam-5-only
ls -la $OUT
