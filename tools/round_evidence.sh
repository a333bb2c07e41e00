#!/bin/bash
# Kernel-level evidence of a round in one go (on the GPU box):   gpurun -- 'bash tools/round_evidence.sh r04'
#   profiles/<round>_train_kernel_stats.csv      rocprofv3 --kernel-trace --stats of tools/train_profile.py --steps 10
#   profiles/<round>_train_step_timeline.txt     one training step per hardware queue (tools/trace_timeline.py)
#   profiles/<round>_decode_{greedy,beam}_kernels.csv   the same for tools/decode_profile.py --batches 8
#   profiles/<round>_transformer_train_kernel_stats.csv  tools/transformer_bench.py --train-only
set -u
R=${1:-r04}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/profiles_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
stats() {   # tag, command...
    local tag=$1; shift
    rm -rf /tmp/rs_$tag
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rs_$tag -- "$@" > /tmp/rs_$tag.log 2>&1
    local f=$(ls /tmp/rs_$tag/*/*_kernel_stats.csv 2>/dev/null | head -1)
    [ -n "$f" ] && head -40 "$f" > $OUT/${R}_${tag}_kernel_stats.csv
    tail -3 /tmp/rs_$tag.log | grep -v amdgpu.ids > $OUT/${R}_${tag}_wall.txt
}
stats train python $ROOT/tools/train_profile.py --steps 10
t=$(ls /tmp/rs_train/*/*_kernel_trace.csv | head -1)
python $ROOT/tools/trace_timeline.py $t 6 30 > $OUT/${R}_train_step_timeline.txt 2>&1
stats decode_greedy python $ROOT/tools/decode_profile.py --mode greedy --batches 8
stats decode_beam python $ROOT/tools/decode_profile.py --mode beam --batches 8
stats transformer_train python $ROOT/tools/transformer_bench.py --train-only
ls -la $OUT
