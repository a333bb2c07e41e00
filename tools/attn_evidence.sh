#!/bin/bash
# Regenerates the rocprofv3 evidence for the attention-decoder step kernel (what bench.py's `roofline` cites), for
# the round given as $1 (e.g. r04), on the GPU box:
#     gpurun -- 'bash tools/attn_evidence.sh r04'
# Per shape: kernel-trace passes cold (1 GiB read sweep between launches) / warm (back to back) / dirty (1 GiB write
# sweep) and -- counters in their own passes, as MI355X_MICROARCH.md prescribes -- `--pmc FETCH_SIZE` and
# `--pmc WRITE_SIZE` (cold), summarised by tools/pmc_summary.py into profiles/<round>_attn_*.json.
#   headline   B=128 S=50 A=C=1024 (translation.ini shape):   53 528 576 algorithmic bytes per launch
#   captioning B=128 S=64 A=512 C=2048 (8x8x2048 maps):        4*(B*S*A + B*S*C + 2*B*S + B*A + B*C) = 85 262 336
set -u
R=${1:-r04}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
run() {   # tag, env assignments, algorithmic bytes
    local tag=$1 envs=$2 bytes=$3
    for mode in cold warm dirty; do
        rm -rf /tmp/ev_${tag}_$mode
        env $envs rocprofv3 --kernel-trace --output-format csv -d /tmp/ev_${tag}_$mode -- python $ROOT/tools/attn_only.py 1 23 $mode > /dev/null 2>&1
        python $ROOT/tools/pmc_summary.py --out $ROOT/profiles/${R}_attn_${tag}_trace_$mode.json --match attn_ \
            --trace /tmp/ev_${tag}_$mode --algorithmic-bytes $bytes \
            --note "rocprofv3 --kernel-trace -- $envs python tools/attn_only.py 1 23 $mode" > /dev/null
    done
    env $envs rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/ev_${tag}_fetch -- python $ROOT/tools/attn_only.py 1 23 cold > /dev/null 2>&1
    env $envs rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/ev_${tag}_write -- python $ROOT/tools/attn_only.py 1 23 cold > /dev/null 2>&1
    python $ROOT/tools/pmc_summary.py --out $ROOT/profiles/${R}_attn_${tag}_pmc_cold.json --match attn_ \
        --fetch /tmp/ev_${tag}_fetch --write /tmp/ev_${tag}_write --trace /tmp/ev_${tag}_fetch --algorithmic-bytes $bytes \
        --note "separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (cold) of: $envs python tools/attn_only.py 1 23 cold" > /dev/null
}
run step "NM_B=128 NM_S=50 NM_A=1024 NM_C=1024" 53528576
run cap "NM_B=128 NM_S=64 NM_A=512 NM_C=2048" 85262336
mkdir -p $ROOT/gpurun_out/profiles_$R
cp $ROOT/profiles/${R}_attn_*.json $ROOT/gpurun_out/profiles_$R/
for f in $ROOT/profiles/${R}_attn_*_trace_*.json $ROOT/profiles/${R}_attn_*_pmc_cold.json; do
    python - "$f" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1].split("/")[-1], {k: round(v.get("avg_us", 0), 2) for k, v in d["kernels"].items()},
      "hbm", d.get("hbm_bytes_per_launch"), "frac", d.get("frac_of_8TBps"))
PY
done
