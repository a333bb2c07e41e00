set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
T=r3l
python -m pytest tests -q -m gpu --timeout=900 -x > gpurun_out/${T}_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/${T}_tests.log
tail -4 gpurun_out/${T}_tests.log
for mode in greedy beam; do python tools/decode_profile.py --mode $mode --batches 8 2>&1 | grep -v amdgpu | tail -1; done > gpurun_out/${T}_decode.log
cat gpurun_out/${T}_decode.log
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
python -c "
import json; l=json.load(open('gpurun_out/${T}_bench.json')); print(l['value'], l['ms_per_step'], l['greedy_ms_per_batch'], l['beam5_ms_per_batch'], l['roofline']['frac'])"
