set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
T=r2w
NM_ATTN_PAIR=1 timeout 600 python -m pytest tests/test_abi.py tests/test_kernels_gpu.py tests/test_step_group_gpu.py tests/test_engine_gpu.py -q --timeout=300 -k "not dispatch" > gpurun_out/${T}_tests_pair.log 2>&1
echo "tests rc=$?" >> gpurun_out/${T}_tests_pair.log
tail -5 gpurun_out/${T}_tests_pair.log
for m in cold warm dirty; do
  NM_ATTN_PAIR=1 timeout 120 python tools/attn_only.py 1 30 $m 2>&1 | grep whole=
  NM_ATTN_PAIR=0 timeout 120 python tools/attn_only.py 1 30 $m 2>&1 | grep whole=
done > gpurun_out/${T}_attn_pair.log 2>&1
for bb in 16 64 256; do
  NM_B=$bb NM_ATTN_PAIR=1 timeout 120 python tools/attn_only.py 1 30 warm 2>&1 | grep whole=
  NM_B=$bb NM_ATTN_PAIR=0 timeout 120 python tools/attn_only.py 1 30 warm 2>&1 | grep whole=
done >> gpurun_out/${T}_attn_pair.log 2>&1
NM_S=30 NM_ATTN_PAIR=1 timeout 120 python tools/attn_only.py 1 30 warm 2>&1 | grep whole= >> gpurun_out/${T}_attn_pair.log
NM_S=30 NM_ATTN_PAIR=0 timeout 120 python tools/attn_only.py 1 30 warm 2>&1 | grep whole= >> gpurun_out/${T}_attn_pair.log
cat gpurun_out/${T}_attn_pair.log
for w in 0 1; do NM_ATTN_PAIR=$w timeout 300 python tools/decode_profile.py --mode greedy --batches 8 2>&1 | grep -v amdgpu | tail -1; done > gpurun_out/${T}_decode_pair.log 2>&1
cat gpurun_out/${T}_decode_pair.log
