set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
T=r03an
NM_ATTN_HALF=1 timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attn or attention" > gpurun_out/${T}_tests.txt 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/${T}_tests.txt | cut -c1-300
for h in 0 1; do
  echo "HALF=$h" >> gpurun_out/${T}_attn.txt
  NM_ATTN_HALF=$h python tools/attn_only.py 1 30 cold >> gpurun_out/${T}_attn.txt 2>&1
  NM_ATTN_HALF=$h python tools/attn_only.py 1 30 warm >> gpurun_out/${T}_attn.txt 2>&1
  NM_ATTN_HALF=$h python tools/decode_profile.py --mode greedy --batches 8 2>&1 | tail -1 >> gpurun_out/${T}_attn.txt
done
grep -v "^W\|amdgpu.ids" gpurun_out/${T}_attn.txt | tail -20
