set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
T=r03am
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_transformer_gpu.py tests/test_transformer_fullsize_gpu.py tests/test_reference_inis_gpu.py -x -q -m gpu -k "layer_norm or transformer or Transformer or greedy or beam or ini" > gpurun_out/${T}_tests.txt 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/${T}_tests.txt | cut -c1-400
python tools/transformer_bench.py --greedy-only --beam-5-only > gpurun_out/${T}_tb.txt 2>&1
tail -2 gpurun_out/${T}_tb.txt
