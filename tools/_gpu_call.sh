set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
T=r2q
python -m pytest tests/test_coverage_gpu.py tests/test_training_gpu.py tests/test_general_gpu.py tests/test_multisource_gpu.py tests/test_captioning_gpu.py tests/test_transformer_gpu.py -q --timeout=900 > gpurun_out/${T}_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/${T}_tests.log
tail -30 gpurun_out/${T}_tests.log
