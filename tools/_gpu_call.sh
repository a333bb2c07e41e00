set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
T=r03ag
timeout 900 python -m pytest tests/test_transformer_gpu.py tests/test_transformer_fullsize_gpu.py tests/test_dotprod_gpu.py tests/test_reference_inis_gpu.py tests/test_multisource_gpu.py -q -m gpu > gpurun_out/${T}_tests.txt 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/${T}_tests.txt | cut -c1-400
