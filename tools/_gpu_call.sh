set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
T=r2u
python -m pytest tests/test_kernels_gpu.py tests/test_dp_gpu.py tests/test_engine_gpu.py tests/test_beam_fused_gpu.py tests/test_ensemble_gpu.py -q --timeout=900 > gpurun_out/${T}_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/${T}_tests.log
tail -40 gpurun_out/${T}_tests.log
