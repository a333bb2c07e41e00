set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
T=r2m
python -m pytest tests/test_beam_fused_gpu.py tests/test_engine_gpu.py tests/test_fullsize_parity_gpu.py tests/test_ensemble_gpu.py tests/test_transformer_gpu.py -q -m gpu --timeout=900 > gpurun_out/${T}_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/${T}_tests.log
python tools/gemm_sweep.py 1 1+NT > gpurun_out/${T}_gemm_sweep.log 2>&1
python tools/decode_profile.py --mode beam --batches 8 > gpurun_out/${T}_beam.log 2>&1
tail -3 gpurun_out/${T}_tests.log; cat gpurun_out/${T}_gemm_sweep.log | head -16; grep -v amdgpu gpurun_out/${T}_beam.log
