set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
T=r03ap
timeout 300 python -m pytest tests/test_runners_gpu.py tests/test_engine_gpu.py tests/test_beam_fused_gpu.py tests/test_ensemble_gpu.py -x -q -m gpu > gpurun_out/${T}_tests.txt 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/${T}_tests.txt | cut -c1-300
python tools/decode_profile.py --mode beam --batches 8 2>&1 | tail -1
python tools/decode_profile.py --mode greedy --batches 8 2>&1 | tail -1
