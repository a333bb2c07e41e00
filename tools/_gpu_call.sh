set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
T=r03k
R=$PWD
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_runners_gpu.py tests/test_general_decode_graphs_gpu.py tests/test_transformer_decode_graphs_gpu.py tests/test_ensemble_gpu.py -x -q -m gpu > gpurun_out/${T}_tests.txt 2>&1
echo "tests rc=$?"; tail -8 gpurun_out/${T}_tests.txt | cut -c1-400
python tools/decode_profile.py --mode greedy --batches 8 2>&1 | tail -1
NM_NO_LOOKAHEAD=1 python tools/decode_profile.py --mode greedy --batches 8 2>&1 | tail -1
python tools/decode_profile.py --mode beam --batches 8 2>&1 | tail -1
NM_NO_LOOKAHEAD=1 python tools/decode_profile.py --mode beam --batches 8 2>&1 | tail -1
