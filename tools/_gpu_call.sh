set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
T=r03t
R=$PWD
timeout 900 python -m pytest tests/test_step_group_gpu.py tests/test_beam_fused_gpu.py tests/test_engine_gpu.py tests/test_fullsize_gpu.py tests/test_ensemble_gpu.py tests/test_captioning_gpu.py -x -q -m gpu > gpurun_out/${T}_tests.txt 2>&1
echo "tests rc=$?"; tail -8 gpurun_out/${T}_tests.txt | cut -c1-400
python tools/decode_profile.py --mode beam --batches 8 2>&1 | tail -1
NM_STEP_TABLES=0 python tools/decode_profile.py --mode beam --batches 8 2>&1 | tail -1
python tools/decode_profile.py --mode greedy --batches 8 2>&1 | tail -1
NM_STEP_TABLES=0 python tools/decode_profile.py --mode greedy --batches 8 2>&1 | tail -1
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_bprof -- python $R/tools/decode_profile.py --mode beam --batches 4 > /dev/null 2>&1
cd $R
find gpurun_out/${T}_bprof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${T}_decode_beam_kernels.csv
rm -rf gpurun_out/${T}_bprof
head -12 gpurun_out/${T}_decode_beam_kernels.csv | cut -c1-150
