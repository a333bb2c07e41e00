set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
T=r03e
R=$PWD
timeout 900 python -m pytest tests/test_step_group_gpu.py tests/test_beam_fused_gpu.py -x -q -m gpu > gpurun_out/${T}_tests.txt 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/${T}_tests.txt | cut -c1-400
python tools/decode_profile.py --mode beam --batches 6 2>&1 | tail -1
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_bprof -- python $R/tools/decode_profile.py --mode beam --batches 4 > /dev/null 2>&1
cd $R
find gpurun_out/${T}_bprof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${T}_decode_beam_kernels.csv
rm -rf gpurun_out/${T}_bprof
head -8 gpurun_out/${T}_decode_beam_kernels.csv | cut -c1-180
timeout 1500 python -m pytest tests/test_fullsize_parity_gpu.py -q -s -m gpu > gpurun_out/${T}_fullsize.txt 2>&1
echo "fullsize rc=$?"; grep -v "^decoder/\|^encoder\|^attention/" gpurun_out/${T}_fullsize.txt | tail -30 | cut -c1-300
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_tprof -- python $R/tools/transformer_bench.py --train-only > /dev/null 2>&1
cd $R
find gpurun_out/${T}_tprof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${T}_transformer_train_kernel_stats.csv
rm -rf gpurun_out/${T}_tprof
head -30 gpurun_out/${T}_transformer_train_kernel_stats.csv | cut -c1-200
