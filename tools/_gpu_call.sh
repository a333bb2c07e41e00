set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
T=r03a
R=$PWD
timeout 900 python -m pytest tests/test_transformer_gpu.py tests/test_engine_gpu.py -x -q -m gpu > gpurun_out/${T}_tests.txt 2>&1
echo "tests rc=$?"; tail -12 gpurun_out/${T}_tests.txt | cut -c1-400
timeout 1500 python -m pytest tests/test_transformer_fullsize_gpu.py -x -q -s -m gpu > gpurun_out/${T}_fullsize.txt 2>&1
echo "fullsize rc=$?"; tail -25 gpurun_out/${T}_fullsize.txt | cut -c1-300
timeout 300 python tools/transformer_bench.py > gpurun_out/${T}_tbench.txt 2>&1
cat gpurun_out/${T}_tbench.txt | tail -5
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_tprof -- python $R/tools/transformer_bench.py > /dev/null 2>&1
cd $R
find gpurun_out/${T}_tprof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${T}_transformer_kernel_stats.csv
find gpurun_out/${T}_tprof -name "*_kernel_trace.csv" -delete
find gpurun_out/${T}_tprof -name "*.db" -delete
head -25 gpurun_out/${T}_transformer_kernel_stats.csv | cut -c1-200
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
echo "bench rc=$?"; tail -c 3000 gpurun_out/${T}_bench.json; tail -5 gpurun_out/${T}_bench.err | cut -c1-300
