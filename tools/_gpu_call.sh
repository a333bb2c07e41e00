set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
T=r03i
R=$PWD
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/${T}_tests.txt 2>&1
echo "tests rc=$?"; tail -6 gpurun_out/${T}_tests.txt | cut -c1-400
python tools/train_profile.py --steps 20 2>&1 | tail -1
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_tprof -- python $R/tools/train_profile.py --steps 10 > /dev/null 2>&1
cd $R
find gpurun_out/${T}_tprof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${T}_train_kernel_stats.csv
rm -rf gpurun_out/${T}_tprof
grep "attn" gpurun_out/${T}_train_kernel_stats.csv | cut -c1-160
timeout 900 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
echo "bench rc=$?"; tail -c 1500 gpurun_out/${T}_bench.json; tail -3 gpurun_out/${T}_bench.err | cut -c1-300
