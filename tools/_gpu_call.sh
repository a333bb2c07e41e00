set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
T=r03fin2
R=$PWD
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_train -- python $R/tools/train_profile.py --steps 10 > $R/gpurun_out/${T}_train.txt 2>&1
cp $(find /tmp/p_train -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${T}_train_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_greedy -- python $R/tools/decode_profile.py --mode greedy --batches 8 > $R/gpurun_out/${T}_greedy.txt 2>&1
cp $(find /tmp/p_greedy -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${T}_greedy_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_beam -- python $R/tools/decode_profile.py --mode beam --batches 8 > $R/gpurun_out/${T}_beam.txt 2>&1
cp $(find /tmp/p_beam -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${T}_beam_kernel_stats.csv
tail -1 $R/gpurun_out/${T}_train.txt $R/gpurun_out/${T}_greedy.txt $R/gpurun_out/${T}_beam.txt
