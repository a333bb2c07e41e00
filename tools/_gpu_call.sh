export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_step_cluster_gpu.py tests/test_step_group_gpu.py tests/test_dp_gpu.py -x -q -m gpu 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -12
cd /tmp
for mode in greedy beam; do
  rm -rf /tmp/rs_$mode
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rs_$mode -- python $GRAFT_REPO_ROOT/tools/decode_profile.py --mode $mode --batches 8 > /tmp/rs_$mode.log 2>&1
  f=$(ls /tmp/rs_$mode/*/*_kernel_stats.csv | head -1)
  head -24 $f > $GRAFT_REPO_ROOT/gpurun_out/r06_decode_${mode}_kernel_stats_v1.csv
  grep "ms/batch" /tmp/rs_$mode.log
done
