set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
T=r2g
python -m pytest tests/test_kernels_gpu.py tests/test_logits_stats_gpu.py tests/test_step_group_gpu.py tests/test_engine_gpu.py tests/test_training_gpu.py tests/test_beam_fused_gpu.py tests/test_reuse_grad_gpu.py tests/test_fullsize_parity_gpu.py tests/test_fullsize_gpu.py tests/test_dp_gpu.py tests/test_captioning_gpu.py -q -m gpu --timeout=900 > gpurun_out/${T}_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/${T}_tests.log
python tools/decode_profile.py --mode greedy > gpurun_out/${T}_greedy.log 2>&1
python tools/decode_profile.py --mode beam > gpurun_out/${T}_beam.log 2>&1
python tools/stats_gemm_probe.py > gpurun_out/${T}_probe.log 2>&1
for mode in greedy beam; do
  rm -rf gpurun_out/${T}_trace_$mode
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${T}_trace_$mode -- python tools/decode_profile.py --mode $mode --batches 2 > /dev/null 2>&1
  python tools/trace_window.py $(find gpurun_out/${T}_trace_$mode -name "*kernel_trace.csv") 0.3 30 > gpurun_out/${T}_window_$mode.log 2>&1
  rm -rf gpurun_out/${T}_trace_$mode
done
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --batch 16 --beam-batches 0 > gpurun_out/${T}_bench_b16.json 2> gpurun_out/${T}_bench_b16.err
rm -rf gpurun_out/${T}_prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_prof -- python tools/train_profile.py --steps 10 > /dev/null 2>&1
cp $(find gpurun_out/${T}_prof -name "*kernel_stats.csv") gpurun_out/${T}_train_kernel_stats.csv
python tools/trace_step.py $(find gpurun_out/${T}_prof -name "*kernel_trace.csv") 4 40 > gpurun_out/${T}_train_step.log 2>&1
rm -rf gpurun_out/${T}_prof
tail -3 gpurun_out/${T}_tests.log; cat gpurun_out/${T}_greedy.log gpurun_out/${T}_beam.log gpurun_out/${T}_probe.log | grep -v amdgpu; head -12 gpurun_out/${T}_window_greedy.log; head -12 gpurun_out/${T}_window_beam.log; cat gpurun_out/${T}_bench.json gpurun_out/${T}_bench_b16.json | cut -c1-900; head -30 gpurun_out/${T}_train_step.log
