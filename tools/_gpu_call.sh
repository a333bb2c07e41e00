set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_logits_stats_gpu.py tests/test_transformer_gpu.py::test_transformer_base_width_matches_the_oracle tests/test_reference_inis_gpu.py tests/test_engine_gpu.py tests/test_fullsize_gpu.py tests/test_fullsize_parity_gpu.py tests/test_kernels_gpu.py -q -m gpu --timeout=900 > gpurun_out/r2b_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r2b_tests.log
python tools/decode_profile.py --mode greedy > gpurun_out/r2b_greedy.log 2>&1
python tools/decode_profile.py --mode beam > gpurun_out/r2b_beam.log 2>&1
for mode in greedy beam; do
  rm -rf gpurun_out/r2b_trace_$mode
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r2b_trace_$mode -- python tools/decode_profile.py --mode $mode --batches 2 > /dev/null 2>&1
  python tools/trace_window.py $(find gpurun_out/r2b_trace_$mode -name "*kernel_trace.csv") 0.3 30 > gpurun_out/r2b_window_$mode.log 2>&1
  rm -rf gpurun_out/r2b_trace_$mode
done
for rows in 12 8 6; do
  for mode in warm cold dirty; do
    rm -rf gpurun_out/r2b_attn
    NM_ATTN_MAXROWS=$rows timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r2b_attn -- python tools/attn_only.py 1 23 $mode > /dev/null 2>&1
    python tools/pmc_summary.py --out gpurun_out/r2b_attn_${rows}_$mode.json --match attn_partial attn_combine --trace gpurun_out/r2b_attn > /dev/null 2>&1
  done
done
rm -rf gpurun_out/r2b_attn
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/r2b_bench.json 2> gpurun_out/r2b_bench.err
tail -5 gpurun_out/r2b_tests.log; cat gpurun_out/r2b_greedy.log gpurun_out/r2b_beam.log | tail -4
