set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
T=r2c
python -m pytest tests/test_step_group_gpu.py tests/test_logits_stats_gpu.py tests/test_engine_gpu.py tests/test_fullsize_gpu.py tests/test_fullsize_parity_gpu.py tests/test_reference_inis_gpu.py "tests/test_transformer_gpu.py::test_transformer_base_width_matches_the_oracle" tests/test_beam_fused_gpu.py tests/test_ensemble_gpu.py -q -m gpu --timeout=900 -x > gpurun_out/${T}_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/${T}_tests.log
python tools/decode_profile.py --mode greedy > gpurun_out/${T}_greedy.log 2>&1
python tools/decode_profile.py --mode beam > gpurun_out/${T}_beam.log 2>&1
NM_NO_FUSED_STEP=1 python tools/decode_profile.py --mode greedy > gpurun_out/${T}_greedy_nofused.log 2>&1
for mode in greedy beam; do
  rm -rf gpurun_out/${T}_trace_$mode
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${T}_trace_$mode -- python tools/decode_profile.py --mode $mode --batches 2 > /dev/null 2>&1
  python tools/trace_window.py $(find gpurun_out/${T}_trace_$mode -name "*kernel_trace.csv") 0.3 30 > gpurun_out/${T}_window_$mode.log 2>&1
  rm -rf gpurun_out/${T}_trace_$mode
done
tail -5 gpurun_out/${T}_tests.log; cat gpurun_out/${T}_greedy.log gpurun_out/${T}_beam.log gpurun_out/${T}_greedy_nofused.log | grep -v amdgpu
