set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
T=r2j
python -m pytest tests -q -m gpu --timeout=1200 > gpurun_out/${T}_pytest_all.log 2>&1
echo "all rc=$?" >> gpurun_out/${T}_pytest_all.log
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
rm -rf gpurun_out/${T}_prof
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_prof -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_prof.log 2>&1
cp $(find gpurun_out/${T}_prof -name "*kernel_stats.csv") gpurun_out/${T}_bench_kernel_stats.csv
rm -rf gpurun_out/${T}_prof
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${T}_prof -- python tools/train_profile.py --steps 8 > /dev/null 2>&1
python tools/trace_timeline.py $(find gpurun_out/${T}_prof -name "*kernel_trace.csv") 4 60 > gpurun_out/${T}_train_timeline.log 2>&1
python tools/trace_step.py $(find gpurun_out/${T}_prof -name "*kernel_trace.csv") 4 30 > gpurun_out/${T}_train_step.log 2>&1
rm -rf gpurun_out/${T}_prof
for mode in warm cold dirty; do
  rm -rf gpurun_out/${T}_attn
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${T}_attn -- python tools/attn_only.py 1 23 $mode > /dev/null 2>&1
  python tools/pmc_summary.py --out gpurun_out/${T}_attn_trace_$mode.json --match attn_partial attn_combine --trace gpurun_out/${T}_attn > /dev/null 2>&1
done
rm -rf gpurun_out/${T}_attn gpurun_out/${T}_pmc_fetch gpurun_out/${T}_pmc_write
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${T}_pmc_fetch -- python tools/attn_only.py 1 23 cold > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${T}_pmc_write -- python tools/attn_only.py 1 23 cold > /dev/null 2>&1
python tools/pmc_summary.py --out gpurun_out/${T}_attn_step_pmc.json --match attn_partial attn_combine --fetch gpurun_out/${T}_pmc_fetch --write gpurun_out/${T}_pmc_write --trace gpurun_out/${T}_pmc_fetch > gpurun_out/${T}_pmc_summary.log 2>&1
rm -rf gpurun_out/${T}_pmc_fetch gpurun_out/${T}_pmc_write
for mode in greedy beam; do
  rm -rf gpurun_out/${T}_trace_$mode
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${T}_trace_$mode -- python tools/decode_profile.py --mode $mode --batches 2 > /dev/null 2>&1
  python tools/trace_window.py $(find gpurun_out/${T}_trace_$mode -name "*kernel_trace.csv") 0.3 30 > gpurun_out/${T}_window_$mode.log 2>&1
  rm -rf gpurun_out/${T}_trace_$mode
done
python tools/decode_profile.py --mode greedy --batches 8 > gpurun_out/${T}_greedy.log 2>&1
python tools/decode_profile.py --mode beam --batches 8 > gpurun_out/${T}_beam.log 2>&1
tail -4 gpurun_out/${T}_pytest_all.log; cut -c1-600 gpurun_out/${T}_bench.json; grep -v amdgpu gpurun_out/${T}_greedy.log gpurun_out/${T}_beam.log
