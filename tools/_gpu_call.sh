set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
T=r2s
python -m pytest tests -q -m gpu --timeout=900 > gpurun_out/${T}_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/${T}_tests.log
tail -4 gpurun_out/${T}_tests.log
python bench.py --steps 20 --warmup 3 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/${T}_prof_bench -- python /root/repo/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /root/repo/gpurun_out/${T}_bench_prof.json 2> /dev/null)
cp $(ls gpurun_out/${T}_prof_bench/*/*kernel_stats.csv | head -1) gpurun_out/${T}_bench_kernel_stats.csv
rm -rf gpurun_out/${T}_prof_bench
for mode in cold warm dirty; do
  rm -rf gpurun_out/${T}_attn
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${T}_attn -- python tools/attn_only.py 1 20 $mode > /dev/null 2>&1
  python tools/pmc_summary.py --out gpurun_out/${T}_attn_trace_$mode.json --match attn_whole attn_partial attn_combine --trace gpurun_out/${T}_attn > /dev/null 2>&1
done
rm -rf gpurun_out/${T}_attn
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${T}_pmc_fetch -- python tools/attn_only.py 1 20 cold > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${T}_pmc_write -- python tools/attn_only.py 1 20 cold > /dev/null 2>&1
python tools/pmc_summary.py --out gpurun_out/${T}_attn_step_pmc.json --match attn_whole attn_partial attn_combine --fetch gpurun_out/${T}_pmc_fetch --write gpurun_out/${T}_pmc_write --trace gpurun_out/${T}_pmc_fetch > gpurun_out/${T}_pmc_summary.log 2>&1
rm -rf gpurun_out/${T}_pmc_fetch gpurun_out/${T}_pmc_write
for mode in greedy beam; do
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/${T}_prof_$mode -- python /root/repo/tools/decode_profile.py --mode $mode --batches 4 > /root/repo/gpurun_out/${T}_decode_${mode}_prof.log 2>&1)
  cp $(ls gpurun_out/${T}_prof_$mode/*/*kernel_stats.csv | head -1) gpurun_out/${T}_decode_${mode}_kernel_stats.csv
  rm -rf gpurun_out/${T}_prof_$mode
  python tools/decode_profile.py --mode $mode --batches 8 2>&1 | grep -v amdgpu | tail -1 > gpurun_out/${T}_decode_$mode.log
done
cat gpurun_out/${T}_decode_greedy.log gpurun_out/${T}_decode_beam.log
python - <<'P'
import json
l=json.load(open('gpurun_out/r2s_bench.json'))
r=l['roofline']
print(l['value'], l['ms_per_step'], l['greedy_ms_per_batch'], l['beam5_ms_per_batch'], l.get('ms_per_step_fresh'), l.get('ms_per_step_strings'))
print({k:r.get(k) for k in ('frac','cold_launch_us','stream_read_cold_us','frac_of_stream_read','warm_launch_us','cold_dirty_launch_us','traffic')})
print(l.get('cpu_baseline'))
for m in ('cold','warm','dirty'):
    print(m, open('gpurun_out/r2s_attn_trace_%s.json'%m).read()[:400])
print(open('gpurun_out/r2s_attn_step_pmc.json').read()[:900])
P
head -12 gpurun_out/${T}_decode_greedy_kernel_stats.csv | cut -c1-150
