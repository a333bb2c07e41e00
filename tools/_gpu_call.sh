set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
T=r3f
python bench.py --steps 20 --warmup 3 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/${T}_prof_bench -- python /root/repo/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /root/repo/gpurun_out/${T}_bench_prof.json 2> /dev/null)
cp $(ls gpurun_out/${T}_prof_bench/*/*kernel_stats.csv | head -1) gpurun_out/${T}_bench_kernel_stats.csv
rm -rf gpurun_out/${T}_prof_bench
for mode in greedy beam; do
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/${T}_prof_$mode -- python /root/repo/tools/decode_profile.py --mode $mode --batches 4 > /dev/null 2>&1)
  cp $(ls gpurun_out/${T}_prof_$mode/*/*kernel_stats.csv | head -1) gpurun_out/${T}_decode_${mode}_kernel_stats.csv
  rm -rf gpurun_out/${T}_prof_$mode
  python tools/decode_profile.py --mode $mode --batches 8 2>&1 | grep -v amdgpu | tail -1 > gpurun_out/${T}_decode_$mode.log
done
cat gpurun_out/${T}_decode_greedy.log gpurun_out/${T}_decode_beam.log
python - <<'P'
import json
l=json.load(open('gpurun_out/r3f_bench.json'))
r=l['roofline']
print(l['value'], l['ms_per_step'], l['greedy_ms_per_batch'], l['beam5_ms_per_batch'], l['ms_per_step_fresh'], l['ms_per_step_strings'])
print({k:r.get(k) for k in ('achieved','frac','cold_rotating_launch_us','stream_read_rotating_us','frac_of_stream_read_rotating','frac_cold_single','cold_launch_us','event_pair_overhead_us','warm_launch_us','cold_dirty_launch_us','traffic','rocprof_kernel_us','frac_rocprof_cold')})
print(l['cpu_baseline'])
P
grep "attn_whole\|attn_partial_fastq" gpurun_out/${T}_bench_kernel_stats.csv | cut -c1-120
