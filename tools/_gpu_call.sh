set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
T=r3k
python -m pytest tests -q -m gpu --timeout=900 > gpurun_out/${T}_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/${T}_tests.log
tail -4 gpurun_out/${T}_tests.log
python bench.py --steps 20 --warmup 3 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/${T}_prof_bench -- python /root/repo/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /root/repo/gpurun_out/${T}_bench_prof.json 2> /dev/null)
cp $(ls gpurun_out/${T}_prof_bench/*/*kernel_stats.csv | head -1) gpurun_out/${T}_bench_kernel_stats.csv
rm -rf gpurun_out/${T}_prof_bench
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/${T}_prof_beam -- python /root/repo/tools/decode_profile.py --mode beam --batches 4 > /dev/null 2>&1)
cp $(ls gpurun_out/${T}_prof_beam/*/*kernel_stats.csv | head -1) gpurun_out/${T}_decode_beam_kernel_stats.csv
rm -rf gpurun_out/${T}_prof_beam
for mode in greedy beam; do python tools/decode_profile.py --mode $mode --batches 8 2>&1 | grep -v amdgpu | tail -1; done > gpurun_out/${T}_decode.log
cat gpurun_out/${T}_decode.log
python - <<'P'
import json
l=json.load(open('gpurun_out/r3k_bench.json'))
r=l['roofline']
print(l['value'], l['ms_per_step'], l['greedy_ms_per_batch'], l['beam5_ms_per_batch'], l['ms_per_step_fresh'], l['ms_per_step_strings'], l['beam5_decode_tok_s'], l['greedy_decode_tok_s'])
print({k:r.get(k) for k in ('achieved','frac','cold_rotating_launch_us','stream_read_rotating_us','frac_cold_single','cold_launch_us','event_pair_overhead_us','warm_launch_us','traffic','frac_rocprof_cold')})
print(l['cpu_baseline'])
P
