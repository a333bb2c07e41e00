export TMPDIR=/tmp
timeout 1500 python bench.py > gpurun_out/r06_bench_final.json 2> gpurun_out/r06_bench_final.err; echo "bench rc=$?"
timeout 900 bash tools/attn_evidence.sh r06 > /tmp/attn_ev.log 2>&1; echo "attn rc=$?"; ls gpurun_out/profiles_r06 | head -30
timeout 900 bash tools/round_evidence.sh r06 > /tmp/round_ev.log 2>&1; echo "round rc=$?"
