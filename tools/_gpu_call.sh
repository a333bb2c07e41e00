export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu --timeout=600 --durations=6 > /tmp/suite.log 2>&1; echo "rc=$?"
grep -v "amdgpu.ids" /tmp/suite.log | grep -E "passed|failed|^FAILED|^ERROR|s call" | tail -12 | cut -c1-200 > gpurun_out/r06_gpu_suite_full.txt
cat gpurun_out/r06_gpu_suite_full.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 1500 python bench.py > gpurun_out/r06_bench_final.json 2> gpurun_out/r06_bench_final.err; echo "bench rc=$?"
