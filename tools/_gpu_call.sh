mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -q -m gpu > gpurun_out/r04_tests_full_v4.log 2>&1
grep -E "passed|failed|error" gpurun_out/r04_tests_full_v4.log | tail -3
