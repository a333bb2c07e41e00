export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_reference_exec_gpu.py tests/test_stateful_context_gpu.py -q -m gpu -k "stateful" 2>&1 | grep -E "passed|failed|Error|assert" | tail -12
