export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_reference_exec_gpu.py -q -m gpu -k "finite" 2>&1 | grep -E "passed|failed|Error|assert|vs finite|diff" | tail -14
