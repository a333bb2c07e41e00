export TMPDIR=/tmp
timeout 40 python -m pytest tests/test_reference_inis_gpu.py -x -q -m gpu -k "factored_ini_on" 2>&1 | grep -E "passed|failed|Error|assert|^E " | tail -8
