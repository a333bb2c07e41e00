export TMPDIR=/tmp
timeout 60 python -m pytest tests/test_reference_inis_gpu.py -x -q -m gpu -k "equals_the_reference_built" 2>&1 | grep -E "passed|failed|Error|assert|^E " | tail -10
