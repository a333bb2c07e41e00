export TMPDIR=/tmp
timeout 100 python -m pytest tests/test_engine_gpu.py tests/test_general_decode_graphs_gpu.py tests/test_runners_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3
