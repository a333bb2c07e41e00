export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_dp_gpu.py -x -q -m gpu 2>&1 | tail -15
