export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_background_gpu.py -x -q -m gpu 2>&1 | tail -15
