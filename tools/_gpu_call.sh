export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_sampling_gpu.py -x -q -m gpu 2>&1 | tail -25
