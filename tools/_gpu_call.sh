export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 150 python -m pytest tests/test_kernels_gpu.py tests/test_sampling_gpu.py tests/test_background_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2
