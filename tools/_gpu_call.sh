set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 100 python -m pytest tests/test_training_gpu.py -x -q -m gpu 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
