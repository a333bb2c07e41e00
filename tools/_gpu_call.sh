set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
T=r03ao
timeout 300 python -m pytest tests/test_training_gpu.py tests/test_engine_gpu.py tests/test_step_graphs_gpu.py tests/test_input_pipeline.py tests/test_kernels_gpu.py -x -q -m gpu > gpurun_out/${T}_tests.txt 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/${T}_tests.txt | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
