set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
T=r03v
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" > gpurun_out/${T}_tests.txt 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/${T}_tests.txt | cut -c1-400
