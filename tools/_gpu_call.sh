set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
T=r03ac
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/${T}_tests.txt 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/${T}_tests.txt | cut -c1-400
