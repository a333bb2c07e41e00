set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
T=r03b
timeout 1500 python -m pytest tests/test_transformer_fullsize_gpu.py -q -s -m gpu > gpurun_out/${T}_fullsize.txt 2>&1
echo "fullsize rc=$?"; grep -v "^decoder/\|^encoder" gpurun_out/${T}_fullsize.txt | tail -40 | cut -c1-300
