set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
T=r03al
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "xent" > gpurun_out/${T}_tests.txt 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/${T}_tests.txt | cut -c1-400
python tools/train_profile.py --steps 20 > gpurun_out/${T}_train.txt 2>&1
NM_XENT_COLSUM=0 python tools/train_profile.py --steps 20 >> gpurun_out/${T}_train.txt 2>&1
grep "train:" gpurun_out/${T}_train.txt
