set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
T=r03aq
python tools/batch_boundary_probe.py greedy 2>&1 | grep "greedy:" > gpurun_out/${T}_probe.txt
python tools/batch_boundary_probe.py beam 2>&1 | grep "beam:" >> gpurun_out/${T}_probe.txt
cat gpurun_out/${T}_probe.txt
