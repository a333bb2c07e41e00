set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
T=r03final2
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
echo "bench rc=$?"
