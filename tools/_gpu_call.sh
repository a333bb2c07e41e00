set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
T=r03s
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/${T}_tests.txt 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/${T}_tests.txt | cut -c1-300
timeout 900 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
echo "bench rc=$?"; tail -c 600 gpurun_out/${T}_bench.json; tail -3 gpurun_out/${T}_bench.err | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
