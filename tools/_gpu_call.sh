set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
T=r03final
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/${T}_tests.txt 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/${T}_tests.txt | cut -c1-300
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
echo "bench rc=$?"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${T}_smoke.txt 2>&1
tail -1 gpurun_out/${T}_smoke.txt
