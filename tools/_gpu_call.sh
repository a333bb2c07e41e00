mkdir -p gpurun_out
export TMPDIR=/tmp
python tools/captioning_train_probe.py 2>&1 | tail -1
python -m pytest tests -q -m gpu 2>&1 | tail -5 > gpurun_out/r04_tests_full_v2.txt
cat gpurun_out/r04_tests_full_v2.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
