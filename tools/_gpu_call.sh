mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import torch; print(torch.cuda.Stream.priority_range())"
echo "== main high prio"; NM_MAIN_PRIO=1 python tools/batch_boundary_probe.py greedy 6 2>&1 | tail -1
echo "== main high prio, ahead not background"; NM_AHEAD_BACKGROUND=0 NM_MAIN_PRIO=1 python tools/batch_boundary_probe.py greedy 6 2>&1 | tail -1
echo "== main high prio beam"; NM_MAIN_PRIO=1 python tools/batch_boundary_probe.py beam 6 2>&1 | tail -1
echo "== default"; python tools/batch_boundary_probe.py greedy 6 2>&1 | tail -1
