#!/bin/bash
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $ROOT/gpurun_out/r05_ev
cd $ROOT && timeout 300 python -m pytest tests/test_engine_gpu.py -q -m gpu -k "device_error" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/mfma
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/mfma -- python $ROOT/tools/train_profile.py --steps 6 > /tmp/mfma.log 2>&1
tail -2 /tmp/mfma.log
python $ROOT/tools/mfma_pmc.py /tmp/mfma $ROOT/gpurun_out/r05_ev/r05_train_mfma_pmc.json | tee $ROOT/gpurun_out/r05_ev/r05_train_mfma_pmc.txt
cd $ROOT && bash tools/scale.sh 8 10 2>&1 | tail -8
cp -r gpurun_out/scale gpurun_out/r05_ev/scale 2>/dev/null
