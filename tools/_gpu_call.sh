export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_transformer_gpu.py tests/test_general_gpu.py -x -q -m "gpu and not slow" 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -3
for cfg in "1 1" "0 1"; do set -- $cfg; echo "NM_LN_BWD_FUSED=$1 NM_WGRAD_GROUPS=$2"; NM_LN_BWD_FUSED=$1 NM_WGRAD_GROUPS=$2 timeout 300 python tools/transformer_bench.py --train-only 2>&1 | grep "ms/step"; done
