export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_dp_gpu.py tests/test_gru_cluster_gpu.py tests/test_step_cluster_gpu.py tests/test_step_group_gpu.py -x -q -m "gpu" 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -15
