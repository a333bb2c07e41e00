export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_cluster_pad_gpu.py -q --timeout=200 2>&1 | grep -v "amdgpu.ids" | grep -E "^FAILED|AssertionError|passed|failed" | cut -c1-300 | head -20
timeout 600 python -m pytest tests/test_kernels_gpu.py -q --timeout=200 -x 2>&1 | grep -v "amdgpu.ids" | grep -E "^FAILED|AssertionError|passed|failed" | cut -c1-300 | head -5
