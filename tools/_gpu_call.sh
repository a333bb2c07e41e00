export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_no_foreign_kernels_gpu.py -q --timeout=120 2>&1 | grep -v "amdgpu.ids" | grep -E "^FAILED|^E   |passed|failed" | cut -c1-300 | head -10
timeout 600 python -m pytest tests/test_engine_gpu.py tests/test_beam_gpu.py tests/test_cluster_recovery_gpu.py tests/test_transformer_gpu.py -q -x --timeout=200 2>&1 | grep -v "amdgpu.ids" | grep -E "^FAILED|^E   |passed|failed" | cut -c1-300 | head -10
