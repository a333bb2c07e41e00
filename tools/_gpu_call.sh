export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels_gpu.py -q --timeout=100 -k "round_robin" 2>&1 | grep -v "amdgpu.ids" | grep -E "^FAILED|^E   |passed|failed" | cut -c1-400 | head -10
