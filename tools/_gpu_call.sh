export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_size_sweep_gpu.py -q --timeout=300 -k general 2>&1 | grep -v "amdgpu.ids" | grep -E "^FAILED|^E   |passed|failed" | cut -c1-400 | head -60
