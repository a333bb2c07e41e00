export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_nematus_cluster_gpu.py -q -x --timeout=300 -k stacked 2>&1 | grep -v "amdgpu.ids" | grep -E "^FAILED|^E   |passed|failed" | cut -c1-400 | head -20
