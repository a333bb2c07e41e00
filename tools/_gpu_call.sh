export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_nematus_cluster_gpu.py tests/test_general_gpu.py tests/test_size_sweep_gpu.py -q -x --timeout=300 -k "not untuned_sizes or general" 2>&1 | grep -v "amdgpu.ids" | grep -E "^FAILED|^E   |passed|failed" | cut -c1-400 | head -20
timeout 600 python tools/general_path_probe.py NM_NEMATUS_CLUSTER 1 2>&1 | grep "NM_NEM" | tail -2
