export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels_gpu.py -q --timeout=120 -k "chained or colsum" 2>&1 | grep -v "amdgpu.ids" | grep -E "^FAILED|^E   |passed|failed" | cut -c1-300 | head -10
timeout 900 python -m pytest tests/test_general_gpu.py tests/test_nematus_cluster_gpu.py tests/test_coverage_gpu.py tests/test_multisource_gpu.py tests/test_captioning_gpu.py tests/test_dotprod_gpu.py -q -x --timeout=200 2>&1 | grep -v "amdgpu.ids" | grep -E "^FAILED|^E   |passed|failed" | cut -c1-300 | head -10
timeout 600 python tools/general_path_probe.py --train-only NM_WGRAD_CHAINS 1 0 2>&1 | grep "NM_WGRAD" | tail -3
