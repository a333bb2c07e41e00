export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_general_gpu.py tests/test_nematus_cluster_gpu.py tests/test_coverage_gpu.py tests/test_multisource_gpu.py tests/test_captioning_gpu.py tests/test_dotprod_gpu.py tests/test_general_decode_graphs_gpu.py tests/test_reference_ini_parity_gpu.py tests/test_reference_exec_gpu.py -q -x --timeout=200 2>&1 | grep -v "amdgpu.ids" | grep -E "^FAILED|^E   |passed|failed" | cut -c1-300 | head -10
timeout 600 python tools/general_path_probe.py NM_NEMATUS_CLUSTER 1 2>&1 | grep "NM_NEM" | tail -3
NM_NEMATUS_CELL_MERGED=0 timeout 600 python tools/general_path_probe.py NM_NEMATUS_CLUSTER 1 2>&1 | grep "NM_NEM" | tail -3
