export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x --timeout=100 -k "outer_products" 2>&1 | grep -E "^FAILED|^E   |passed|failed" | cut -c1-300 | head -5
timeout 1200 python -m pytest tests/test_general_gpu.py tests/test_size_sweep_gpu.py tests/test_coverage_gpu.py tests/test_multisource_gpu.py tests/test_captioning_gpu.py tests/test_dotprod_gpu.py tests/test_reference_ini_parity_gpu.py tests/test_reference_exec_gpu.py tests/test_transformer_gpu.py tests/test_transformer_multisource_gpu.py -q -x --timeout=300 -k "not untuned_sizes or general" 2>&1 | grep -v "amdgpu.ids" | grep -E "^FAILED|^E   |passed|failed" | cut -c1-400 | head -20
timeout 600 python tools/general_path_probe.py NM_NEMATUS_CLUSTER 1 2>&1 | grep "NM_NEM" | tail -2
