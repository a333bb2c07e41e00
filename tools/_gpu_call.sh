export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_transformer_fullsize_gpu.py tests/test_fullsize_parity_gpu.py tests/test_captioning_fullsize_gpu.py -q --timeout=600 2>&1 | grep -v "amdgpu.ids" | grep -E "^FAILED|^E   |passed|failed" | cut -c1-330 | head -20
