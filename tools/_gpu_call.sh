export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x --timeout=100 -k "layer_norm" 2>&1 | grep -E "^FAILED|^E   |passed|failed" | cut -c1-300 | head -10
timeout 900 python -m pytest tests/test_transformer_gpu.py tests/test_transformer_multisource_gpu.py tests/test_general_gpu.py tests/test_captioning_gpu.py -q -x --timeout=200 2>&1 | grep -v "amdgpu.ids" | grep -E "^FAILED|^E   |passed|failed" | cut -c1-300 | head -10
timeout 300 python tools/transformer_bench.py --train-only 2>&1 | grep -E "train:"
