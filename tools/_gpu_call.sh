export TMPDIR=/tmp
timeout 900 python tools/transformer_bench.py 2>&1 | grep -E "train:|greedy|beam|rror" | head -5
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x --timeout=100 -k "layer_norm" 2>&1 | grep -E "^FAILED|^E   |passed|failed" | cut -c1-300 | head -5
