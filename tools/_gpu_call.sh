export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_size_sweep_gpu.py -q --timeout=400 -k "190 or 191 or 192 or 193 or 590 or 591" 2>&1 | grep -v "amdgpu.ids" | grep -E "^FAILED|^E   |passed|failed" | cut -c1-400 | head -30
