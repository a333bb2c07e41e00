export TMPDIR=/tmp
B="python bench.py --no-configs --no-feed-legs --no-cpu-baseline --beam-batches 0"
for i in 1 2; do
echo "== HEAD (side-stream column sums: low-pressure kernel)"; timeout 300 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
echo "== 46bd334 (before the float4 column sums)"; (cd _bisect/46bd334 && timeout 300 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
done
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_train_gpu.py -q --timeout=120 -x 2>&1 | grep -E "^FAILED|^E   |passed|failed" | cut -c1-300 | head -10
