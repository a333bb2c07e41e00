export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_transformer_gpu.py tests/test_transformer_multisource_gpu.py tests/test_general_gpu.py tests/test_captioning_gpu.py tests/test_engine_gpu.py tests/test_transformer_decode_graphs_gpu.py -q -x --timeout=200 2>&1 | grep -v "amdgpu.ids" | grep -E "^FAILED|^E   |passed|failed" | cut -c1-300 | head -10
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tt_prof -- python $GRAFT_REPO_ROOT/tools/transformer_bench.py --train-only > /tmp/tt.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls /tmp/tt_prof/*/*_kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then head -60 "$f" > gpurun_out/r06_transformer_train_kernel_stats_v6.csv; else tail -5 /tmp/tt.log; fi
