set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
T=r3i
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_beam_fused_gpu.py tests/test_engine_gpu.py tests/test_logits_stats_gpu.py tests/test_ensemble_gpu.py tests/test_general_decode_graphs_gpu.py tests/test_step_graphs_gpu.py tests/test_transformer_decode_graphs_gpu.py -q --timeout=600 > gpurun_out/${T}_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/${T}_tests.log
tail -4 gpurun_out/${T}_tests.log
for mode in beam greedy; do timeout 300 python tools/decode_profile.py --mode $mode --batches 8 2>&1 | grep -v amdgpu | tail -1; done
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/${T}_prof -- python /root/repo/tools/decode_profile.py --mode beam --batches 4 > /dev/null 2>&1)
f=$(ls gpurun_out/${T}_prof/*/*kernel_stats.csv | head -1); grep "beam_t\|greedy_fin" $f | cut -c1-50,200-330
cp $f gpurun_out/${T}_beam_kernel_stats.csv; rm -rf gpurun_out/${T}_prof
