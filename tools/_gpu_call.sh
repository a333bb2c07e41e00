set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
T=r2h
python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py tests/test_training_gpu.py tests/test_runners_gpu.py tests/test_distributed_cpu.py -q --timeout=900 -k "gemm or engine or training or runner or shard" > gpurun_out/${T}_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/${T}_tests.log
python tools/gemm_sweep.py 1 1+SK3 1+SK4 1+SK5 1+SK6 1+SK8 > gpurun_out/${T}_gemm_sweep.log 2>&1
python tools/decode_profile.py --mode greedy --batches 8 > gpurun_out/${T}_greedy.log 2>&1
python tools/decode_profile.py --mode beam --batches 8 > gpurun_out/${T}_beam.log 2>&1
python -m cProfile -s tottime tools/decode_profile.py --mode greedy --batches 20 2>&1 | head -45 > gpurun_out/${T}_cprofile_greedy.log
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --beam-batches 0 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
tail -3 gpurun_out/${T}_tests.log; cat gpurun_out/${T}_gemm_sweep.log; grep -v amdgpu gpurun_out/${T}_greedy.log gpurun_out/${T}_beam.log; cut -c1-400 gpurun_out/${T}_bench.json
