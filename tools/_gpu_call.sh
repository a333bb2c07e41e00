set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
T=r03g
R=$PWD
cd /tmp
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/${T}_pmc1 -- python $R/tools/decode_profile.py --mode beam --batches 2 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/${T}_pmc2 -- python $R/tools/decode_profile.py --mode beam --batches 2 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${T}_pmc3 -- python $R/tools/decode_profile.py --mode beam --batches 2 > /dev/null 2>&1
cd $R
python tools/pmc_kernel.py gpurun_out/${T}_pmc1 gpurun_out/${T}_pmc2 gpurun_out/${T}_pmc3 --match step_group_medium > gpurun_out/${T}_medium_pmc.txt 2>&1
cat gpurun_out/${T}_medium_pmc.txt
rm -rf gpurun_out/${T}_pmc1 gpurun_out/${T}_pmc2 gpurun_out/${T}_pmc3
timeout 900 python -m pytest tests/test_transformer_gpu.py tests/test_general_gpu.py tests/test_training_gpu.py tests/test_kernels_gpu.py tests/test_step_graphs_gpu.py -x -q -m gpu > gpurun_out/${T}_tests.txt 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/${T}_tests.txt | cut -c1-400
python tools/transformer_bench.py --train-only 2>&1 | tail -2
