set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
T=r03o
R=$PWD
cd /tmp
timeout 600 rocprofv3 --pmc TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/${T}_pmc1 -- python $R/tools/decode_profile.py --mode beam --batches 2 > pmc.log 2>&1
tail -2 pmc.log
timeout 600 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL --kernel-trace --output-format csv -d $R/gpurun_out/${T}_pmc2 -- python $R/tools/decode_profile.py --mode beam --batches 2 > pmc2.log 2>&1
tail -2 pmc2.log
cd $R
python tools/pmc_kernel.py gpurun_out/${T}_pmc1 gpurun_out/${T}_pmc2 --match step_group_medium "gemm_tiled<4, 2, 1, 2, false, false, true, 16, true" 2>&1 | head -60
rm -rf gpurun_out/${T}_pmc1 gpurun_out/${T}_pmc2
