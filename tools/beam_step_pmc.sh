#!/bin/bash
# PMC counters of the kernels of one beam-5 decoding step (separate rocprofv3 --pmc passes, --kernel-trace only):
#   gpurun -- 'bash tools/beam_step_pmc.sh r06'   ->  gpurun_out/<round>_beam_step_pmc.txt
set -u
R=${1:-r06}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/bpmc_*
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
           "FETCH_SIZE TCC_HIT_sum TCC_MISS_sum" \
           "WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    i=$((i + 1))
    timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/bpmc_$i -- \
        python $ROOT/tools/decode_profile.py --mode beam --batches 2 > /tmp/bpmc_$i.log 2>&1 || echo "pass $i failed"
done
python $ROOT/tools/pmc_kernel.py /tmp/bpmc_1 /tmp/bpmc_2 /tmp/bpmc_3 /tmp/bpmc_4 \
    --match step_group_medium proj_astat attn_partial_fastq beam_tile_scan > $ROOT/gpurun_out/${R}_beam_step_pmc.txt 2>&1
head -80 $ROOT/gpurun_out/${R}_beam_step_pmc.txt
