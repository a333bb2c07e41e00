#!/bin/bash
# First GPU call of round 5 (written at the end of round 4, when the round's GPU minutes were spent):
#   gpurun --timeout 1500 -- 'bash tools/r05_first_call.sh'
# 1. the GPU tests that have never run (tests/test_pending_gpu.py);
# 2. the early optimizer half (NM_OPT_EARLY=1, trainers/generic_trainer.py) and the encoder-backward prologue inside the
#    BPTT loop's graph (NM_ENC_BWD_GRAPH=1, encoders/recurrent.py) against the default, same box:
#    training step time from tools/train_profile.py (median of its steps) and one rocprofv3 kernel trace for the
#    timeline (tools/trace_timeline.py);
# 3. the full GPU suite with NM_OPT_EARLY=1, if 2 says it pays (decide on the numbers, then flip the default).
# Everything lands under gpurun_out/r05_first/.
export TMPDIR=/tmp
out=gpurun_out/r05_first
mkdir -p $out
NM_RUN_PENDING=1 timeout 600 python -m pytest tests/test_pending_gpu.py -q -m gpu > $out/pending_tests.txt 2>&1
tail -5 $out/pending_tests.txt
# A/B/C/D on one box, twice: default | early optimizer half | encoder-backward prologue in the loop graph | both
for round in 1 2; do
    for mode in "0 0" "1 0" "0 1" "1 1"; do
        set -- $mode
        NM_OPT_EARLY=$1 NM_ENC_BWD_GRAPH=$2 timeout 300 python tools/train_profile.py --steps 30 \
            > $out/train_early$1_graph$2_r$round.txt 2>&1
    done
done
grep -H "ms" $out/train_early*.txt | tail -20
(cd /tmp && NM_OPT_EARLY=1 NM_ENC_BWD_GRAPH=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OLDPWD/$out/trace_early -- \
    python $OLDPWD/tools/train_profile.py --steps 6 > $OLDPWD/$out/trace_early.log 2>&1)
NM_OPT_EARLY=1 NM_ENC_BWD_GRAPH=1 timeout 900 python -m pytest tests -q -m gpu -x -k "training or background or engine or general or dp" \
    > $out/suite_opt_early.txt 2>&1
tail -3 $out/suite_opt_early.txt
