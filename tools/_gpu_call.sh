set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
T=r2k
python tools/train_profile.py --steps 20 > gpurun_out/${T}_base.log 2>&1
NM_MAIN_PRIO=1 python tools/train_profile.py --steps 20 > gpurun_out/${T}_prio.log 2>&1
NM_SIDE_STREAM=0 python tools/train_profile.py --steps 20 > gpurun_out/${T}_noside.log 2>&1
python tools/train_profile.py --steps 20 --batch 16 > gpurun_out/${T}_b16.log 2>&1
NM_SIDE_STREAM=0 python tools/train_profile.py --steps 20 --batch 16 > gpurun_out/${T}_b16_noside.log 2>&1
grep -h train gpurun_out/${T}_*.log
