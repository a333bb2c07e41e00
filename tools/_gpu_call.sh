export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gp_prof -- python $GRAFT_REPO_ROOT/tools/general_path_probe.py --train-only NM_NEMATUS_CLUSTER 1 2>&1 | grep "NM_NEMATUS" | tail -3
cd $GRAFT_REPO_ROOT
f=$(ls /tmp/gp_prof/*/*_kernel_stats.csv | head -1); head -60 $f > gpurun_out/r06_general_train_kernel_stats.csv; head -45 $f | cut -c1-170
