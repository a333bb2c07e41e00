set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
T=r03dp
NM_DIST_BACKEND=gloo timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --beam-batches 2 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
echo "bench rc=$?"
tail -c 600 gpurun_out/${T}_bench.err
