set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
T=r3m
NM_DIST_BACKEND=gloo timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/${T}_dp2.json 2> gpurun_out/${T}_dp2.err
echo "dp2 rc=$?"
tail -c 1500 gpurun_out/${T}_dp2.json
tail -3 gpurun_out/${T}_dp2.err | cut -c1-300
