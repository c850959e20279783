# developer tool (gpurun): the driver's multi-GPU launch line with ONE rank (the only world size a 1-GPU box allows)
ulimit -c 0
cd $GRAFT_REPO_ROOT
timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 128 --warmup 16 --no-cpu-baseline 2>&1 | tail -2 | cut -c1-400
