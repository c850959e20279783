# developer tool (gpurun): exercise the RCCL path end to end with ONE rank under torch.distributed.run
ulimit -c 0
cd $GRAFT_REPO_ROOT
export NL_FORCE_TP_PLAN=1
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --workload big:q4_0 --steps 32 --warmup 8 --no-cpu-baseline 2>&1 | tail -4 | cut -c1-900
echo "--- eager"
NL_NO_GRAPH=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --workload big:q4_0 --steps 32 --warmup 8 --no-cpu-baseline 2>&1 | tail -2 | cut -c1-400
