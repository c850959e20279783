# developer tool (gpurun): the N>1 control flow of bench.py with 2 ranks sharing the box's single GPU.
# The nano replicas must report; the big tensor-parallel child cannot succeed (RCCL refuses two ranks on one
# device) and must come back as a captured error instead of taking the line down.
ulimit -c 0
cd $GRAFT_REPO_ROOT
export NL_BENCH_ONE_DEVICE=1 NL_TP_CHILD_TIMEOUT=150
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --steps 128 --warmup 16 > gpurun_out/tp2.log 2>&1; grep -v "^\[W\|^W0\|^\*\*\*" gpurun_out/tp2.log | head -40 | cut -c1-300
echo; echo "..."; 
