#!/bin/bash
# round 3: tensor-parallel plan without reduce launches -- parity (rank processes on one GPU), then the shard-only probe
ulimit -c 0; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1; mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_p2p.py tests/test_gpu_parity.py -x -q -k "p2p or push or tensor_parallel or collective or missing_rank" 2>&1 | tail -12) > gpurun_out/r3_tp_pytest.log; cat gpurun_out/r3_tp_pytest.log
for n in 8 4 2; do
  (timeout 600 python bench.py --shard-of $n --steps 64 --warmup 8 2>gpurun_out/r3_shard_$n.err | tail -1) > gpurun_out/r3_shard_$n.json
  python3 -c "
import json,sys
d=json.load(open('gpurun_out/r3_shard_$n.json'))
pr=d['per_rank']['$n']
print('shard of $n:', pr.get('ms_per_step'), 'ms/step', pr.get('launches_per_step'), 'launches', {k:v['us_per_launch'] for k,v in pr.get('kernels',{}).items()}, d['predicted_scaling'])
" 2>&1 | tail -3
done
