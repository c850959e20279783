#!/bin/bash
ulimit -c 0; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1; mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -25) > gpurun_out/r3_pytest2.log; cat gpurun_out/r3_pytest2.log
(timeout 900 python bench.py --steps 20 --warmup 5 2>gpurun_out/r3_bench2.err | tail -1) > gpurun_out/r3_bench2.json
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r3_bench2.json'))
print('nano', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], {k:v['us_per_launch'] for k,v in d['kernels'].items()})
s=d['secondary']; print('big', s.get('value'), s.get('ms_per_step'), s.get('hbm_frac_whole_step'), s.get('cpu_baseline',{}).get('value'))
print(json.dumps(d['other_configs'])[:900])
print(json.dumps(d.get('tensor_parallel_shard_probe',{}).get('predicted_scaling')), {k:v.get('ms_per_step') for k,v in d.get('tensor_parallel_shard_probe',{}).get('per_rank',{}).items()})
PY
(timeout 600 python bench.py --steps 512 --warmup 64 --no-secondary 2>/dev/null | tail -1) > gpurun_out/r3_bench2_512.json
python3 -c "
import json; d=json.load(open('gpurun_out/r3_bench2_512.json')); print('nano 512 steps', d['value'], d['ms_per_step'], d['config']['workload'])"
