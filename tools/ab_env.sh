#!/bin/bash
# developer tool (run via gpurun): nano / big decode bench with and without one environment setting ($1, e.g. NL_KW=4)
ulimit -c 0
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do for setting in "NL_NONE=1" "$1"; do
  echo "== $setting"
  for wl in ${WLS:-nano:q8_0}; do
  env $setting timeout 250 python bench.py --workload $wl --steps 128 --warmup 16 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$wl', d['value'],'tok/s', d['ms_per_step'],'ms', {k:v['us_per_launch'] for k,v in d['kernels'].items() if k in ('qkv_rope','attention','wo_resid','gate_up_swiglu','down_resid','lm_head')})"
  done
done; done
