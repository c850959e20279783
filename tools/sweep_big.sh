#!/bin/bash
# developer tool: sweep GEMV geometry overrides on the big Q4_0 tier
ulimit -c 0
cd "$GRAFT_REPO_ROOT" || exit 1
for cfg in "0 0" "4 1" "8 1" "4 2" "2 4" "2 2" "1 8"; do
  set -- $cfg
  echo "== NL_KW=$1 NL_TW=$2"
  NL_KW=$1 NL_TW=$2 timeout 200 python bench.py --workload ${WL:-big:q4_0} --steps 48 --warmup 8 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'],'tok/s', d['ms_per_step'],'ms', {k:(v['us_per_launch'],v['GBps']) for k,v in d['kernels'].items() if k in ('qkv_rope','wo_resid','gate_up_swiglu','down_resid','lm_head')})"
done
