#!/bin/bash
# developer tool (run via gpurun): bench one-liners for the in-tree library and an alternative build (NL_LIB_PATH)
ulimit -c 0
cd "$GRAFT_REPO_ROOT" || exit 1
for lib in "" "$GRAFT_REPO_ROOT/$1" ""  "$GRAFT_REPO_ROOT/$1"; do
  echo "== lib: ${lib:-default}"
  for wl in nano:q8_0 big:q4_0; do
  NL_LIB_PATH=$lib timeout 250 python bench.py --workload $wl --steps 64 --warmup 8 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$wl', d['value'],'tok/s', d['ms_per_step'],'ms', {k:v['us_per_launch'] for k,v in d['kernels'].items() if k in ('qkv_rope','attention','wo_resid','gate_up_swiglu','down_resid','lm_head')})"
  done
done
