# developer tool: sweep the GEMV wavefront target (choose_geometry) on every tier
ulimit -c 0
cd "$GRAFT_REPO_ROOT" || exit 1
for w in 2048 3072 4096 6144; do
  for wl in nano:q8_0 mini:q4_0 goldie:q4_0 big:q4_0; do
    NL_WAVES=$w timeout 200 python bench.py --workload $wl --steps 96 --warmup 16 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('waves $w', '$wl', d['value'],'tok/s', {k:v['us_per_launch'] for k,v in d['kernels'].items() if k in ('qkv_rope','wo_resid','gate_up_swiglu','down_resid','lm_head')})"
  done
done
