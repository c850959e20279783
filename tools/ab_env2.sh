#!/bin/bash
# developer tool (run via gpurun): big (and nano) bench with and without one environment knob, alternating, two rounds.
# usage: ab_env2.sh NAME=VALUE [workload ...]
cd "$GRAFT_REPO_ROOT" || exit 1
KV=$1; shift
WL=${@:-big:q4_0}
for rep in 1 2; do for kv in "" "$KV"; do for wl in $WL; do
  echo -n "${kv:-default} $wl: "
  env $kv timeout 300 python bench.py --workload $wl --steps 64 --warmup 8 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'],'tok/s', d['ms_per_step'],'ms', d.get('plan'), {k:v['us_per_launch'] for k,v in d['kernels'].items()})"
done; done; done
