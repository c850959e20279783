#!/bin/bash
# round 4: nano + big one-GPU bench one-liners (A/B after a kernel change)
cd "$(dirname "$0")/.." || exit 1
for rep in 1 2; do
for wl in nano:q8_0 big:q4_0 mini:q4_0; do
  timeout 250 python bench.py --workload $wl --steps 64 --warmup 8 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$wl', d['value'],'tok/s', d['ms_per_step'],'ms', {k:v['us_per_launch'] for k,v in d['kernels'].items()})"
done
done
