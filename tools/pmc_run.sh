#!/bin/bash
# developer tool (run via gpurun): HBM-side traffic counters per kernel, one PMC pass per counter
ulimit -c 0
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
WL=${1:-big:q4_0}; TAG=${2:-big_q4_0}
for ctr in FETCH_SIZE WRITE_SIZE; do
  out=gpurun_out/pmc_${TAG}_${ctr}
  rm -rf $out; mkdir -p $out
  NL_NO_GRAPH=1 timeout 420 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out -o p -- python3 bench.py --workload $WL --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > $out/log.txt 2>&1 < /dev/null
  f=$(ls $out/*counter_collection.csv $out/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then
    python3 - "$f" "$ctr" > gpurun_out/pmc_${TAG}_${ctr}_summary.csv <<'PY'
import csv, sys, collections
f, ctr = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0, 0.0])
with open(f) as fh:
    for row in csv.DictReader(fh):
        if row.get("Counter_Name") != ctr: continue
        k = row["Kernel_Name"]
        acc[k][0] += 1; acc[k][1] += float(row["Counter_Value"])
print("kernel,dispatches,mean_" + ctr)
for k, (n, s) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f'"{k}",{n},{s / n:.3f}')
PY
    head -12 gpurun_out/pmc_${TAG}_${ctr}_summary.csv
    rm -rf $out
  else
    tail -5 $out/log.txt; ls -R $out | head
  fi
done
