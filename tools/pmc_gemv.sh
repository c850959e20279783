#!/bin/bash
# developer tool (run via gpurun): SQ counters of the decode GEMV kernels, one PMC pass per counter group
ulimit -c 0
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
WL=${1:-big:q4_0}
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_SALU"; do
  out=gpurun_out/pmc_gv; rm -rf $out; mkdir -p $out
  NL_NO_GRAPH=1 timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out -o p -- python3 bench.py --workload $WL --steps 4 --warmup 1 --no-cpu-baseline > $out/log.txt 2>&1 < /dev/null
  f=$(ls $out/*counter_collection.csv $out/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then
    python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        k = row["Kernel_Name"]
        if "gemv_kernel" not in k: continue
        key = (k[k.index("gemv_kernel"):k.index("gemv_kernel") + 24], row["Counter_Name"])
        acc[key][0] += 1; acc[key][1] += float(row["Counter_Value"])
for (k, c), (n, s) in sorted(acc.items()): print(f"{k:26s} {c:24s} {s / n:14.1f}")
PY
  else tail -3 $out/log.txt; fi
done
rm -rf gpurun_out/pmc_gv
