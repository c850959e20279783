#!/bin/bash
# developer tool (run via gpurun): SQ / TCC counters of the kernels matching $1 in `python3 $2 ...`, one PMC pass per group
ulimit -c 0
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
pat=$1; shift
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum"; do
  out=gpurun_out/pmc_k; rm -rf $out; mkdir -p $out
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out -o p -- python3 "$@" > $out/log.txt 2>&1 < /dev/null
  f=$(ls $out/*counter_collection.csv $out/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then
    python3 - "$f" "$pat" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        if sys.argv[2] not in row["Kernel_Name"]: continue
        k = row["Counter_Name"]; v = float(row["Counter_Value"]); acc[k][0] += 1; acc[k][1] += v; acc[k][2] = max(acc[k][2], v)
for k, (n, s, mx) in acc.items(): print(f"{k:28s} mean {s / n:14.1f}  max {mx:14.1f}  ({n} dispatches)")
PY
  else tail -3 $out/log.txt; fi
done
rm -rf gpurun_out/pmc_k
