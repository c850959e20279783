#!/bin/bash
# developer tool (run via gpurun): SQ / TCC counters per kernel of a native binary: tools/pmc_bin.sh <binary> [args...]
ulimit -c 0
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE"; do
  out=gpurun_out/pmc_b; rm -rf $out; mkdir -p $out
  timeout 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out -o p -- "$@" > $out/log.txt 2>&1 < /dev/null
  f=$(ls $out/*counter_collection.csv $out/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then
    python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        if "dgemm" not in row["Kernel_Name"] and "qgemm" not in row["Kernel_Name"]: continue
        k = (row["Kernel_Name"][:44], row["Counter_Name"]); acc[k][0] += 1; acc[k][1] += float(row["Counter_Value"])
for (kn, c), (n, s) in sorted(acc.items()): print(f"{kn:44s} {c:28s} mean {s / n:16.1f} ({n})")
PY
  else tail -3 $out/log.txt; fi
done
rm -rf gpurun_out/pmc_b
