#!/bin/bash
# developer tool (run via gpurun): SQ counters of qgemm_kernel and qgemm2_kernel side by side on one shape of
# tools/qgemm2_bench.bin, one PMC pass per counter group.   bash tools/pmc_qgemm2.sh 2047 4 "goldie gate" 4
ulimit -c 0
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "GRBM_GUI_ACTIVE"; do
  out=gpurun_out/pmc_qg2; rm -rf $out; mkdir -p $out
  timeout 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out -o p -- tools/qgemm2_bench.bin "$@" > $out/log.txt 2>&1 < /dev/null
  f=$(ls $out/*counter_collection.csv $out/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then
    python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        n = row["Kernel_Name"]
        which = "qgemm2" if "qgemm2_kernel" in n else "qgemm3" if "qgemm3_kernel" in n else "qgemm " if "qgemm_kernel" in n else None
        if not which: continue
        k = (row["Counter_Name"], which); acc[k][0] += 1; acc[k][1] += float(row["Counter_Value"])
for (k, w), (n, s) in sorted(acc.items()): print(f"{k:28s} {w} {s / n:14.1f}  ({n} dispatches)")
PY
  else tail -3 $out/log.txt; fi
done
rm -rf gpurun_out/pmc_qg2
