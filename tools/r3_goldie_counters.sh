#!/bin/bash
# round 3: what the decode-batch GEMM of goldie x 64 streams waits for -- texture / L2 counters of
# qgemm_kernel in the real batched step, one PMC pass per group (no tracing domains beside --kernel-trace)
ulimit -c 0; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r3_goldie_counters.txt; : > $out
rocprofv3 -L 2>/dev/null | grep -oE "\b(TCP|TA|TCC|TD|SQ|SPI|GRBM)_[A-Za-z0-9_]+" | sort -u > gpurun_out/r3_counter_names.txt
for grp in "SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "TA_BUSY_avr TA_TA_BUSY_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_TAG_STALL_sum TCC_BUSY_sum" "SPI_RA_LDS_CU_FULL_CSN SPI_RA_WAVE_SIMD_FULL_CSN"; do
  d=gpurun_out/pmc_g; rm -rf $d; mkdir -p $d
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $d -o p -- python3 tools/prof_batch.py goldie q4_0 64 > $d/log.txt 2>&1 < /dev/null
  f=$(ls $d/*counter_collection.csv $d/*/*counter_collection.csv 2>/dev/null | head -1)
  echo "== $grp" >> $out
  if [ -n "$f" ]; then
    python3 - "$f" >> $out <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        name = row["Kernel_Name"]
        if "qgemm_kernel" not in name and "bnorm" not in name: continue
        key = (name.split("(")[0][-60:], row["Grid_Size"], row["Counter_Name"])
        acc[key][0] += 1; acc[key][1] += float(row["Counter_Value"])
for (name, grid, c), (n, s) in sorted(acc.items()): print(f"{name:62s} grid {grid:>8s} {c:36s} {s / n:16.1f}  ({n} dispatches)")
PY
  else tail -2 $d/log.txt >> $out; fi
done
rm -rf gpurun_out/pmc_g
cat $out | cut -c1-200
