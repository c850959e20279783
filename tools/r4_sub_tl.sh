#!/bin/bash
# round 4: kernel-trace timeline of a goldie x 64 decode step cut into concurrent groups (NL_SUB_BATCHES)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp NL_QUIET=1 NL_SUB_BATCHES=${SUB:-2}
rm -rf gpurun_out/tls; mkdir -p gpurun_out/tls
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tls -o t -- python3 tools/bench_subbatch.py one goldie q4_0 64 8 4 > gpurun_out/tls/run.log 2>&1
python3 - <<'PY' > gpurun_out/r4_subbatch_timeline_${SUB:-2}.txt
import csv, glob
rows = list(csv.DictReader(open(glob.glob("gpurun_out/tls/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-120:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{r['Kernel_Name'].split('<')[0].split('(')[0][-26:]:26s} q={r.get('Queue_Id','?'):>3s} start {0.001*(s-t0):9.2f} end {0.001*(e-t0):9.2f} dur {0.001*(e-s):6.2f}")
PY
tail -40 gpurun_out/r4_subbatch_timeline_${SUB:-2}.txt; tail -3 gpurun_out/tls/run.log
