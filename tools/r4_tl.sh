#!/bin/bash
# round 4: kernel-trace timeline of the two-launch tensor-parallel layers (big Q4_0, rank 0's shard of 8, loopback)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp NL_QUIET=1
rm -rf gpurun_out/tl; mkdir -p gpurun_out/tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o t -- python3 tools/timeline.py run big q4_0 ${N:-8} > gpurun_out/tl/run.log 2>&1
python3 tools/timeline.py show $(ls gpurun_out/tl/*kernel_trace.csv | head -1) > gpurun_out/r4_tp_timeline_${N:-8}.txt
tail -30 gpurun_out/r4_tp_timeline_${N:-8}.txt
