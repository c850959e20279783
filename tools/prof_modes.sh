ulimit -c 0
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_modes; mkdir -p gpurun_out/prof_modes
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_modes -o m -- python3 tools/bench_modes.py > gpurun_out/prof_modes/log.txt 2>&1 < /dev/null
f=$(ls gpurun_out/prof_modes/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then head -16 "$f" | cut -c1-160; fi
rm -f gpurun_out/prof_modes/*kernel_trace.csv
