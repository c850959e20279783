ulimit -c 0
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_b; mkdir -p gpurun_out/prof_b
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b -o m -- python3 tools/prof_batch.py "$@" > gpurun_out/prof_b/log.txt 2>&1 < /dev/null
tail -2 gpurun_out/prof_b/log.txt
f=$(ls gpurun_out/prof_b/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then head -14 "$f" | cut -c1-170; fi
rm -f gpurun_out/prof_b/*kernel_trace.csv
