ulimit -c 0
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_pf; mkdir -p gpurun_out/prof_pf
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pf -o m -- python3 tools/prof_prefill.py > gpurun_out/prof_pf/log.txt 2>&1 < /dev/null
f=$(ls gpurun_out/prof_pf/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then head -12 "$f" | cut -c1-150; fi
rm -f gpurun_out/prof_pf/*kernel_trace.csv
