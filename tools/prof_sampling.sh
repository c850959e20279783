ulimit -c 0
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_s; mkdir -p gpurun_out/prof_s
NL_NO_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_s -o m -- python3 tools/bench_sampling.py > gpurun_out/prof_s/log.txt 2>&1 < /dev/null
grep "tok/s" gpurun_out/prof_s/log.txt
f=$(ls gpurun_out/prof_s/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then head -24 "$f" | cut -c1-200; fi
rm -f gpurun_out/prof_s/*kernel_trace.csv
