# developer tool (run via gpurun): kernel timeline of greedy decode -- start of every launch relative to its predecessor's
# end, duration, hardware queue.   bash tools/timeline.sh [tier] [wtype]
ulimit -c 0
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tl; mkdir -p gpurun_out/tl
timeout 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o t -- python3 tools/timeline.py run ${1:-nano} ${2:-q8_0} > gpurun_out/tl/log.txt 2>&1 < /dev/null
f=$(ls gpurun_out/tl/*kernel_trace.csv 2>/dev/null | head -1)
tail -2 gpurun_out/tl/log.txt
[ -n "$f" ] && python3 tools/timeline.py show $f | tail -${TL_ROWS:-40}
rm -f gpurun_out/tl/*kernel_trace.csv
