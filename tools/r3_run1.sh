#!/bin/bash
ulimit -c 0; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1; mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -25) > gpurun_out/r3_pytest1.log; cat gpurun_out/r3_pytest1.log
(timeout 600 python bench.py --steps 20 --warmup 5 2>gpurun_out/r3_bench1.err | tail -1) > gpurun_out/r3_bench1.json; cut -c1-1500 gpurun_out/r3_bench1.json
bash tools/r3_goldie_counters.sh > /dev/null 2>&1; wc -l gpurun_out/r3_goldie_counters.txt
