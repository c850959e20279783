#!/bin/bash
# usage: scripts_gpu_run.sh [pytest] [bench] [prof]  -- helper executed on the GPU box via gpurun
ulimit -c 0
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for what in "$@"; do
  case $what in
    pytest) (timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/pytest_gpu.log; cat gpurun_out/pytest_gpu.log;;
    smoke) (timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3) > gpurun_out/smoke.log; cat gpurun_out/smoke.log;;
    bench) (timeout 300 python bench.py 2>&1 | tail -3) > gpurun_out/bench.log; cat gpurun_out/bench.log;;
    prof) rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
          # headline workload only (--no-secondary skips big and the side configs); big gets its own pass
          NL_NO_GRAPH=1 timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o nano -- python3 bench.py --steps 256 --warmup 32 --no-cpu-baseline --no-secondary > gpurun_out/bench_prof.log 2>&1 < /dev/null
          NL_NO_GRAPH=1 timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o big -- python3 bench.py --workload big:q4_0 --steps 64 --warmup 8 --no-cpu-baseline > gpurun_out/bench_prof_big.log 2>&1 < /dev/null
          grep '^{' gpurun_out/bench_prof.log | tail -1 | cut -c1-300
          for f in gpurun_out/prof/nano_kernel_stats.csv gpurun_out/prof/big_kernel_stats.csv; do [ -f "$f" ] && head -12 "$f" | cut -c1-150; done
          rm -f gpurun_out/prof/*kernel_trace.csv gpurun_out/prof/*/*kernel_trace.csv;;
  esac
done
