#!/bin/bash
# round 5: measurements behind profiles/r05_*, by target (gpurun -- 'bash tools/collect_r05.sh <target>...').
#   xcd      tools/xcd_exchange_probe.hip: same-XCD hand-offs through the XCD's own L2 against the cross-XCD granule forms
#   tests    the whole -m gpu suite
#   bench    the driver's bench line (python bench.py --steps 20 --warmup 5)
ulimit -c 0; export TMPDIR=/tmp NL_QUIET=1; cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r05; mkdir -p $O
hip="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17"
for target in "$@"; do
case $target in
xcd)
  $hip tools/xcd_exchange_probe.hip -o /tmp/xcdp 2>&1 | grep -E "error" | head
  timeout 600 /tmp/xcdp > $O/r05_xcd_exchange_probe.log 2>&1; cat $O/r05_xcd_exchange_probe.log ;;
tests)
  (timeout 2000 python -m pytest tests -m gpu -q -x 2>&1 | tail -8) > $O/pytest_gpu.log; cat $O/pytest_gpu.log ;;
bench)
  (timeout 900 python bench.py --steps 20 --warmup 5 2>$O/bench_steps20.err | tail -1) > $O/r05_bench_n1_steps20.json.log
  cut -c1-700 $O/r05_bench_n1_steps20.json.log ;;
*) echo "unknown target $target" ;;
esac
done
