#!/bin/bash
# round 5: measurements behind profiles/r05_*, by target (gpurun -- 'bash tools/collect_r05.sh <target>...').
#   xcd      tools/xcd_exchange_probe.hip: same-XCD hand-offs through the XCD's own L2 against the cross-XCD granule forms
#   tests    the whole -m gpu suite
#   persist  tests/test_gpu_persist.py (the persistent decode of nl_persist.h)
#   newtests the round's new parity tests of the wide tiers
#   nano     nano bench one-liner
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
  (timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -25) > $O/pytest_gpu.log; cat $O/pytest_gpu.log ;;
bench)
  (timeout 900 python bench.py --steps 20 --warmup 5 2>$O/bench_steps20.err | tail -1) > $O/r05_bench_n1_steps20.json.log
  cut -c1-700 $O/r05_bench_n1_steps20.json.log ;;
persist)
  (timeout 900 python -m pytest tests/test_gpu_persist.py -x -q -s 2>&1 | tail -40) > $O/pytest_persist.log; cat $O/pytest_persist.log ;;
group)
  (GPU_MAX_HW_QUEUES=8 timeout 900 python -m pytest tests/test_gpu_group.py -x -q -s 2>&1 | tail -25) > $O/pytest_group.log; cat $O/pytest_group.log ;;
newtests)
  (timeout 1500 python -m pytest tests/test_gpu_tp_fused.py tests/test_gpu_group.py tests/test_gpu_parity.py -q -k "big_attention_geometry or staying_blocks or group or fused_projection_attention_launch" 2>&1 | tail -15) > $O/pytest_newtests2.log; cat $O/pytest_newtests2.log ;;
newtests_unused)
  (timeout 1500 python -m pytest tests/test_gpu_tp_fused.py -x -q -s -k "big_attention_geometry or keep_out_of" 2>&1 | tail -15) > $O/pytest_newtests.log; cat $O/pytest_newtests.log ;;
sampling)
  (timeout 900 python -m pytest tests/test_gpu_sampling.py tests/test_sampling_kat.py -q 2>&1 | tail -8) > $O/pytest_sampling.log; cat $O/pytest_sampling.log
  python3 tools/bench_sampling.py > $O/r05_bench_sampling.log 2>&1; tail -12 $O/r05_bench_sampling.log ;;
m4)
  $hip tools/mfma4_probe.hip -o /tmp/m4p 2>/dev/null && /tmp/m4p > $O/r05_mfma4_probe.log 2>&1; cat $O/r05_mfma4_probe.log ;;
pstamps)
  (cd nanollama_amd/csrc && $hip -fPIC -ffp-contract=off -fno-slp-vectorize -fvisibility=hidden -DNL_PD_STAMPS -DNL_SRC_SHA=\"pdstamps\" -DNL_GIT_HEAD=\"pdstamps\" -shared -o /tmp/libnl_pd.so nl_engine.hip -ldl 2>&1 | grep -E " error" | head)
  NL_LIB_PATH=/tmp/libnl_pd.so python3 tools/persist_stamps.py 64 8 > $O/r05_persist_stamps.log 2>&1; cat $O/r05_persist_stamps.log ;;
nano)
  for rep in 1 2; do timeout 250 python bench.py --workload nano:q8_0 --steps 64 --warmup 8 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('nano', d['value'],'tok/s', d['ms_per_step'],'ms')"; done ;;
*) echo "unknown target $target" ;;
esac
done
