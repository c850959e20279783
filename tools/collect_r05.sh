#!/bin/bash
# round 5: measurements behind profiles/r05_*, by target (gpurun -- 'bash tools/collect_r05.sh <target>...').
#   xcd      tools/xcd_exchange_probe.hip: same-XCD hand-offs through the XCD's own L2 against the cross-XCD granule forms
#   tests    the whole -m gpu suite
#   persist  tests/test_gpu_persist.py (the persistent decode of nl_persist.h)
#   newtests the round's new parity tests of the wide tiers
#   nano     nano bench one-liner
#   dropin   tools/dropin_rates.py: nl_forward / nl_forward_argmax per call (integration/c/dropin_loop.c) against nl_decode_greedy
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
pfab)
  # NL_PREFETCH A/B on the big tier (nl_tp.h PfTiles): bits 1 attention -> ffn round 0, 2 ffn -> next projection, 4 ffn round-ahead
  for pf in ${PFAB_SET:-0 1 2 3 4 7}; do
    echo -n "NL_PREFETCH=$pf "; NL_PREFETCH=$pf timeout 600 python3 bench.py --workload big:q4_0 --steps 64 --warmup 8 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('kernels'))"
  done > $O/r05_prefetch_ab.log 2>&1; cat $O/r05_prefetch_ab.log ;;
mfab)
  # Q4_0 dot products of the wide fused launches on the matrix pipe (nl_tp.h mf_*) against the vector pipe: parity tests, then big A/B
  (timeout 1500 python -m pytest tests/test_gpu_tp_fused.py -x -q 2>&1 | tail -6) > $O/pytest_tp_fused.log; cat $O/pytest_tp_fused.log
  for mf in 0 1; do
    echo -n "NL_MFMA_DOT=$mf "; NL_MFMA_DOT=$mf timeout 600 python3 bench.py --workload big:q4_0 --steps 64 --warmup 8 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('greedy_ids_vs_cpu'), {k: v['us_per_launch'] for k, v in d.get('kernels', {}).items()})"
  done > $O/r05_mfma_dot_ab.log 2>&1; cat $O/r05_mfma_dot_ab.log ;;
ktimes)
  # per-launch times of big under the library variants: tree (matrix-pipe dots), NL_MFMA_DOT=0 (vector pipe), -DNL_FAKE_DOT (a quarter of the vector dots)
  (cd nanollama_amd/csrc && $hip -fPIC -ffp-contract=off -fno-slp-vectorize -fvisibility=hidden -DNL_FAKE_DOT -DNL_SRC_SHA=\"fakedot\" -DNL_GIT_HEAD=\"fakedot\" -shared -o /tmp/libnl_fake.so nl_engine.hip -ldl 2>&1 | grep -E " error" | head)
  (timeout 300 python3 tools/kernel_times.py big q4_0 2>&1 | tail -1; NL_MFMA_DOT=0 timeout 300 python3 tools/kernel_times.py big q4_0 2>&1 | tail -1
   NL_MFMA_DOT=0 NL_LIB_PATH=/tmp/libnl_fake.so timeout 300 python3 tools/kernel_times.py big q4_0 2>&1 | tail -1) > $O/r05_big_kernel_times_variants.log; cat $O/r05_big_kernel_times_variants.log ;;
fakedot)
  # how much of big's step is vector work?  A -DNL_FAKE_DOT copy of the library (a quarter of the Q4_0 dot products; wrong results) against the real one
  (cd nanollama_amd/csrc && $hip -fPIC -ffp-contract=off -fno-slp-vectorize -fvisibility=hidden -DNL_FAKE_DOT -DNL_SRC_SHA=\"fakedot\" -DNL_GIT_HEAD=\"fakedot\" -shared -o /tmp/libnl_fake.so nl_engine.hip -ldl 2>&1 | grep -E " error" | head)
  (echo "real:"; timeout 600 python3 tools/dropin_rates.py big q4_0 32 2>&1 | tail -1 | cut -c1-200; echo "a quarter of the Q4_0 dot-product instructions (wrong results):"; NL_LIB_PATH=/tmp/libnl_fake.so timeout 600 python3 tools/dropin_rates.py big q4_0 32 2>&1 | tail -1 | cut -c1-200) > $O/r05_fake_dot.log; cat $O/r05_fake_dot.log ;;
dropin)
  # per-call loops of a C host against the chained loop, nano Q8_0 (and with the resident session off)
  for sess in 1 0; do NL_PERSIST_SESSION=$sess timeout 600 python3 tools/dropin_rates.py nano q8_0 2>&1 | tail -3; done > $O/r05_dropin_rates.log; cat $O/r05_dropin_rates.log ;;
persist)
  (timeout 900 python -m pytest tests/test_gpu_persist.py -x -q -s 2>&1 | tail -40) > $O/pytest_persist.log; cat $O/pytest_persist.log ;;
group)
  (GPU_MAX_HW_QUEUES=8 timeout 900 python -m pytest tests/test_gpu_group.py -x -q -s 2>&1 | tail -25) > $O/pytest_group.log; cat $O/pytest_group.log ;;
newtests)
  (timeout 1500 python -m pytest tests/test_gpu_tp_fused.py tests/test_gpu_group.py tests/test_gpu_parity.py -q -k "big_attention_geometry or staying_blocks or group or fused_projection_attention_launch" 2>&1 | tail -15) > $O/pytest_newtests2.log; cat $O/pytest_newtests2.log ;;
newtests_unused)
  (timeout 1500 python -m pytest tests/test_gpu_tp_fused.py -x -q -s -k "big_attention_geometry or keep_out_of" 2>&1 | tail -15) > $O/pytest_newtests.log; cat $O/pytest_newtests.log ;;
prof)
  # rocprofv3 per-kernel stats of the driver's own bench command (nano: the persistent launch) and of big; HBM traffic counters, one PMC pass per counter
  rm -rf gpurun_out/prof5; mkdir -p gpurun_out/prof5
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof5 -o nano -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/prof5/nano.log 2>&1 < /dev/null
  NL_NO_GRAPH=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof5 -o big -- python3 bench.py --workload big:q4_0 --steps 64 --warmup 8 --no-cpu-baseline > gpurun_out/prof5/big.log 2>&1 < /dev/null
  for t in nano big; do f=$(ls gpurun_out/prof5/${t}_kernel_stats.csv gpurun_out/prof5/*/${t}_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/r05_${t}_kernel_stats.csv && head -8 "$f" | cut -c1-170; done
  grep '^{' gpurun_out/prof5/nano.log | tail -1 > $O/r05_bench_n1_steps20_under_rocprof.json.log
  # every dispatch of the persistent launch in order (the bench issues 1 x 128 tokens, 1 x 5 (warm-up), 5 x 20 timed, 7 x 20 profiled)
  python3 - $(ls gpurun_out/prof5/nano_kernel_trace.csv gpurun_out/prof5/*/nano_kernel_trace.csv 2>/dev/null | head -1) > $O/r05_nano_pd_decode_dispatch_durations.csv <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "pd_decode_kernel" in r.get("Kernel_Name", "")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print("dispatch,duration_us")
for i, r in enumerate(rows):
    print(f'{i},{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:.2f}')
PY
  cat $O/r05_nano_pd_decode_dispatch_durations.csv | tr '\n' ' '; echo
  for ctr in FETCH_SIZE WRITE_SIZE; do
    out=gpurun_out/pmc5_$ctr; rm -rf $out; mkdir -p $out
    timeout 400 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out -o p -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $out/log.txt 2>&1 < /dev/null
    f=$(ls $out/*counter_collection.csv $out/*/*counter_collection.csv 2>/dev/null | head -1)
    if [ -n "$f" ]; then python3 - "$f" "$ctr" > $O/r05_nano_q8_0_pmc_${ctr}.csv <<'PY'
import csv, sys, collections
f, ctr = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0, 0.0])
with open(f) as fh:
    for row in csv.DictReader(fh):
        if row.get("Counter_Name") != ctr: continue
        k = row["Kernel_Name"]
        acc[k][0] += 1; acc[k][1] += float(row["Counter_Value"])
print("kernel,dispatches,mean_" + ctr)
for k, (n, s) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f'"{k}",{n},{s / n:.3f}')
# the persistent launch dispatch by dispatch (dispatch order = Dispatch_Id)
with open(f) as fh:
    pd = sorted(((int(r["Dispatch_Id"]), float(r["Counter_Value"])) for r in csv.DictReader(fh)
                 if r.get("Counter_Name") == ctr and "pd_decode_kernel" in r["Kernel_Name"]))
for i, (_, v) in enumerate(pd):
    print(f'"pd_decode_kernel dispatch {i}",1,{v:.3f}')
PY
      head -4 $O/r05_nano_q8_0_pmc_${ctr}.csv | cut -c1-160; grep "pd_decode_kernel dispatch" $O/r05_nano_q8_0_pmc_${ctr}.csv | tr '\n' ' '; echo
    else tail -5 $out/log.txt; fi
    rm -rf $out
  done
  rm -rf gpurun_out/prof5 ;;
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
