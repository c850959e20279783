#!/bin/bash
# round 4: every measurement profiles/r04_* holds, by target, from the build in the tree (gpurun -- 'bash tools/collect_r04.sh <target>...').
#   tp       parity tests of the two-launch tensor-parallel layers (nl_tp.h) + phase stamps / block census + per-rank shard times
#   tl       rocprofv3 kernel trace of a tp 8 shard's decode (launch durations incl. the boundary)
#   probes   transport microbenchmarks: in-launch all-gather (allgather_probe.hip), two-stream overlap (stream_overlap_probe.hip)
#   sub      goldie x 64 decode streams stepped as 1 / 2 / 4 concurrent groups (NL_SUB_BATCHES)
#   nt       big / nano with non-temporal weight loads (-DNL_NT_WEIGHTS build) against the default
#   x1       mini prefill per-kernel stats in both prompt precision modes (hi + lo, fp16x1) and the mode's logit error
#   ab       nano / big / mini one-GPU bench one-liners (A/B after a kernel change)
#   bench    the driver's bench line (python bench.py --steps 20 --warmup 5) and the default one
#   stats    rocprofv3 per-kernel stats of nano / big decode (eager launches), goldie x 64 decode streams, nano sampled decode;
#            HBM traffic counters of nano / big (one PMC pass per counter) -> r04_traffic.json
#   sampler  phase stamps of the sort-free top-p selection launch
#   ingest   tools/ingest_probe.hip
#   wide     big on one GPU: 2 launches per layer (mode 4 + wide_ffn_kernel) against NL_WIDE_FFN=0 and NL_ATTN_WO=0, alternating
#   tests    the whole -m gpu suite
ulimit -c 0; export TMPDIR=/tmp NL_QUIET=1; cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r04; mkdir -p $O
hip="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17"
for target in "$@"; do
case $target in
tp)
  python -m pytest tests/test_gpu_tp_fused.py tests/test_gpu_p2p.py -x -q 2>&1 | tail -6 > $O/tp_tests.log
  for n in 8 4; do
    N=$n bash tools/tp_stamps.sh > $O/r04_tp${n}_two_launch_layer_stamps.log 2>&1
    (python bench.py --shard-of $n --steps 96 --warmup 16 2>/dev/null | tail -1) > $O/r04_bench_shard_of_$n.json.log
    (NL_TP_FUSED=0 python bench.py --shard-of $n --steps 96 --warmup 16 2>/dev/null | tail -1) > $O/r04_bench_shard_of_${n}_four_launch_plan.json.log
  done
  for ct in 1 2 4; do (NL_TP_CT=$ct python bench.py --shard-of 8 --steps 96 --warmup 16 2>/dev/null | tail -1) > $O/shard_of_8_ct$ct.json.log; done
  (NL_TP_PAIR=1 python bench.py --shard-of 8 --steps 96 --warmup 16 2>/dev/null | tail -1) > $O/shard_of_8_pair.json.log
  cat $O/tp_tests.log ;;
tl)
  rm -rf gpurun_out/tl; mkdir -p gpurun_out/tl
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o t -- python3 tools/timeline.py run big q4_0 8 > gpurun_out/tl/run.log 2>&1
  python3 tools/timeline.py show $(ls gpurun_out/tl/*kernel_trace.csv | head -1) > $O/r04_tp8_two_launch_layer_kernel_trace.txt
  tail -12 $O/r04_tp8_two_launch_layer_kernel_trace.txt ;;
probes)
  $hip tools/allgather_probe.hip -o /tmp/agp 2>/dev/null && /tmp/agp > $O/r04_allgather_probe.log 2>&1
  $hip tools/stream_overlap_probe.hip -o /tmp/sop 2>/dev/null && /tmp/sop > $O/r04_stream_overlap_probe.log 2>&1
  tail -4 $O/r04_allgather_probe.log $O/r04_stream_overlap_probe.log ;;
sub)
  python3 tools/bench_subbatch.py > $O/r04_subbatch_groups.log 2>&1; cat $O/r04_subbatch_groups.log ;;
nt)
  (cd nanollama_amd/csrc && $hip -fPIC -ffp-contract=off -fno-slp-vectorize -fvisibility=hidden -DNL_NT_WEIGHTS -DNL_SRC_SHA=\"nt\" -DNL_GIT_HEAD=\"nt\" -shared -o /tmp/libnl_nt.so nl_engine.hip -ldl 2>&1 | grep -E " error" | head)
  for lib in "" /tmp/libnl_nt.so "" /tmp/libnl_nt.so; do
    echo "== lib: ${lib:-default}"
    for wl in big:q4_0 nano:q8_0; do
      NL_LIB_PATH=$lib timeout 250 python bench.py --workload $wl --steps 64 --warmup 8 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$wl', d['value'],'tok/s', d['ms_per_step'],'ms', {k:v['us_per_launch'] for k,v in d['kernels'].items()})"
    done
  done > $O/r04_nt_weights_ab.log 2>&1; cat $O/r04_nt_weights_ab.log ;;
x1)
  bash tools/prof_prefill.sh > /dev/null 2>&1; cp gpurun_out/prof_pf/*kernel_stats.csv $O/r04_mini_q4_0_prefill2047_kernel_stats.csv
  NL_PREFILL_PRECISION=fp16x1 bash tools/prof_prefill.sh > /dev/null 2>&1; cp gpurun_out/prof_pf/*kernel_stats.csv $O/r04_mini_q4_0_prefill2047_fp16x1_kernel_stats.csv
  python3 tools/x1_debug.py float > $O/r04_fp16x1_logit_error.log 2>&1; cat $O/r04_fp16x1_logit_error.log ;;
ab)
  for rep in 1 2; do for wl in nano:q8_0 big:q4_0 mini:q4_0; do
    timeout 250 python bench.py --workload $wl --steps 64 --warmup 8 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$wl', d['value'],'tok/s', d['ms_per_step'],'ms', {k:v['us_per_launch'] for k,v in d['kernels'].items()})"
  done; done ;;
bench)
  (timeout 900 python bench.py --steps 20 --warmup 5 2>$O/bench_steps20.err | tail -1) > $O/r04_bench_n1_steps20.json.log
  (timeout 900 python bench.py 2>$O/bench_n1.err | tail -1) > $O/r04_bench_n1.json.log
  cut -c1-600 $O/r04_bench_n1_steps20.json.log ;;
stats)
  bash scripts_gpu_run.sh prof > $O/prof_head.txt 2>&1
  cp gpurun_out/prof/nano_kernel_stats.csv $O/r04_nano_q8_0_kernel_stats.csv 2>/dev/null; cp gpurun_out/prof/big_kernel_stats.csv $O/r04_big_q4_0_kernel_stats.csv 2>/dev/null
  bash tools/prof_batch.sh > /dev/null 2>&1; cp gpurun_out/prof_b/*kernel_stats.csv $O/r04_goldie_q4_0_batch64_kernel_stats.csv 2>/dev/null
  bash tools/prof_sampling.sh > /dev/null 2>&1; cp gpurun_out/prof_s/*kernel_stats.csv $O/r04_nano_q8_0_sampling_kernel_stats.csv 2>/dev/null
  bash tools/pmc_run.sh nano:q8_0 nano_q8_0 > /dev/null 2>&1
  bash tools/pmc_run.sh big:q4_0 big_q4_0 > /dev/null 2>&1
  for t in nano_q8_0 big_q4_0; do for c in FETCH_SIZE WRITE_SIZE; do cp gpurun_out/pmc_${t}_${c}_summary.csv $O/r04_${t}_pmc_${c}.csv 2>/dev/null; done; done
  head -12 $O/r04_big_q4_0_kernel_stats.csv | cut -c1-160 ;;     # (then, in the repo: cp the csvs to profiles/ and python tools/make_traffic_json.py r04)
sampler)
  $hip -DNL_SAMP_STAMPS -Inanollama_amd/csrc tools/samp_probe.hip -o /tmp/sp 2>/dev/null && (/tmp/sp 1.0; /tmp/sp 0.2) > $O/r04_sampler_select_stamps.log 2>&1
  tail -18 $O/r04_sampler_select_stamps.log ;;
ingest)
  $hip tools/ingest_probe.hip -o /tmp/ip 2>/dev/null && /tmp/ip > $O/r04_ingest_probe.log 2>&1; tail -5 $O/r04_ingest_probe.log ;;
wide)
  (bash tools/ab_env2.sh NL_WIDE_FFN=0; bash tools/ab_env2.sh NL_ATTN_WO=0) > $O/r04_big_two_launch_layer_ab.log 2>&1; cat $O/r04_big_two_launch_layer_ab.log ;;
tests)
  (timeout 2000 python -m pytest tests -m gpu -q 2>&1 | tail -6) > $O/pytest_gpu.log; cat $O/pytest_gpu.log ;;
*) echo "unknown target $target" ;;
esac
done
