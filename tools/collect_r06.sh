#!/bin/bash
# round 6: measurements behind profiles/r06_*, by target (gpurun -- 'bash tools/collect_r06.sh <target>...').
#   bench    the driver's bench line (--steps 20 --warmup 5) and the default-length line (128-token segments)
#   prof     rocprofv3 per-kernel stats of the driver's bench command (nano) and of big; FETCH_SIZE / WRITE_SIZE passes of both
#   goldie   goldie Q4_0 x 64 decode streams: kernel stats and L2 / fabric request counters, dgemm path and NL_DGEMM=0
#   dgemm    tools/dgemm_bench.hip: the four launches of a goldie layer at 64 tokens, phase stamps, 16 / 32 tokens
#   dropin   tools/dropin_rates.py: per-call loops of a C host, big and nano (launch plans), call box on / off
#   prompts  tools/bench_short_prompt.py: short prompts, dgemm against the split-K / long-run GEMMs
ulimit -c 0; export TMPDIR=/tmp NL_QUIET=1; cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r06; mkdir -p $O
hip="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17"
pmc_by_kernel() {   # <csv> <counter>: mean per dispatch by kernel name
  python3 - "$1" "$2" <<'PY'
import csv, sys, collections
f, ctr = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0, 0.0])
with open(f) as fh:
    for row in csv.DictReader(fh):
        if row.get("Counter_Name") != ctr: continue
        acc[row["Kernel_Name"]][0] += 1; acc[row["Kernel_Name"]][1] += float(row["Counter_Value"])
print("kernel,dispatches,mean_" + ctr)
for k, (n, s) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f'"{k}",{n},{s / n:.3f}')
PY
}
for target in "$@"; do
case $target in
bench)
  (timeout 900 python bench.py --steps 20 --warmup 5 2>$O/bench_steps20.err | tail -1) > $O/r06_bench_n1_steps20.json.log
  cut -c1-600 $O/r06_bench_n1_steps20.json.log; echo
  (timeout 900 python bench.py 2>$O/bench_default.err | tail -1) > $O/r06_bench_n1.json.log
  cut -c1-600 $O/r06_bench_n1.json.log; echo ;;
prof)
  rm -rf gpurun_out/prof6; mkdir -p gpurun_out/prof6
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof6 -o nano -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/prof6/nano.log 2>&1 < /dev/null
  NL_NO_GRAPH=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof6 -o big -- python3 bench.py --workload big:q4_0 --steps 64 --warmup 8 --no-cpu-baseline > gpurun_out/prof6/big.log 2>&1 < /dev/null
  for t in nano big; do f=$(ls gpurun_out/prof6/${t}_kernel_stats.csv gpurun_out/prof6/*/${t}_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/r06_${t}_kernel_stats.csv && head -6 "$f" | cut -c1-170; done
  grep '^{' gpurun_out/prof6/nano.log | tail -1 > $O/r06_bench_n1_steps20_under_rocprof.json.log
  python3 - $(ls gpurun_out/prof6/nano_kernel_trace.csv gpurun_out/prof6/*/nano_kernel_trace.csv 2>/dev/null | head -1) > $O/r06_nano_pd_decode_dispatch_durations.csv <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "pd_decode_kernel" in r.get("Kernel_Name", "")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print("dispatch,duration_us")
for i, r in enumerate(rows):
    print(f'{i},{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:.2f}')
PY
  tr '\n' ' ' < $O/r06_nano_pd_decode_dispatch_durations.csv; echo
  for ctr in FETCH_SIZE WRITE_SIZE; do
    out=gpurun_out/pmc6_$ctr; rm -rf $out; mkdir -p $out
    timeout 400 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out -o p -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $out/log.txt 2>&1 < /dev/null
    f=$(ls $out/*counter_collection.csv $out/*/*counter_collection.csv 2>/dev/null | head -1)
    if [ -n "$f" ]; then
      pmc_by_kernel "$f" $ctr > $O/r06_nano_q8_0_pmc_${ctr}.csv
      python3 - "$f" "$ctr" >> $O/r06_nano_q8_0_pmc_${ctr}.csv <<'PY'
import csv, sys
f, ctr = sys.argv[1], sys.argv[2]
with open(f) as fh:
    pd = sorted(((int(r["Dispatch_Id"]), float(r["Counter_Value"])) for r in csv.DictReader(fh)
                 if r.get("Counter_Name") == ctr and "pd_decode_kernel" in r["Kernel_Name"]))
for i, (_, v) in enumerate(pd):
    print(f'"pd_decode_kernel dispatch {i}",1,{v:.3f}')
PY
      head -3 $O/r06_nano_q8_0_pmc_${ctr}.csv | cut -c1-160
    else tail -5 $out/log.txt; fi
    rm -rf $out
    # big: a short run (16 steps) so that the counter pass fits its limit
    out=gpurun_out/pmc6b_$ctr; rm -rf $out; mkdir -p $out
    NL_NO_GRAPH=1 timeout 800 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out -o p -- python3 bench.py --workload big:q4_0 --steps 16 --warmup 2 --no-cpu-baseline --no-shard-probe > $out/log.txt 2>&1 < /dev/null
    f=$(ls $out/*counter_collection.csv $out/*/*counter_collection.csv 2>/dev/null | head -1)
    if [ -n "$f" ]; then pmc_by_kernel "$f" $ctr > $O/r06_big_q4_0_pmc_${ctr}.csv; head -4 $O/r06_big_q4_0_pmc_${ctr}.csv | cut -c1-160; else tail -5 $out/log.txt; fi
    rm -rf $out
  done
  rm -rf gpurun_out/prof6 ;;
goldie)
  for dg in 1 0; do
    tag=$([ $dg = 1 ] && echo dgemm || echo splitk)
    rm -rf gpurun_out/prof6g; mkdir -p gpurun_out/prof6g
    NL_DGEMM=$dg timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof6g -o m -- python3 tools/prof_batch.py goldie q4_0 64 > gpurun_out/prof6g/log.txt 2>&1 < /dev/null
    f=$(ls gpurun_out/prof6g/*kernel_stats.csv gpurun_out/prof6g/*/*kernel_stats.csv 2>/dev/null | head -1)
    [ -n "$f" ] && cp "$f" $O/r06_goldie_q4_0_batch64_${tag}_kernel_stats.csv && head -8 "$f" | cut -c1-150
    for grp in "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
      out=gpurun_out/pmc6g; rm -rf $out; mkdir -p $out
      NL_DGEMM=$dg timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out -o p -- python3 tools/prof_batch.py goldie q4_0 64 > $out/log.txt 2>&1 < /dev/null
      f=$(ls $out/*counter_collection.csv $out/*/*counter_collection.csv 2>/dev/null | head -1)
      for c in $grp; do [ -n "$f" ] && pmc_by_kernel "$f" $c | grep -v repack | head -9 >> $O/r06_goldie_q4_0_batch64_${tag}_l2_counters.csv; done
      rm -rf $out
    done
    cat $O/r06_goldie_q4_0_batch64_${tag}_l2_counters.csv | cut -c1-150
    NL_DGEMM=$dg timeout 300 python3 tools/bench_batch64.py 2>&1 | tail -3 > $O/r06_goldie_q4_0_batch64_${tag}_ms_per_step.log; cat $O/r06_goldie_q4_0_batch64_${tag}_ms_per_step.log
  done
  rm -rf gpurun_out/prof6g ;;
dgemm)
  $hip -ffp-contract=off -fno-slp-vectorize -I nanollama_amd/csrc tools/dgemm_bench.hip -o /tmp/dgb 2>&1 | grep -E " error" | head
  $hip -ffp-contract=off -fno-slp-vectorize -DDG_STAMPS -I nanollama_amd/csrc tools/dgemm_bench.hip -o /tmp/dgbs 2>&1 | grep -E " error" | head
  (for n in 64 32 16; do timeout 60 /tmp/dgb $n 200; done; echo "== phase stamps (developer build, 20 launches each)"; timeout 60 /tmp/dgbs 64 20) > $O/r06_dgemm_bench.log 2>&1; cat $O/r06_dgemm_bench.log ;;
dropin)
  (timeout 400 python3 tools/dropin_rates.py big q4_0 48 2>&1 | tail -1; NL_NO_CALL_BOX=1 timeout 400 python3 tools/dropin_rates.py big q4_0 48 2>&1 | tail -1
   NL_PERSIST=0 timeout 200 python3 tools/dropin_rates.py nano q8_0 64 2>&1 | tail -1; NL_PERSIST=0 NL_NO_CALL_BOX=1 timeout 200 python3 tools/dropin_rates.py nano q8_0 64 2>&1 | tail -1
   timeout 200 python3 tools/dropin_rates.py nano q8_0 64 2>&1 | tail -1) > $O/r06_dropin_rates.log; cat $O/r06_dropin_rates.log ;;
prompts)
  (for t in mini goldie; do timeout 200 python3 tools/bench_short_prompt.py $t q4_0 24 64 127 192 256 512; NL_DGEMM=0 timeout 200 python3 tools/bench_short_prompt.py $t q4_0 24 64 127 192 256 512; done) > $O/r06_short_prompts.log 2>&1; cat $O/r06_short_prompts.log ;;
*) echo "unknown target $target" ;;
esac
done
