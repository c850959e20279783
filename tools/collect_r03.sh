#!/bin/bash
# round 3: every pass profiles/r03_* holds, from the build in the tree, in one gpurun call (results in gpurun_out/r03/)
ulimit -c 0; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03; rm -rf $O; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8) > $O/pytest_gpu.log
(timeout 900 python bench.py 2>$O/bench_n1.err | tail -1) > $O/r03_bench_n1.json.log
(timeout 900 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1) > $O/r03_bench_n1_steps20.json.log
# kernel stats (eager launches: rocprofv3 cannot trace graph replays on this image)
bash scripts_gpu_run.sh prof > $O/prof_head.txt 2>&1
cp gpurun_out/prof/nano_kernel_stats.csv $O/r03_nano_q8_0_kernel_stats.csv 2>/dev/null; cp gpurun_out/prof/big_kernel_stats.csv $O/r03_big_q4_0_kernel_stats.csv 2>/dev/null
bash tools/prof_prefill.sh > $O/prof_pf_head.txt 2>&1; cp gpurun_out/prof_pf/*kernel_stats.csv $O/r03_mini_q4_0_prefill2047_kernel_stats.csv 2>/dev/null
bash tools/prof_batch.sh > $O/prof_b_head.txt 2>&1; cp gpurun_out/prof_b/*kernel_stats.csv $O/r03_goldie_q4_0_batch64_kernel_stats.csv 2>/dev/null
bash tools/prof_sampling.sh > $O/prof_s_head.txt 2>&1; cp gpurun_out/prof_s/*kernel_stats.csv $O/r03_nano_q8_0_sampling_kernel_stats.csv 2>/dev/null
rm -rf gpurun_out/prof_l; mkdir -p gpurun_out/prof_l
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_l -o m -- python3 tools/prof_longctx.py > $O/r03_mini_q4_0_longctx_per_launch.txt 2>&1 < /dev/null
cp gpurun_out/prof_l/*kernel_stats.csv $O/r03_mini_q4_0_longctx_kernel_stats.csv 2>/dev/null; rm -rf gpurun_out/prof_l
# HBM traffic counters (one PMC pass per counter)
bash tools/pmc_run.sh nano:q8_0 nano_q8_0 > /dev/null 2>&1
bash tools/pmc_run.sh big:q4_0 big_q4_0 > /dev/null 2>&1
for t in nano_q8_0 big_q4_0; do for c in FETCH_SIZE WRITE_SIZE; do cp gpurun_out/pmc_${t}_${c}_summary.csv $O/r03_${t}_pmc_${c}.csv 2>/dev/null; done; done
# fused block launches of nano: phase stamps + SQ / TCC counters; decode-batch GEMM counters of goldie x 64
bash tools/r3_stamps.sh > /dev/null 2>&1
cp gpurun_out/r3_nano_block_stamps.txt $O/r03_nano_q8_0_block_phase_stamps.txt; cp gpurun_out/r3_nano_attn_block_counters.txt $O/r03_nano_q8_0_attn_block_sq_counters.txt; cp gpurun_out/r3_nano_ffn_block_counters.txt $O/r03_nano_q8_0_ffn_block_sq_counters.txt
bash tools/r3_goldie_counters.sh > /dev/null 2>&1; cp gpurun_out/r3_goldie_counters.txt $O/r03_goldie_q4_0_batch64_qgemm_bnorm_counters.txt
# prompt attention (mini, 2047 tokens): per-chunk stamps of one workgroup, census of every workgroup, SQ / TCC counters
(bash tools/att_stamps.sh 20; bash tools/pmc_kernel.sh "attn_tile16_kernel<64, 4, 64, true>" tools/prof_prefill.py) > $O/r03_mini_q4_0_prompt_attention_stamps_counters.txt 2>&1
# tensor parallelism: the shard-only probe and the N > 1 control flow with two ranks on the one GPU
(timeout 600 python bench.py --shard-of 8 --steps 96 --warmup 16 2>/dev/null | tail -1) > $O/r03_bench_shard_of_8.json.log
bash tools/tp2_flow.sh > $O/tp2_flow_head.txt 2>&1; grep '^{' gpurun_out/tp2.log | tail -1 > $O/r03_bench_tp2_one_device.json.log
ls -la $O; cat $O/pytest_gpu.log; cut -c1-400 $O/r03_bench_n1_steps20.json.log
