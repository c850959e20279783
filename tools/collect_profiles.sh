#!/bin/bash
# developer tool (run via gpurun): everything profiles/ holds, from the build in the tree, in one call.
# Results land in gpurun_out/; copy them into profiles/ as described in profiles/README.md.
ulimit -c 0
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
bash scripts_gpu_run.sh prof
bash tools/prof_prefill.sh > gpurun_out/prof_pf_head.txt 2>&1
bash tools/prof_batch.sh > gpurun_out/prof_b_head.txt 2>&1
bash tools/pmc_run.sh nano:q8_0 nano_q8_0 > /dev/null 2>&1
bash tools/pmc_run.sh big:q4_0 big_q4_0 > /dev/null 2>&1
bash tools/pmc_kernel.sh qgemm_kernel tools/prof_batch.py > gpurun_out/pmc_qgemm_goldie_b64.txt 2>&1
ls gpurun_out
