#!/bin/bash
# prompt attention experiments: prefill parity tests, mini 2047-token prefill, workgroup census, per-kernel times, SQ counters
ulimit -c 0; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1; mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -q -x -k "prefill or prompt or batch" 2>&1 | tail -8) > gpurun_out/r3_live_pytest.log; cat gpurun_out/r3_live_pytest.log
(timeout 600 python tools/bench_modes.py 2>&1 | tail -30) > gpurun_out/r3_live_modes.log; cat gpurun_out/r3_live_modes.log
bash tools/att_stamps.sh 20 > gpurun_out/r3_live_census.log 2>&1; tail -16 gpurun_out/r3_live_census.log
bash tools/prof_prefill.sh 2>&1 | head -12
[ -n "$PMC" ] && bash tools/pmc_kernel.sh "attn_tile16_kernel<64, 4, 64, true>" tools/prof_prefill.py 2>&1 | grep -E "BANK|INST_LDS|MFMA_BUSY|INSTS_VALU|WAIT_ANY|WAVE_CYCLES|GRBM"
