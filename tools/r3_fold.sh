#!/bin/bash
ulimit -c 0; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1; mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "prefill or folded or prompt or fused_attention_block or fast_exp" 2>&1 | tail -15) > gpurun_out/r3_fold_pytest.log; cat gpurun_out/r3_fold_pytest.log
for k in 1 0; do echo "== NL_FOLD_NORM=$k"; NL_FOLD_NORM=$k timeout 300 python -c "
import sys; sys.path.insert(0, 'tools')
import bench_modes as b
b.prefill(); b.prefill('goldie', 'q4_0', 2047)
"; done 2>&1 | tee gpurun_out/r3_fold_modes.log
