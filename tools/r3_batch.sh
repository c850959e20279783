#!/bin/bash
ulimit -c 0; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1; mkdir -p gpurun_out
(timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/r3_batch_pytest.log; cat gpurun_out/r3_batch_pytest.log
for cfg in "1 1"; do set -- $cfg
  echo "== NL_ROPE_IN_ATTN=$1 NL_QG_RSTAGE=$2"
  NL_ROPE_IN_ATTN=$1 NL_QG_RSTAGE=$2 timeout 300 python -c "
import sys; sys.path.insert(0, 'tools')
import bench_modes as b
b.batch(); b.batch('nano', 'q8_0'); b.batch('goldie', 'q4_0', 16)
"; done 2>&1 | tee gpurun_out/r3_batch_modes.log
