#!/bin/bash
# run length of the prompt attention workgroups (NL_ATT_RUN): mini 2047-token prefill + per-kernel times
ulimit -c 0; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1; mkdir -p gpurun_out
for r in 0 1 2 3 6; do
  echo "== NL_ATT_RUN=$r"
  NL_ATT_RUN=$r python3 -c "
import sys; sys.path.insert(0,'tools'); sys.argv=['x']
import bench_modes as b; b.prefill(); b.prefill()" 2>&1 | tail -1
  export NL_ATT_RUN=$r; bash tools/prof_prefill.sh 2>&1 | grep -E "attn_tile16_kernel<64, 4, 64|battn_merge" | cut -d, -f1-4,7; unset NL_ATT_RUN
done
