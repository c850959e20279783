#!/bin/bash
ulimit -c 0; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1; mkdir -p gpurun_out
for mc in 6 8 12 16 32; do for mw in 1024; do
  echo "max_chunks=$mc max_wg=$mw: $(NL_QG_MAX_CHUNKS=$mc NL_QG_MAX_WG=$mw timeout 200 python3 -c "
import sys; sys.path.insert(0,'tools'); import bench_modes as b; b.batch('goldie','q4_0',64,steps=24)" 2>&1 | tail -1)"
done; done 2>&1 | tee gpurun_out/r3_sweep.log
