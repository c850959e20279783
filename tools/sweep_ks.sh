#!/bin/bash
# developer tool (run via gpurun): split-K cap of the multi-token GEMM against the batched-decode step time
ulimit -c 0
cd "$GRAFT_REPO_ROOT" || exit 1
for ks in 1 2 4 8 16; do
  echo "== NL_KS_CAP=$ks"
  NL_KS_CAP=$ks timeout 200 python3 -c "
import sys; sys.path.insert(0,'tools')
import bench_modes as b
b.batch('goldie','q4_0',64); b.batch('goldie','q4_0',16); b.batch('nano','q8_0',64); b.batch('goldie','q4_0',8)" 2>&1 | tail -4
done
