#!/bin/bash
# developer tool (run via gpurun): the multi-token GEMM microbenchmark built with several tile geometries
# (WAVES RT OCC KC per line; NS = token counts)
cd "$GRAFT_REPO_ROOT" || exit 1
for v in ${VARIANTS:-"8 1 4 4" "4 1 2 4" "4 2 2 4" "16 1 4 4" "8 1 6 2" "2 1 4 2"}; do
  set -- $v
  out=/tmp/qgb_$1_$2_$3_$4
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -DNL_QG_WAVES=$1 -DNL_QG_RT=$2 -DNL_QG_OCC=$3 -DNL_QG_KC=$4 \
     -I nanollama_amd/csrc tools/qgemm_bench.hip -o $out 2>/dev/null || { echo "build failed $v"; continue; }
  echo "== WAVES=$1 RT=$2 OCC=$3 KC=$4"
  for n in ${NS:-64 512 2047}; do timeout 120 $out $n 0 -1 50 | grep -E "mini|big|goldie gate|goldie down"; done
done
