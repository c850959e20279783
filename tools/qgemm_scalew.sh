#!/bin/bash
# developer tool (run via gpurun): the multi-token GEMM microbenchmark with and without the scale folded into the
# weight operand (-DNL_QG_SCALEW: 3 MFMAs per block, no per-block FMAs)
cd "$GRAFT_REPO_ROOT" || exit 1
for v in "" "-DNL_QG_SCALEW"; do
  out=/tmp/qgb_sw${v:+1}
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize $v -I nanollama_amd/csrc tools/qgemm_bench.hip -o $out 2>/dev/null || { echo "build failed $v"; continue; }
  echo "== ${v:-baseline}"
  for n in ${NS:-64 512 2047}; do timeout 120 $out $n 0 -1 50 | grep -E "mini|big|goldie gate|goldie down"; done
done
