#!/bin/bash
# round 4: the two-launches-per-layer tensor-parallel plan (nl_tp.h): parity tests, then the per-rank shard timing
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export NL_QUIET=1
python -m pytest tests/test_gpu_tp_fused.py -x -q 2>&1 | tail -40 > gpurun_out/r4_tp_tests.log
echo "--- p2p" >> gpurun_out/r4_tp_tests.log
python -m pytest tests/test_gpu_p2p.py -x -q 2>&1 | tail -30 >> gpurun_out/r4_tp_tests.log
for n in 8 4; do
  python bench.py --shard-of $n --steps 96 --warmup 16 > gpurun_out/r4_shard_of_$n.json 2> gpurun_out/r4_shard_of_$n.err
  NL_TP_FUSED=0 python bench.py --shard-of $n --steps 96 --warmup 16 > gpurun_out/r4_shard_of_${n}_mode2.json 2>> gpurun_out/r4_shard_of_$n.err
done
tail -5 gpurun_out/r4_tp_tests.log
