#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export NL_QUIET=1
python -m pytest tests/test_gpu_tp_fused.py -x -q 2>&1 | tail -15 > gpurun_out/r4_tp_tests.log
N=8 bash tools/tp_stamps.sh > gpurun_out/r4_tp_stamps_8.log 2>&1
for ct in 1 2 4; do
  NL_TP_CT=$ct python bench.py --shard-of 8 --steps 96 --warmup 16 > gpurun_out/r4_shard_of_8_ct$ct.json 2> gpurun_out/r4_shard_of_8.err
done
NL_TP_PAIR=1 python bench.py --shard-of 8 --steps 96 --warmup 16 > gpurun_out/r4_shard_of_8_pair.json 2>> gpurun_out/r4_shard_of_8.err
for ct in 1 2; do
  NL_TP_CT=$ct python bench.py --shard-of 4 --steps 96 --warmup 16 > gpurun_out/r4_shard_of_4_ct$ct.json 2> gpurun_out/r4_shard_of_4.err
done
tail -5 gpurun_out/r4_tp_tests.log
