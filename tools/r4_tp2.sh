#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export NL_QUIET=1
python -m pytest tests/test_gpu_tp_fused.py -x -q 2>&1 | tail -15 > gpurun_out/r4_tp_tests.log
python -m pytest tests/test_gpu_p2p.py -x -q 2>&1 | tail -15 >> gpurun_out/r4_tp_tests.log
N=8 bash tools/tp_stamps.sh > gpurun_out/r4_tp_stamps_8.log 2>&1
N=4 bash tools/tp_stamps.sh > gpurun_out/r4_tp_stamps_4.log 2>&1
for n in 8 4; do
  python bench.py --shard-of $n --steps 96 --warmup 16 > gpurun_out/r4_shard_of_$n.json 2> gpurun_out/r4_shard_of_$n.err
done
tail -5 gpurun_out/r4_tp_tests.log
