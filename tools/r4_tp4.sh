#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export NL_QUIET=1
N=${N:-8} bash tools/tp_stamps.sh > gpurun_out/r4_tp_stamps_${N:-8}.log 2>&1
tail -12 gpurun_out/r4_tp_stamps_${N:-8}.log
