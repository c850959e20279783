#!/bin/bash
ulimit -c 0; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1; mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/mfma_valu_probe.hip -o /tmp/mfma_valu_probe 2>&1 | grep error
timeout 120 /tmp/mfma_valu_probe | tee gpurun_out/r3_mfma_valu_probe.log
