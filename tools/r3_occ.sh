#!/bin/bash
ulimit -c 0; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1; mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I nanollama_amd/csrc tools/occ_probe.hip -o /tmp/occ_probe 2>&1 | grep error; timeout 60 /tmp/occ_probe | tee gpurun_out/r3_occ_probe.log
