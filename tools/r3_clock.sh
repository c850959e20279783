#!/bin/bash
ulimit -c 0; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1; mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 tools/clock_probe2.hip -o /tmp/clock_probe2 2>&1 | grep -E "error" ; timeout 120 /tmp/clock_probe2 | tee gpurun_out/r3_clock_probe2.log
