#!/bin/bash
ulimit -c 0; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1; mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 tools/icache_probe.hip -o /tmp/icache_probe 2>/dev/null && timeout 120 /tmp/icache_probe | tee gpurun_out/r3_icache_probe.log
