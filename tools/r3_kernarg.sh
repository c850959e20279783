#!/bin/bash
ulimit -c 0; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1; mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 tools/kernarg_probe.hip -o /tmp/kp0 2>&1 | grep error
hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=16 tools/kernarg_probe.hip -o /tmp/kp16 2>&1 | grep error
(echo "no preload:"; timeout 60 /tmp/kp0; echo "-amdgpu-kernarg-preload-count=16:"; timeout 60 /tmp/kp16; echo "no preload (again):"; timeout 60 /tmp/kp0) | tee gpurun_out/r3_kernarg_probe.log
