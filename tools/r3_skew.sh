#!/bin/bash
# attention kernel build variants (-D flags in $VARIANTS, ';'-separated): per-chunk stamps, census and mini prefill time
ulimit -c 0; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1; mkdir -p gpurun_out
IFS=';' read -ra VS <<< "${VARIANTS:-;-DNL_ATT_PRIO=8}"
k=0
for v in "${VS[@]}"; do
  k=$((k+1))
  cd $GRAFT_REPO_ROOT/nanollama_amd/csrc
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -fvisibility=hidden $v -DNL_ATT_STAMPS=${TILE:-20} -DNL_SRC_SHA=\"var\" -DNL_GIT_HEAD=\"var\" -shared -o /tmp/libnl_var$k.so nl_engine.hip -ldl 2>&1 | grep -E "error" | head
  cd $GRAFT_REPO_ROOT
  echo "== variant: $v"
  NL_LIB_PATH=/tmp/libnl_var$k.so python3 tools/att_stamps.py 2>&1 | grep -E "census|exit \+|chunk 1 "
  NL_LIB_PATH=/tmp/libnl_var$k.so python3 -c "
import sys; sys.path.insert(0,'tools'); sys.argv=['x']
import bench_modes as b; b.prefill(); b.prefill()" 2>&1 | tail -2
done
