#!/bin/bash
# skew of the two co-resident attention workgroups: census + prefill time for several sleep lengths
ulimit -c 0; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1; mkdir -p gpurun_out
for sk in ${SKEWS:-0 80}; do
  cd $GRAFT_REPO_ROOT/nanollama_amd/csrc
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -fvisibility=hidden -DNL_ATT_SKEW=$sk -DNL_ATT_STAMPS=40 -DNL_SRC_SHA=\"skew\" -DNL_GIT_HEAD=\"skew\" -shared -o /tmp/libnl_skew$sk.so nl_engine.hip -ldl 2>&1 | grep -E "error" | head
  cd $GRAFT_REPO_ROOT
  echo "== NL_ATT_SKEW=$sk"
  NL_LIB_PATH=/tmp/libnl_skew$sk.so python3 tools/att_stamps.py 2>&1 | grep -E "census|stored|LDS_ALLOC"
  NL_LIB_PATH=/tmp/libnl_skew$sk.so python3 -c "
import sys; sys.path.insert(0,'tools'); sys.argv=['x']
import bench_modes as b; b.prefill(); b.prefill()" 2>&1 | tail -2
done
