#!/bin/bash
# developer tool (run via gpurun): wall-clock phase stamps of the two-launch tensor-parallel layer (nl_tp.h) of big Q4_0,
# rank 0's shard of N (loopback), from a -DNL_TP_STAMPS=<layer + 1> copy of the library built into /tmp.   N=8 bash tools/tp_stamps.sh
ulimit -c 0; cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}/nanollama_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -fvisibility=hidden -DNL_TP_STAMPS=${LAYER:-21} -DNL_SRC_SHA=\"stamps\" -DNL_GIT_HEAD=\"stamps\" -shared -o /tmp/libnl_tpstamps.so nl_engine.hip -ldl 2>&1 | grep -E "error" | head
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
NL_LIB_PATH=/tmp/libnl_tpstamps.so python3 tools/tp_stamps.py ${N:-8}
