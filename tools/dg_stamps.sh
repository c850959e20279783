#!/bin/bash
# developer tool (run via gpurun): phase stamps of ONE kind of dgemm launch inside the real decode-batch step (goldie Q4_0 x 64 streams),
# from a -DDG_STAMPS -DDG_STAMP_EPI=<kind> copy of the library built into /tmp.   EPI=1 bash tools/dg_stamps.sh   (0 WO, 10 down, 1 gate || up, 2 Q|K|V, 3 LM head)
ulimit -c 0; cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}/nanollama_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -fvisibility=hidden -DDG_STAMPS -DDG_STAMP_EPI=${EPI:-1} -DNL_SRC_SHA=\"stamps\" -DNL_GIT_HEAD=\"stamps\" -shared -o /tmp/libnl_dgstamps.so nl_engine.hip -ldl 2>&1 | grep -E "error" | head
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
NL_LIB_PATH=/tmp/libnl_dgstamps.so NL_QUIET=1 python3 tools/dg_stamps.py
