# developer tool (run via gpurun): build a stamped copy of the library and print the phases of one attention workgroup
ulimit -c 0
cd $GRAFT_REPO_ROOT/nanollama_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -fvisibility=hidden -DNL_ATT_STAMPS=${1:-40} -DNL_SRC_SHA=\"stamps\" -DNL_GIT_HEAD=\"stamps\" -shared -o /tmp/libnl_stamps.so nl_engine.hip -ldl 2>&1 | grep -E "error" | head
cd $GRAFT_REPO_ROOT
NL_LIB_PATH=/tmp/libnl_stamps.so python3 tools/att_stamps.py 2>&1 | tail -120
