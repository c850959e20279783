#!/bin/bash
# round 4: big Q4_0 on one GPU with non-temporal weight loads in the layer GEMVs (-DNL_NT_WEIGHTS build) against the default
ulimit -c 0; cd $GRAFT_REPO_ROOT/nanollama_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -fvisibility=hidden -DNL_NT_WEIGHTS -DNL_SRC_SHA=\"nt\" -DNL_GIT_HEAD=\"nt\" -shared -o /tmp/libnl_nt.so nl_engine.hip -ldl 2>&1 | grep -E "error" | head
cd $GRAFT_REPO_ROOT
for lib in "" /tmp/libnl_nt.so "" /tmp/libnl_nt.so; do
  echo "== lib: ${lib:-default}"
  for wl in big:q4_0 nano:q8_0; do
  NL_LIB_PATH=$lib timeout 250 python bench.py --workload $wl --steps 64 --warmup 8 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$wl', d['value'],'tok/s', d['ms_per_step'],'ms', {k:v['us_per_launch'] for k,v in d['kernels'].items()})"
  done
done
