#!/bin/bash
# developer tool (run via gpurun): phase stamps of one workgroup of the decode-batch attention kernel (goldie x 64 streams)
ulimit -c 0; cd $GRAFT_REPO_ROOT/nanollama_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -fvisibility=hidden -DNL_ATTN_STAMPS -DNL_SRC_SHA=\"stamps\" -DNL_GIT_HEAD=\"stamps\" -shared -o /tmp/libnl_astamps.so nl_engine.hip -ldl 2>&1 | grep -E "error" | head
cd $GRAFT_REPO_ROOT
NL_LIB_PATH=/tmp/libnl_astamps.so python3 - <<'PY'
import os, sys, ctypes as C
sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
import numpy as np
import bench_modes as b
from nanollama_amd import model, _lib
g = b.gen("goldie", "q4_0")
ns = 64
dev = model.load_llama_model(g, max_streams=ns)
rng = np.random.Generator(np.random.PCG64(3))
ids = [int(t) for t in rng.integers(3, g.meta.vocab_size, size=ns)]
L = _lib.lib()
out = (C.c_longlong * 16)()
L.nl_debug_attn_stamps.argtypes = [C.POINTER(C.c_longlong)]
for pos in range(int(os.environ.get("POS0", "20")), int(os.environ.get("POS0", "20")) + 6):
    ids, _ = dev.forward_batch(list(range(ns)), ids, [pos] * ns)
    dev.synchronize()
    L.nl_debug_attn_stamps(out)
    names = ["entry", "loads issued + prologue done", "K staged + barrier", "scores + barrier", "softmax + barrier", "PV + partials in LDS + barrier", "finalize"]
    print(f"pos {pos}: " + "  ".join(f"{names[i]} +{out[i] - out[i - 1]}" for i in range(1, 7)) + f"  total {out[6] - out[0]}")
dev.close()
PY
