#!/bin/bash
ulimit -c 0; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1; mkdir -p gpurun_out
for v in "" "-DNL_ATT_OLD_PCONV" "-DNL_ATT_NO_SHIFT" "-DNL_ATT_OLD_PCONV -DNL_ATT_NO_SHIFT"; do
cd $GRAFT_REPO_ROOT/nanollama_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -fvisibility=hidden $v -DNL_SRC_SHA=\"dbg\" -DNL_GIT_HEAD=\"dbg\" -shared -o /tmp/libnl_dbg.so nl_engine.hip -ldl 2>&1 | grep -E "error" | head
cd $GRAFT_REPO_ROOT
echo "== variant: $v"
NL_LIB_PATH=/tmp/libnl_dbg.so python3 - <<'PY' 2>&1 | tail -3
import os, sys, numpy as np
sys.path.insert(0, '.')
from nanollama_amd import gguf, model
g = gguf.load_gguf('tests/golden/tiny_q4_0.gguf'); v = np.load('tests/golden/tiny_q4_0.npz')
toks = [int(t) for t in v['prompt']]
a = model.load_llama_model(g); b = model.load_llama_model(g)
a.prefill(toks)
for pos, t in enumerate(toks): b.forward(t, pos)
print('prefill vs token-at-a-time', np.abs(a.state.logits - b.state.logits).max(), 'vs golden', np.abs(a.state.logits - v["logits_full"][len(toks) - 1]).max())
PY
done
