#!/bin/bash
# round 3: phase stamps of nano's two fused launches + one SQ / TCC counter pass for each, kept under profiles/
ulimit -c 0; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1; mkdir -p gpurun_out
(timeout 200 python3 tools/block_probe.py nano q8_0) > gpurun_out/r3_nano_block_stamps.txt 2>&1
cat > /tmp/nano_steps.py <<'PY'
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from nanollama_amd import gguf, model, synth
path = "/tmp/probe_nano_q8_0.gguf"
if not os.path.exists(path):
    synth.generate_gguf(path, synth.TIERS["nano"], "q8_0", mode="float")
dev = model.load_llama_model(gguf.load_gguf(path))
p = synth.prompt_ids(8, synth.TIERS["nano"].vocab)
dev.prefill(p)
import numpy as np
first = int(np.argmax(dev.state.logits))
for _ in range(4):
    dev.decode_greedy(first, 8, 64)
dev.close()
PY
(NL_NO_GRAPH=1 bash tools/pmc_kernel.sh attn_block_kernel /tmp/nano_steps.py) > gpurun_out/r3_nano_attn_block_counters.txt 2>&1
(NL_NO_GRAPH=1 bash tools/pmc_kernel.sh ffn_block_kernel /tmp/nano_steps.py) > gpurun_out/r3_nano_ffn_block_counters.txt 2>&1
cat gpurun_out/r3_nano_block_stamps.txt; cat gpurun_out/r3_nano_attn_block_counters.txt
