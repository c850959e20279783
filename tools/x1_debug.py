import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from nanollama_amd import gguf, model, synth
mode = sys.argv[1] if len(sys.argv) > 1 else "qrand"
path = f"/tmp/nl_x1dbg_mini_q4_0_{mode}.gguf"
if not os.path.exists(path): synth.generate_gguf(path, synth.TIERS["mini"], "q4_0", mode=mode)
g = gguf.load_gguf(path)
dev = model.load_llama_model(g)
for n in (300, 1920):
    toks = synth.prompt_ids(n, g.meta.vocab_size, seed=6)
    os.environ.pop("NL_PREFILL_PRECISION", None)
    dev.reset(); dev.prefill(toks); base = dev.state.logits.copy()
    for mode in ("fp16x1-gemm", "fp16x1-attn", "fp16x1"):
        os.environ["NL_PREFILL_PRECISION"] = mode
        dev.reset(); dev.prefill(toks)
        print(n, mode, float(np.abs(dev.state.logits - base).max()), float(base.std()), np.isnan(dev.state.logits).sum())
dev.close()
