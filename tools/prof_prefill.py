import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nanollama_amd import gguf, model, synth
tier = sys.argv[1] if len(sys.argv) > 1 else "mini"
path = f"/tmp/nl_modes_{tier}_q4_0.gguf"
if not os.path.exists(path):
    synth.generate_gguf(path, synth.TIERS[tier], "q4_0", mode="qrand")
g = gguf.load_gguf(path)
dev = model.load_llama_model(g)
toks = synth.prompt_ids(2047, g.meta.vocab_size)
dev.prefill(toks[:64]); dev.prefill(toks); dev.close()
