"""Developer tool: per-token time of chained greedy decode at several context lengths, fused plan vs five-launch plan
(NL_FUSED_MAX_POS decides which one serves a position)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nanollama_amd import gguf, model, synth
tier, wtype = (sys.argv[1], sys.argv[2]) if len(sys.argv) > 2 else ("nano", "q8_0")
path = f"/tmp/probe_{tier}_{wtype}.gguf"
if not os.path.exists(path):
    synth.generate_gguf(path, synth.TIERS[tier], wtype, mode="qrand" if tier in ("big", "goldie") else "float")
g = gguf.load_gguf(path)
for limit in ("0", "4096"):
    os.environ["NL_FUSED_MAX_POS"] = limit
    dev = model.load_llama_model(g)
    toks = synth.prompt_ids(1960, g.meta.vocab_size)
    dev.prefill(toks)
    out = []
    for pos0 in (64, 300, 600, 980, 1150, 1500, 1900):     # (the in-launch attention of modes 3 / 4 ends at position 512)
        dev.decode_greedy(5, pos0, 32)
        t0 = time.perf_counter()
        dev.decode_greedy(5, pos0, 32)
        out.append(f"pos {pos0}: {(time.perf_counter() - t0) / 32 * 1e6:.0f} us")
    print(("five-launch plan " if limit == "0" else "fused plan       ") + "  ".join(out))
    dev.close()
