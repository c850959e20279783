"""Developer tool: per-token time of chained greedy decode of nano at several context lengths, the persistent launch
(nl_persist.h) against the launch plans (NL_PERSIST=0) -- where PD_MAX_PASSES comes from.  python tools/persist_limit.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nanollama_amd import gguf, model, synth
path = "/tmp/probe_nano_q8_0.gguf"
if not os.path.exists(path):
    synth.generate_gguf(path, synth.TIERS["nano"], "q8_0", mode="float")
g = gguf.load_gguf(path)
for knob in ("1", "0"):
    os.environ["NL_PERSIST"] = knob
    dev = model.load_llama_model(g)
    toks = synth.prompt_ids(1100, g.meta.vocab_size)
    dev.prefill(toks)
    out = []
    for pos0 in (64, 300, 470, 520, 600, 700, 800, 900, 980):
        dev.decode_greedy(5, pos0, 32)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); dev.decode_greedy(5, pos0, 32); best = min(best, time.perf_counter() - t0)
        out.append(f"pos {pos0}: {best / 32 * 1e6:.0f} us")
    print(("persistent launch " if knob == "1" else "launch plans      ") + "  ".join(out), dev.persist_info())
    dev.close()
