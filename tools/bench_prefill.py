"""Developer tool: 2047-token prompt time of mini / goldie Q4_0 in both precision modes (best of 5, host wall clock incl. the logits
read-back), and the logits' checksum (an A/B of two builds must print the same one).   python3 tools/bench_prefill.py [tier ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nanollama_amd import gguf, model, synth
for tier in (sys.argv[1:] or ["mini", "goldie"]):
    path = f"/tmp/nl_modes_{tier}_q4_0.gguf"
    if not os.path.exists(path):
        synth.generate_gguf(path, synth.TIERS[tier], "q4_0", mode="qrand")
    g = gguf.load_gguf(path)
    dev = model.load_llama_model(g)
    toks = synth.prompt_ids(2047, g.meta.vocab_size)
    for mode in ("", "fp16x1"):
        if mode: os.environ["NL_PREFILL_PRECISION"] = mode
        else: os.environ.pop("NL_PREFILL_PRECISION", None)
        dev.reset(); dev.prefill(toks); dev.synchronize()
        best = 1e9
        for _ in range(5):
            dev.reset(); t0 = time.perf_counter(); dev.prefill(toks); best = min(best, time.perf_counter() - t0)
        print(f"{tier} q4_0 2047-token prompt, {mode or 'hi+lo'}: {best * 1e3:.3f} ms  ({2047 / best:.0f} tok/s)  logits sum {float(dev.state.logits.astype(np.float64).sum()):.6f}")
    dev.close()
