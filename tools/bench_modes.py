"""Developer tool: prefill throughput (BASELINE config 3) and 64-stream batched decode (config 4)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nanollama_amd import gguf, model, synth

def gen(tier, wtype):
    path = f"/tmp/nl_modes_{tier}_{wtype}.gguf"
    if not os.path.exists(path):
        synth.generate_gguf(path, synth.TIERS[tier], wtype, mode="qrand")
    return gguf.load_gguf(path)

def prefill(tier="mini", wtype="q4_0", n=2047):
    g = gen(tier, wtype)
    dev = model.load_llama_model(g)
    toks = synth.prompt_ids(n, g.meta.vocab_size)
    dev.prefill(toks[:128])           # warm-up (allocates the batch buffers)
    dev.synchronize()
    dt = 1e9
    for rep in range(3):
        dev.reset()
        t0 = time.perf_counter(); dev.prefill(toks); dt = min(dt, time.perf_counter() - t0)
    first = int(np.argmax(dev.state.logits))
    t1 = time.perf_counter(); ids = dev.decode_greedy(first, n, 1); dt2 = time.perf_counter() - t1
    print(f"{tier} {wtype} prefill {n} tokens: {dt*1e3:.1f} ms = {n/dt:.0f} tok/s; next decode step {dt2*1e3:.2f} ms")
    dev.close()

def batch(tier="goldie", wtype="q4_0", nstreams=64, steps=32, pos0=8):
    g = gen(tier, wtype)
    dev = model.load_llama_model(g, max_streams=nstreams)
    rng = np.random.Generator(np.random.PCG64(3))
    toks = [int(t) for t in rng.integers(3, g.meta.vocab_size, size=nstreams)]
    streams = list(range(nstreams))
    for p in range(pos0):
        ids, _ = dev.forward_batch(streams, toks, [p] * nstreams)
    t0 = time.perf_counter()
    for k in range(steps):
        ids, _ = dev.forward_batch(streams, ids, [pos0 + k] * nstreams)
    dt = time.perf_counter() - t0
    print(f"{tier} {wtype} {nstreams} streams: {dt/steps*1e3:.3f} ms/step = {nstreams*steps/dt:.0f} tok/s aggregate")
    dev.close()

if __name__ == "__main__":
    prefill(); batch()
    prefill("nano", "q8_0", 1024); batch("nano", "q8_0")
