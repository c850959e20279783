"""Developer tool: forward_batch at small stream counts vs the same tokens through the single-token plan."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nanollama_amd import gguf, model, synth
import bench_modes as b
for tier, wtype in (("nano", "q8_0"), ("goldie", "q4_0")):
    g = b.gen(tier, wtype)
    dev = model.load_llama_model(g, max_streams=64)
    rng = np.random.Generator(np.random.PCG64(3))
    for ns in (2, 3, 4, 6, 8, 16, 32, 64):
        toks = [int(t) for t in rng.integers(3, g.meta.vocab_size, size=ns)]
        streams = list(range(ns))
        for p in range(4):
            ids, _ = dev.forward_batch(streams, toks, [p] * ns)
        t0 = time.perf_counter()
        for k in range(16):
            ids, _ = dev.forward_batch(streams, ids, [4 + k] * ns)
        dt_b = (time.perf_counter() - t0) / 16
        t0 = time.perf_counter()
        for k in range(4):
            for s in streams:
                dev.forward_argmax(toks[s], 20 + k, stream=s)
        dt_s = (time.perf_counter() - t0) / 4
        print(f"{tier} {ns:3d} streams: batch step {dt_b*1e3:.3f} ms ({ns/dt_b:.0f} tok/s)   {ns} single steps {dt_s*1e3:.3f} ms ({ns/dt_s:.0f} tok/s)")
    dev.close()
