"""Developer tool: 64-stream batched decode steps at a long position (for rocprofv3 --kernel-trace --stats)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench_modes as b
from nanollama_amd import model
pos0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
g = b.gen("goldie", "q4_0")
ns = 64
dev = model.load_llama_model(g, max_streams=ns)
rng = np.random.Generator(np.random.PCG64(3))
ids = [int(t) for t in rng.integers(3, g.meta.vocab_size, size=ns)]
for k in range(8):
    ids, _ = dev.forward_batch(list(range(ns)), ids, [pos0 + k] * ns)
dev.close()
