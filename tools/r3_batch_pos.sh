#!/bin/bash
# goldie x 64 streams: step time against the position of the streams
ulimit -c 0; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1; mkdir -p gpurun_out
python3 - <<'PY' 2>&1 | tee gpurun_out/r3_batch_pos.log
import os, sys, time
sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
import numpy as np
import bench_modes as b
from nanollama_amd import model
g = b.gen("goldie", "q4_0")
ns = 64
dev = model.load_llama_model(g, max_streams=ns)
rng = np.random.Generator(np.random.PCG64(3))
ids = [int(t) for t in rng.integers(3, g.meta.vocab_size, size=ns)]
streams = list(range(ns))
for pos0 in (8, 100, 130, 300, 600, 1000, 1900):
    for p in range(pos0 - 4, pos0):
        ids, _ = dev.forward_batch(streams, ids, [p] * ns)
    dev.synchronize()
    t0 = time.perf_counter()
    for k in range(16):
        ids, _ = dev.forward_batch(streams, ids, [pos0 + k] * ns)
    dt = (time.perf_counter() - t0) / 16
    print(f"positions {pos0}..{pos0 + 15}: {dt * 1e3:.3f} ms/step, {ns / dt:.0f} tok/s aggregate")
PY
