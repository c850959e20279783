"""Per-launch times of one tier's decode plan (nl_profile_forward: every launch of the plan replayed back to back) and the chained
step, for A/B runs of library variants: python tools/kernel_times.py big q4_0   (NL_LIB_PATH, NL_MFMA_DOT, NL_PREFETCH select)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import bench  # noqa: E402
from nanollama_amd import gguf, model, synth  # noqa: E402

tier, wtype = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 32
g = gguf.load_gguf(bench.ensure_gguf(synth.TIERS[tier], wtype, "qrand" if tier == "big" else "float"))
dev = model.load_llama_model(g)
prompt = synth.prompt_ids(16, dev.config.vocab_size)
dev.prefill(prompt)
first = int(np.argmax(dev.state.logits))
best = None
for _ in range(5):
    dev.reset(); dev.prefill(prompt)
    t0 = time.perf_counter(); dev.decode_greedy(first, len(prompt), n); dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)
prof = dev.profile_forward(first, len(prompt), iters=20)
out = {"tier": tier, "wtype": wtype, "lib": os.environ.get("NL_LIB_PATH", "tree"), "NL_MFMA_DOT": os.environ.get("NL_MFMA_DOT"),
       "chained_ms_per_token": round(best / n * 1e3, 4), "plan": dev.plan_info(),
       "us_per_launch": {k: round(ms / max(c, 1) * 1e3, 2) for k, (ms, c) in prof.items() if c}}
print(json.dumps(out))
