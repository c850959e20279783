"""Per-call Forward against the chained loop on one tier: tokens/s of integration/c/dropin_loop.c (nl_forward + host argmax, the
patched Go loop of go/main.go:173-219; nl_forward_argmax) next to nl_decode_greedy.  python tools/dropin_rates.py nano q8_0"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from nanollama_amd import gguf, model, synth  # noqa: E402

tier, wtype = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 64
path = bench.ensure_gguf(synth.TIERS[tier], wtype, "qrand" if tier == "big" else "float")
g = gguf.load_gguf(path)
dev = model.load_llama_model(g)
prompt = synth.prompt_ids(16, dev.config.vocab_size)
dev.prefill(prompt)
import numpy as np  # noqa: E402
first = int(np.argmax(dev.state.logits))
ids = dev.decode_greedy(first, len(prompt), n)
best = None
for _ in range(5):
    dev.reset(); dev.prefill(prompt)
    t0 = time.perf_counter(); dev.decode_greedy(first, len(prompt), n); dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)
dev.reset(); dev.prefill(prompt)
out = {"tier": tier, "wtype": wtype, "session": os.environ.get("NL_PERSIST_SESSION", "1"), "chained_tokens_per_s": round(n / best, 1)}
out.update(bench.dropin_rates(dev, first, len(prompt), n, ids))
out["persist"] = dev.persist_info()
print(json.dumps(out))
