"""Developer tool: tokens/s of the reference's DEFAULT generation settings (temp 0.8, top-p 0.9, repetition penalty
1.15 over 64 tokens -- go/main.go:29-34) with the loop on the device vs the per-token host loop."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nanollama_amd import gguf, model, synth
from nanollama_amd.engine import Engine, GenParams

tier, wtype = (sys.argv[1], sys.argv[2]) if len(sys.argv) > 2 else ("nano", "q8_0")
path = f"/tmp/nl_samp_{tier}_{wtype}.gguf"
if not os.path.exists(path):
    synth.generate_gguf(path, synth.TIERS[tier], wtype, mode="float" if tier == "nano" else "qrand")
g = gguf.load_gguf(path)
dev = model.load_llama_model(g)
prompt = synth.prompt_ids(8, g.meta.vocab_size)
for label, kw in (("device loop, top-p 0.9", dict(device_sampling=True)), ("host loop,   top-p 0.9", dict(device_sampling=False))):
    eng = Engine(dev, eos_id=-1, rep_penalty=1.15, rep_window=64, seed=1, **kw)
    eng.generate_ids(prompt, GenParams(max_tokens=16, temperature=0.8, top_p=0.9))
    n = 256 if kw["device_sampling"] else 64
    t0 = time.perf_counter(); ids = eng.generate_ids(prompt, GenParams(max_tokens=n, temperature=0.8, top_p=0.9)); dt = time.perf_counter() - t0
    print(f"{tier} {wtype} {label}: {len(ids) / dt:8.1f} tok/s  ({dt / len(ids) * 1e3:.3f} ms/token)")
eng = Engine(dev, eos_id=-1, rep_penalty=1.15, rep_window=64, seed=1)
t0 = time.perf_counter(); ids = eng.generate_ids(prompt, GenParams(max_tokens=256, temperature=0.8, top_p=1.0, top_k=50)); dt = time.perf_counter() - t0
print(f"{tier} {wtype} device loop, top-k 50 : {len(ids) / dt:8.1f} tok/s")
eng = Engine(dev, eos_id=-1, rep_penalty=1.0)
t0 = time.perf_counter(); ids = eng.generate_ids(prompt, GenParams(max_tokens=256, temperature=0.0)); dt = time.perf_counter() - t0
print(f"{tier} {wtype} greedy chained        : {len(ids) / dt:8.1f} tok/s")
dev.close()
