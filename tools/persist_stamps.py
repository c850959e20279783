"""Phase stamps of the persistent decode launch (nanollama_amd/csrc/nl_persist.h, PD_ST): step 2 of a chunk on XCD 1 -- a head
and a worker of the XCD's first layer, the LM-head phase of unit 0 -- on the 100 MHz wall clock, printed as microseconds from
the head's slot start; plus the launch's own stamps (entry, weights resident, first token done, exit).
usage (gpurun): python tools/persist_stamps.py [tokens] [start position]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nanollama_amd import gguf, model, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
pos0 = int(sys.argv[2]) if len(sys.argv) > 2 else 8
path = "/tmp/nano_q8_0.gguf"
if not os.path.exists(path):
    synth.generate_gguf(path, synth.TIERS["nano"], "q8_0", synth.TIER_SEED["nano"], mode="qrand")
g = gguf.load_gguf(path)
dev = model.load_llama_model(g)
print("persist:", dev.persist_info())
prompt = synth.prompt_ids(pos0, synth.TIERS["nano"].vocab, seed=7)
dev.prefill(prompt)
first = int(np.argmax(dev.state.logits))
for rep in range(3):
    dev.decode_greedy(first, pos0, n)
    st = dev.debug_read("pd_dbg", 128).view(np.int64)
    us = lambda a, b: (st[b] - st[a]) / 100.0
    print(f"launch: weights resident after {us(0, 1):.1f} us, first token {us(1, 2):.1f} us, {n} tokens {us(1, 3):.1f} us = {us(1, 3) / n:.2f} us per token")
    h = st[8:15]; w = st[24:36]; l = st[40:44]
    t0 = h[0]
    names_h = ["slot start", "x gathered", "rms + digits", "Q|K|V units (this unit's rows)", "row sums, publish, head gathers q|k|v, RoPE, KV store", "attention passes", "o published"]
    names_w = ["-", "-", "o gathered", "WO units", "x' published", "x' gathered", "rms", "gate|up units", "h published",
               "h gathered", "down units", "x'' published"]
    print("  head  :", ", ".join(f"{nm} {(v - t0) / 100.0:.2f}" for nm, v in zip(names_h, h)))
    print("  worker:", ", ".join(f"{nm} {(v - t0) / 100.0:.2f}" for nm, v in zip(names_w, w)))
    print("  LM    :", ", ".join(f"{nm} {(v - l[0]) / 100.0:.2f}" for nm, v in zip(["start", "x gathered", "units + sums", "argmax published"], l)), f"(LM start {(l[0] - t0) / 100.0:.2f} after the slot start)")
print(dev.last_error())
