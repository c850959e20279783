"""Developer tool: the 7.9B tier's chained greedy decode over positions 440 .. 2039 on the two-launch layers (attention passes
shared by helper blocks, nl_tp.h) against the five-launch plan (NL_FUSED_MAX_POS=0): ids must be identical, a replay too.
   gpurun -- python3 tools/long_context_check.py   (profiles/r05_big_long_context_check.log)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nanollama_amd import gguf, model, synth
import bench

g = gguf.load_gguf(bench.ensure_gguf(synth.TIERS["big"], "q4_0", "qrand"))
prompt = synth.prompt_ids(440, g.meta.vocab_size)
ids = {}
for knob in ("general", "fused"):
    if knob == "general":
        os.environ["NL_FUSED_MAX_POS"] = "0"
    else:
        os.environ.pop("NL_FUSED_MAX_POS", None)
    dev = model.load_llama_model(g)
    runs = []
    for rep in range(2 if knob == "fused" else 1):
        dev.reset(); dev.prefill(prompt)
        first = int(np.argmax(dev.state.logits))
        t0 = time.perf_counter()
        out = dev.decode_greedy(first, len(prompt), 1600)
        dt = time.perf_counter() - t0
        runs.append([first] + out)
        print(knob, rep, f"{1600 / dt:.1f} tok/s over positions 440..2039", dev.plan_info(), repr(dev.last_error()), flush=True)
    ids[knob] = runs
    dev.close()
a, b, c = ids["general"][0], ids["fused"][0], ids["fused"][1]
print("fused replay identical:", b == c)
diff = [i for i, (x, y) in enumerate(zip(a, b)) if x != y]
print("fused vs general: first difference at token", diff[:1], "of", len(a))
