"""Developer tool: phase stamps of one workgroup of attn_tile16_kernel (library built with -DNL_ATT_STAMPS=<tile>).
   bash tools/att_stamps.sh"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nanollama_amd import gguf, model, synth, _lib
path = "/tmp/nl_modes_mini_q4_0.gguf"
if not os.path.exists(path):
    synth.generate_gguf(path, synth.TIERS["mini"], "q4_0", mode="qrand")
g = gguf.load_gguf(path)
dev = model.load_llama_model(g)
toks = synth.prompt_ids(2047, g.meta.vocab_size)
dev.prefill(toks[:64]); dev.prefill(toks); dev.synchronize()
L = _lib.lib()
out = (C.c_longlong * 64)()
L.nl_debug_att_stamps.argtypes = [C.POINTER(C.c_longlong)]
print("rc", L.nl_debug_att_stamps(out))
names = ["entry", "loads issued", "staged (LDS stores done)", "before barrier", "after barrier", "S^T done", "softmax done", "PV done", "stored"]
t0 = out[0]
for i, nm in enumerate(names):
    print(f"{nm:28s} +{out[i] - t0}")
dev.close()
