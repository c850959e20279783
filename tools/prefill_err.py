"""developer tool: max |prefill logits - golden| for the tiny golden models (run via gpurun; NL_LIB_PATH selects the build)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nanollama_amd import gguf, model as hip
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
for tag in ["tiny_q8_0", "tiny_qknorm_q8_0", "tiny_conj_q4_0", "tiny_tied_q8_0", "tiny_mha_q4_0", "tiny_q4_0", "tiny_f16"]:
    g = gguf.load_gguf(os.path.join(G, tag + ".gguf"))
    v = np.load(os.path.join(G, tag + ".npz"))
    toks = [int(t) for t in v["prompt"]]
    out = []
    for n in (4, 8, 12):
        dev = hip.load_llama_model(g)
        dev.prefill(toks[:n])
        out.append(float(np.abs(dev.state.logits - v["logits_full"][n - 1]).max()))
        dev.close()
    dev = hip.load_llama_model(g)
    for p, t in enumerate(toks):
        dev.forward(t, p)
    dec = float(np.abs(dev.state.logits - v["logits_full"][len(toks) - 1]).max())
    dev.close()
    print(f"{tag:20s} prefill n=4/8/12: " + " ".join(f"{e:.2e}" for e in out) + f"   decode: {dec:.2e}   logit std {float(v['logits_full'].std()):.2f}")
