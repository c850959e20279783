"""Developer tool (GPU): random prompt lengths / call splits through nl_prefill against token-at-a-time nl_forward on the
same device, for both staging variants of the prompt attention kernel (NL_KV16_MIN_TOKENS) and several GQA shapes.
   gpurun -- python3 tools/fuzz_prefill.py [cases]"""
import os, sys
from dataclasses import replace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nanollama_amd import gguf, model, synth

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.Generator(np.random.PCG64(77))
worst = 0.0
for heads, kv, hd in ((4, 1, 64), (6, 2, 32), (8, 8, 32), (6, 3, 64), (3, 1, 64)):
    shape = replace(synth.TIERS["tiny"], name=f"f{heads}_{kv}_{hd}", dim=heads * hd, n_head=heads, n_kv_head=kv, seq_len=1024, interm=96, n_layer=2)
    path = f"/tmp/nl_fuzz_{heads}_{kv}_{hd}.gguf"
    synth.generate_gguf(path, shape, "q8_0", 5)
    g = gguf.load_gguf(path)
    a, b = model.load_llama_model(g), model.load_llama_model(g)
    for case in range(cases):
        n = int(rng.integers(9, 1000))
        toks = synth.prompt_ids(n, shape.vocab, seed=int(rng.integers(1 << 30)))
        cuts = sorted(set(int(x) for x in rng.integers(1, n, size=int(rng.integers(0, 4)))))
        os.environ["NL_KV16_MIN_TOKENS"] = str([16, 256, 1 << 30][case % 3])
        a.reset(); b.reset()
        lo = 0
        for hi in cuts + [n]:
            a.prefill(toks[lo:hi], pos0=lo, want_logits=(hi == n))
            lo = hi
        for pos, t in enumerate(toks):
            b.forward(t, pos)
        err = float(np.abs(a.state.logits - b.state.logits).max())
        worst = max(worst, err)
        flag = "" if err <= 1e-4 else "   <-- ABOVE 1e-4"
        if flag or case % 8 == 0:
            print(f"heads {heads} kv {kv} hd {hd}: n={n} cuts={cuts} images from {os.environ['NL_KV16_MIN_TOKENS']}: max|prefill - stepwise| = {err:.2e}{flag}", flush=True)
    a.close(); b.close()
print(f"worst {worst:.2e}")
sys.exit(0 if worst <= 1e-4 else 1)
