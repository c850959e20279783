"""Developer tool: soak of the decode batch's LM head on resident workgroups (dghead_kernel): goldie Q4_0 x 64 streams, a few
hundred steps, twice -- every step's logits must be bitwise the same in both runs (fixed summation orders; a race between the
LDS-DMA double buffer, the partial tiles and the barriers would show as a difference) -- and once on the split-K launches
(NL_DGEMM_HEAD=0): same ids, logits within 2e-5 of the spread.   python tools/soak_batch_head.py [steps]"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nanollama_amd import gguf, model, synth
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
path = os.path.join(os.environ.get("NL_BENCH_DIR", "/tmp"), "nl_bench_goldie_q4_0_qrand.gguf")
if not os.path.exists(path):
    synth.generate_gguf(path + ".tmp", synth.TIERS["goldie"], "q4_0", mode="qrand")
    os.replace(path + ".tmp", path)
g = gguf.load_gguf(path)
ns = 64
def run(knob):
    os.environ["NL_DGEMM_HEAD"] = knob
    dev = model.load_llama_model(g, max_streams=ns)
    rng = np.random.Generator(np.random.PCG64(3))
    toks = [int(t) for t in rng.integers(3, g.meta.vocab_size, size=ns)]
    hashes, ids_all, last = [], [], None
    for k in range(steps):
        ids, lg = dev.forward_batch(list(range(ns)), toks, [k] * ns, want_logits=True)
        hashes.append(hashlib.sha1(lg.tobytes()).hexdigest())
        ids_all.append([int(i) for i in ids])
        assert np.isfinite(lg).all(), (knob, k)
        toks = ids_all[-1]
        last = lg.copy()
    dev.close()
    return hashes, ids_all, last
a = run("1"); b = run("1"); c = run("0")
same = sum(x == y for x, y in zip(a[0], b[0]))
ids_eq = sum(x == y for x, y in zip(a[1], c[1]))
print(f"goldie q4_0 x {ns} streams, {steps} steps: resident-workgroup head twice: {same} / {steps} steps bitwise equal; "
      f"against the split-K head: ids equal in {ids_eq} / {steps} steps, last step max |diff| / std = "
      f"{float(np.abs(a[2] - c[2]).max()) / float(c[2].std()):.2e}")
sys.exit(0 if same == steps and ids_eq == steps else 1)
