"""Developer tool: chained greedy decode rate of a tier at a few positions (launch plans or the persistent launch, whichever the
handle takes).  python tools/bench_decode_tier.py mini q4_0"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nanollama_amd import gguf, model, synth
tier, wt = (sys.argv[1], sys.argv[2]) if len(sys.argv) > 2 else ("mini", "q4_0")
path = os.path.join(os.environ.get("NL_BENCH_DIR", "/tmp"), f"nl_bench_{tier}_{wt}_qrand.gguf")
if not os.path.exists(path):
    synth.generate_gguf(path + ".tmp", synth.TIERS[tier], wt, mode="qrand")
    os.replace(path + ".tmp", path)
g = gguf.load_gguf(path)
dev = model.load_llama_model(g)
toks = synth.prompt_ids(2040, g.meta.vocab_size)
dev.prefill(toks)
out = []
for pos0 in (64, 300, 1000, 2000):
    dev.decode_greedy(5, pos0, 32)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); dev.decode_greedy(5, pos0, 32); best = min(best, time.perf_counter() - t0)
    out.append(f"pos {pos0}: {best / 32 * 1e6:.0f} us ({32 / best:.0f} tok/s)")
print(f"{tier} {wt}: " + "  ".join(out), dev.plan_info(), dev.persist_info())
dev.close()
