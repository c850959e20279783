"""Developer tool: prefill time of short prompts (n tokens, one stream) on a tier -- the multi-token step below qgemm2's 128 tokens.
python tools/bench_short_prompt.py mini q4_0 24 48 64 100 127"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from nanollama_amd import gguf, model, synth
tier, wt = sys.argv[1], sys.argv[2]
g = gguf.load_gguf(bench.ensure_gguf(synth.TIERS[tier], wt, "qrand"))
dev = model.load_llama_model(g)
for n in [int(a) for a in sys.argv[3:]]:
    toks = synth.prompt_ids(n, g.meta.vocab_size)
    dev.reset(); dev.prefill(toks); dev.synchronize()
    best = 1e9
    for _ in range(5):
        dev.reset()
        t0 = time.perf_counter(); dev.prefill(toks); best = min(best, time.perf_counter() - t0)
    print(f"{tier} {wt} prompt of {n} tokens: {best * 1e3:.3f} ms (NL_DGEMM={os.environ.get('NL_DGEMM', '1')}, max {os.environ.get('NL_DGEMM_MAX_TOKENS', '64')})")
dev.close()
