"""Developer tool: per-kernel decode time at a long context (mini Q4_0, pos 2040) vs a short one."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench_modes as b
from nanollama_amd import model, synth
tier, wtype = (sys.argv[1], sys.argv[2]) if len(sys.argv) > 2 else ("mini", "q4_0")
g = b.gen(tier, wtype)
dev = model.load_llama_model(g)
toks = synth.prompt_ids(2040, g.meta.vocab_size)
dev.prefill(toks)
for pos in (2040, 72):
    prof = dev.profile_forward(toks[5], pos, iters=20)
    print(pos, {k: round(v[0] / max(v[1], 1) * 1e3, 2) for k, v in prof.items()} if isinstance(prof, dict) else prof)
dev.close()
