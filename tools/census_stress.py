"""Developer tool: tries to provoke a placement miss of the persistent launch (nl_persist.h census): the sequence of the GPU suite in
which one was seen once -- a handle whose launch gives up at its first poll, closed, then a fresh handle's resident session -- over
and over, with a few variations.  Prints every warning (the note carries the per-XCD counts).   python tools/census_stress.py [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nanollama_amd import gguf, model, synth
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
os.environ["NL_QUIET"] = "1"
shape = synth.ModelShape("pd_stress", 13, 256, 4, 4, 1024, seq_len=192, interm=512)
path = "/tmp/pd_stress.gguf"
synth.generate_gguf(path, shape, "q8_0", 151)
g = gguf.load_gguf(path)
tokens = synth.prompt_ids(150, shape.vocab, seed=21)
misses = 0
for r in range(rounds):
    # (a) a launch that gives up at its first poll
    os.environ["NL_PERSIST_SPIN_LIMIT"] = "0"
    dev = model.load_llama_model(g)
    dev.prefill(tokens[:6])
    dev.decode_greedy(5, 6, 20)
    dev.close()
    del os.environ["NL_PERSIST_SPIN_LIMIT"]
    # (b) a fresh handle: resident session, 150 forced tokens, then a chained decode, then two handles alternating
    dev = model.load_llama_model(g)
    for pos, t in enumerate(tokens):
        dev.forward(t, pos)
    err = dev.last_error()
    if err:
        misses += 1
        print(f"round {r}: session: {err}", flush=True)
    dev.decode_greedy(7, 150, 8)
    other = model.load_llama_model(g)
    for pos in range(12):
        dev.forward(tokens[pos], pos); other.forward(tokens[pos], pos)
    for d, name in ((dev, "first"), (other, "second")):
        if d.last_error():
            misses += 1
            print(f"round {r}: alternating, {name} handle: {d.last_error()}", flush=True)
    info = dev.persist_info()
    dev.close(); other.close()
print(f"{rounds} rounds, {misses} warnings")
