import os, sys, time, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np
from nanollama_amd import gguf, model, synth, _lib
path = "/tmp/nl_bench_nano_q8_0_float.gguf"
if not os.path.exists(path): synth.generate_gguf(path, synth.TIERS["nano"], "q8_0", mode="float")
g = gguf.load_gguf(path)
dev = model.load_llama_model(g)
L = _lib.lib()
toks = synth.prompt_ids(8, g.meta.vocab_size)
dev.prefill(toks)
V = g.meta.vocab_size
buf = np.zeros(V, np.float32)
p = buf.ctypes.data_as(C.POINTER(C.c_float))
nid = C.c_int(0)
def loop(fn, n=300):
    fn(0); fn(1)
    t0 = time.perf_counter()
    for i in range(n): fn(i)
    return (time.perf_counter() - t0) / n * 1e6
h = dev._h
print("nl_forward(logits=NULL)   %.1f us" % loop(lambda i: L.nl_forward(h, 0, 5, 8 + i % 100, None)))
print("nl_forward(logits)        %.1f us" % loop(lambda i: L.nl_forward(h, 0, 5, 8 + i % 100, p)))
print("nl_forward_argmax         %.1f us" % loop(lambda i: L.nl_forward_argmax(h, 0, 5, 8 + i % 100, C.byref(nid))))
print("host np.argmax            %.1f us" % loop(lambda i: int(np.argmax(buf))))
ids = dev.decode_greedy(5, 8, 128)
t0 = time.perf_counter(); ids = dev.decode_greedy(5, 8, 256); dt = time.perf_counter() - t0
print("chained                   %.1f us" % (dt / 256 * 1e6))
dev.close()
