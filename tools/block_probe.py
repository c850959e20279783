"""Developer tool: phase stamps (shader clock) of the fused attention block launch of layer 0 (nl_block.h)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nanollama_amd import _lib, gguf, model, synth
tier, wtype = (sys.argv[1], sys.argv[2]) if len(sys.argv) > 2 else ("nano", "q8_0")
path = f"/tmp/probe_{tier}_{wtype}.gguf"
if not os.path.exists(path):
    synth.generate_gguf(path, synth.TIERS[tier], wtype, mode="qrand" if tier in ("big", "goldie") else "float")
dev = model.load_llama_model(gguf.load_gguf(path))
for pos, t in enumerate(synth.prompt_ids(72, synth.TIERS[tier].vocab)):
    dev.forward(t, pos)
L = _lib.lib()
L.nl_debug_stamps.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_longlong)]
names = {15: "loads issued", 16: "dots", 1: "x staged", 2: "barrier1", 3: "quad+ss", 4: "barrier2", 5: "published", 6: "gathered", 7: "barrier3", 8: "kv stored",
         13: "scores+softmax+PV per wavefront+bar", 14: "pass merge (wave 0)", 9: "loop end", 10: "merged+bar", 11: "wo done"}
order = [15, 1, 2, 16, 3, 4, 5, 6, 7, 8, 13, 14, 9, 10, 11]
for rep in range(4):
    buf = (C.c_longlong * 128)()
    _lib.check(dev._h, L.nl_debug_stamps(dev._h, 9, buf))
    st = list(buf)
    prev = st[0]
    out = []
    for k in order:
        out.append(f"{names[k]}:+{st[k] - prev}")
        prev = st[k]
    print(f"total {st[11] - st[0]} cycles | " + " ".join(out))
fn = {1: "loads issued", 2: "x staged+ss", 3: "barrier1", 4: "inv+dots+quad", 5: "barrier2", 6: "silu+publish", 7: "gathered", 8: "barrier3", 9: "down+store"}
for rep in range(3):
    buf = (C.c_longlong * 128)()
    _lib.check(dev._h, L.nl_debug_stamps(dev._h, 10, buf))
    st = list(buf)
    prev, out = st[0], []
    for k in range(1, 10):
        out.append(f"{fn[k]}:+{st[k] - prev}")
        prev = st[k]
    print(f"ffn_block total {st[9] - st[0]} cycles | " + " ".join(out))
