"""Developer tool: in-kernel phase timing (shader cycles) of the nano GEMV launches."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nanollama_amd import _lib, gguf, model, synth
tier, wtype = (sys.argv[1], sys.argv[2]) if len(sys.argv) > 2 else ("nano", "q8_0")
path = f"/tmp/probe_{tier}_{wtype}.gguf"
if not os.path.exists(path):
    synth.generate_gguf(path, synth.TIERS[tier], wtype, mode="qrand" if tier in ("big", "goldie") else "float")
dev = model.load_llama_model(gguf.load_gguf(path))
for pos, t in enumerate([1, 5, 9, 11]):
    dev.forward(t, pos)
L = _lib.lib()
L.nl_debug_stamps.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_longlong)]
names = ["start", "w-loads issued", "x staged", "barrier1", "dots done", "reduce", "barrier2", "end"]
for kind, nm in ((4, "gate_up"), (5, "down")):
    for rep in range(3):
        buf = (C.c_longlong * 128)()
        _lib.check(dev._h, L.nl_debug_stamps(dev._h, kind, buf))
    print(nm)
    for w in range(4):
        st = [buf[w * 8 + k] for k in range(8)]
        if st[0] == 0:
            continue
        print("  wave", w, " ".join(f"{names[k]}:+{st[k]-st[0]}" for k in range(1, 8) if st[k]))
