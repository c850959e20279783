"""Developer tool (tools/dg_stamps.sh): stamps of the last launch of one kind of dgemm launch in a goldie Q4_0 x 64 streams step."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nanollama_amd import gguf, model, synth, _lib
path = os.path.join(os.environ.get("NL_BENCH_DIR", "/tmp"), "nl_bench_goldie_q4_0_qrand.gguf")
if not os.path.exists(path):
    synth.generate_gguf(path + ".tmp", synth.TIERS["goldie"], "q4_0", mode="qrand")
    os.replace(path + ".tmp", path)
g = gguf.load_gguf(path)
ns = 64
dev = model.load_llama_model(g, max_streams=ns)
toks = [5 + s for s in range(ns)]
L = _lib.lib()
st, cen = (C.c_longlong * 64)(), (C.c_longlong * 4096)()
L.nl_debug_dg_stamps.argtypes = [C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
for k in range(12):
    ids, _ = dev.forward_batch(list(range(ns)), toks, [k] * ns)
    toks = [int(i) for i in ids]
    if k < 8:
        continue
    assert L.nl_debug_dg_stamps(st, cen) == 0
    s = np.array(st[:], dtype=np.int64)
    c = np.array(cen[:], dtype=np.int64).reshape(2048, 2)
    m = (c[:, 0] > 0) & (c[:, 1] > c[:, 0])
    ent, ext = c[m, 0], c[m, 1]
    # (stale entries of other launch shapes: keep the workgroups whose entry lies within 50 us of the latest one)
    recent = ent > ent.max() - 5000
    ent, ext = ent[recent], ext[recent]
    print(f"step {k}: workgroup 9 wavefront 0 cycles since entry: " + " ".join(str(int(v - s[0])) for v in s[1:40] if v > 0) + f" | barrier {int(s[40] - s[0])}")
    print(f"   census: {int(recent.sum())} workgroups; entries spread {(ent.max() - ent.min()) / 100:.2f} us; exits first +{(ext.min() - ent.min()) / 100:.2f} / last +{(ext.max() - ent.min()) / 100:.2f} us; "
          f"time in kernel per workgroup: median {np.median(ext - ent) / 100:.2f} us")
# the launch log of the last step: entry / exit of workgroup 0 of every dgemm / dghead launch, in launch order
lg, n = (C.c_longlong * 16384)(), C.c_uint(0)
L.nl_debug_dg_log.argtypes = [C.POINTER(C.c_longlong), C.POINTER(C.c_uint)]
if L.nl_debug_dg_log(lg, C.byref(n)) == 0:
    a = np.array(lg[:], dtype=np.int64).reshape(4096, 4)
    per = 4 * 28 + 1                                     # launches with a log entry per step
    last = [(n.value - 1 - i) & 4095 for i in range(per)][::-1]
    ent, ext, end = a[last, 0], a[last, 1], a[last, 2]
    names = ["Q|K|V", "WO", "gate|up", "down"]
    print("last step, workgroup 0 of each launch (us): in-kernel, then the gap to the next launch's entry")
    for k in range(4):
        ins = (ext[k:per - 1:4] - ent[k:per - 1:4]) / 100.0
        gap = (ent[k + 1:per:4] - ext[k:per - 1:4]) / 100.0
        epi = (end[k:per - 1:4] - ext[k:per - 1:4]) / 100.0
        print(f"   {names[k]:8s} entry -> barrier median {np.median(ins):.2f} (min {ins.min():.2f} max {ins.max():.2f});  wavefront 0's epilogue {np.median(epi):.2f};  "
              f"barrier -> next launch's entry median {np.median(gap):.2f} (min {gap.min():.2f} max {gap.max():.2f})")
    print(f"   LM head  in-kernel {(ext[-1] - ent[-1]) / 100.0:.2f};  first Q|K|V entry -> head exit {(ext[-1] - ent[0]) / 100.0:.1f} us")
dev.close()
