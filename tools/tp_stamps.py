"""Developer tool (tools/tp_stamps.sh): phase stamps (wall clock, 10 ns) of tp_attn_kernel / tp_ffn_kernel of one layer of
big Q4_0, rank 0's shard of N in loopback, taken from the last step of a chained greedy decode (cold tags: real waits)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nanollama_amd import gguf, model, synth, _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
path = os.path.join(os.environ.get("NL_BENCH_DIR", "/tmp"), "nl_bench_big_q4_0_qrand.gguf")
if not os.path.exists(path):
    synth.generate_gguf(path + ".tmp", synth.TIERS["big"], "q4_0", mode="qrand")
    os.replace(path + ".tmp", path)
g = gguf.load_gguf(path)
dev = model.load_llama_model(g, tp_rank=0, tp_size=n, p2p_loopback=True) if n > 1 else model.load_llama_model(g)   # 1: the whole layer on one GPU (mode 4)
print("plan", dev.plan_info())
dev.prefill(synth.prompt_ids(8, g.meta.vocab_size))
L = _lib.lib()
out = (C.c_longlong * 128)()
L.nl_debug_tp_stamps.argtypes = [C.POINTER(C.c_longlong)]
names_a = ["entry", "loads issued", "dots done", "barrier", "published q|k|v", "runner: gathered", "runner: passes done", "runner: published out",
           "WO: gathered", "WO: dots + barrier", "pushed + polled + stored"]
names_p = ["entry", "loads issued", "dots done", "barrier", "published"]
names_c = ["entry", "loads issued", "", "", "", "h gathered", "dots + barrier", "pushed + polled + stored"]
for rep in range(3):
    dev.decode_greedy(5, 8, 48)
    dev.synchronize()
    L.nl_debug_tp_stamps(out)
    a = np.array(out[:], dtype=np.int64).reshape(8, 16)
    base = a[0, 0]
    print(f"--- run {rep}: times in us relative to the entry of the attention launch's first runner workgroup")
    for slot, label, names in ((0, "attn runner (cl 0, mem 0)", names_a), (1, "attn WO-only (cl 0, mem G)", names_a), (2, "attn last workgroup", names_a)):
        row = a[slot]
        print(f"  {label}: " + "  ".join(f"{names[i]} {(row[i] - base) / 100.0:.2f}" for i in range(len(names)) if row[i] > 0))
    base_b = a[4, 0]
    print(f"  (feed-forward launch entered {(base_b - base) / 100.0:.2f} us after the attention launch)")
    for slot, label, names in ((4, "ffn producer 0", names_p), (5, "ffn last producer", names_p), (6, "ffn consumer 0", names_c), (7, "ffn last consumer", names_c)):
        row = a[slot]
        print(f"  {label}: " + "  ".join(f"{names[i]} {(row[i] - base_b) / 100.0:.2f}" for i in range(len(names)) if names[i] and row[i] > 0))
    cen = (C.c_longlong * 2048)()
    L.nl_debug_tp_census.argtypes = [C.POINTER(C.c_longlong)]
    if L.nl_debug_tp_census(cen) == 0:
        c = np.array(cen[:], dtype=np.int64).reshape(2, 512, 2)
        for kind, name in ((0, "attention launch"), (1, "feed-forward launch")):
            m = c[kind][:, 0] > 0
            ent, ext = c[kind][m, 0], c[kind][m, 1]
            done = ext > 0
            if not m.any() or not done.any():      # (a launch that keeps no census: wide_ffn_kernel on one GPU)
                print(f"  census {name}: {int(m.sum())} blocks entered, {int(done.sum())} recorded an exit")
                continue
            print(f"  census {name}: {int(m.sum())} blocks entered over {(ent.max() - ent.min()) / 100.0:.2f} us; first entry {(ent.min() - base) / 100.0:.2f}, "
                  f"exits (blocks that ran a role: {int(done.sum())}) first {(ext[done].min() - base) / 100.0:.2f} / median {(np.median(ext[done]) - base) / 100.0:.2f} / last {(ext[done].max() - base) / 100.0:.2f} us")
dev.close()
