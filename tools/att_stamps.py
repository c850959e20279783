"""Developer tool: phase stamps of one workgroup of attn_tile16_kernel (library built with -DNL_ATT_STAMPS=<tile>).
   bash tools/att_stamps.sh"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nanollama_amd import gguf, model, synth, _lib
path = "/tmp/nl_modes_mini_q4_0.gguf"
if not os.path.exists(path):
    synth.generate_gguf(path, synth.TIERS["mini"], "q4_0", mode="qrand")
g = gguf.load_gguf(path)
dev = model.load_llama_model(g)
toks = synth.prompt_ids(2047, g.meta.vocab_size)
dev.prefill(toks[:64]); dev.prefill(toks); dev.synchronize()
L = _lib.lib()
out = (C.c_longlong * 64)()
L.nl_debug_att_stamps.argtypes = [C.POINTER(C.c_longlong)]
print("rc", L.nl_debug_att_stamps(out))
t0 = out[0]
print(f"entry +0; exit +{out[8] - t0}")
names = {1: "chunk requested (DMA issued / loads issued)", 3: "chunk in LDS (own part)", 4: "after barrier", 5: "S^T issued", 6: "softmax done", 7: "PV issued"}
prev = t0
for c in range(6):
    if out[10 * c + 4] == 0 and c > 0: break
    for k in (1, 3, 4, 5, 6, 7):
        t = out[10 * c + k]
        if t: print(f"  chunk {c} {names[k]:44s} +{t - t0:7d}  (+{t - prev})"); prev = t
# census of the last launch: how many workgroups were resident per CU over time, how long each lived
cen = (C.c_longlong * (4 * 8192))()
L.nl_debug_att_census.argtypes = [C.POINTER(C.c_longlong)]
if L.nl_debug_att_census(cen) == 0:
    a = np.array(cen[:], dtype=np.int64).reshape(8192, 4)
    a = a[a[:, 1] > 0]
    t_in, t_out, hw, xcc = a[:, 0], a[:, 1], a[:, 2], a[:, 3] & 0xff
    lds_alloc = a[:, 3] >> 8
    print('  HW_REG_LDS_ALLOC values seen:', {hex(int(v)): int((lds_alloc == v).sum()) for v in np.unique(lds_alloc)})
    base = t_in.min()
    span = (t_out.max() - base) / 100.0
    life = (t_out - t_in) / 100.0
    cu = (xcc << 16) | (((hw >> 13) & 7) << 8) | ((hw >> 8) & 15)          # (XCC, SE, CU)
    print(f"census: {len(a)} workgroups on {len(set(cu.tolist()))} CUs, launch span {span:.1f} us; lifetime us min/median/mean/max "
          f"{life.min():.2f}/{np.median(life):.2f}/{life.mean():.2f}/{life.max():.2f}; sum of lifetimes / span = {life.sum() / span:.0f} resident on average")
    # quartiles of the launch: starts per quarter, mean lifetime per quarter
    for q in range(4):
        lo, hi = base + span * 100 * q / 4, base + span * 100 * (q + 1) / 4
        m = (t_in >= lo) & (t_in < hi)
        print(f"  quarter {q}: {int(m.sum())} started, mean lifetime {life[m].mean() if m.any() else 0:.2f} us")
    per = {}
    for c in cu.tolist():
        per[c] = per.get(c, 0) + 1
    v = np.array(list(per.values()))
    print(f"  workgroups per CU: min {v.min()} median {int(np.median(v))} max {v.max()}")
    # peak concurrency on one CU
    peak = 0
    for c in list(per)[:64]:
        m = cu == c
        ev = sorted([(t, 1) for t in t_in[m]] + [(t, -1) for t in t_out[m]])
        k = 0
        for _, d in ev:
            k += d; peak = max(peak, k)
    print(f"  peak workgroups resident on one CU (first 64 CUs): {peak}")
dev.close()
