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
names = {1: "chunk landed (own pieces)", 3: "-", 4: "barrier, next chunk requested", 5: "S^T issued", 6: "softmax done", 7: "PV issued"}
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
    lo_half = ids_all = np.nonzero(np.array(cen[:], dtype=np.int64).reshape(8192, 4)[:, 1] > 0)[0]
    for name, m in (("first #CUs workgroups", lo_half < 256), ("the rest", lo_half >= 256)):
        if m.any(): print(f"  {name}: {int(m.sum())}, lifetime us min/median/max {life[m].min():.2f}/{np.median(life[m]):.2f}/{life[m].max():.2f}, start us median {np.median((t_in[m] - base) / 100.0):.2f}, end us median {np.median((t_out[m] - base) / 100.0):.2f}")
    order = np.argsort(-life)[:8]
    print("  longest-lived workgroups (linear id, start us, lifetime us):", [(int(lo_half[i]), round(float((t_in[i] - base) / 100.0), 1), round(float(life[i]), 1)) for i in order])
    ids = np.nonzero(np.array(cen[:], dtype=np.int64).reshape(8192, 4)[:, 1] > 0)[0]
    byc = {}
    for wg, c_ in zip(ids.tolist(), cu.tolist()):
        byc.setdefault(c_, []).append(wg)
    diffs = [b_ - a_ for a_, b_ in (sorted(v_)[:2] for v_ in byc.values() if len(v_) >= 2)]
    print("  linear ids of the workgroups sharing a CU (first 12 CUs):", [sorted(v_) for v_ in list(byc.values())[:12]])
    print("  id distance between the two workgroups of a CU: ", {d: diffs.count(d) for d in sorted(set(diffs))})
# the two workgroups of one CU side by side (ids i and i + 256), shader clock relative to the earlier entry
tl = (C.c_longlong * (512 * 64))()
L.nl_debug_att_timeline.argtypes = [C.POINTER(C.c_longlong)]
if L.nl_debug_att_timeline(tl) == 0:
    t = np.array(tl[:], dtype=np.int64).reshape(512, 64)
    nch = lambda w: sum(1 for c in range(6) if t[w, 10 * c + 7] > 0)
    pairs = [(i, i + 256) for i in range(224) if t[i, 0] and t[i + 256, 0]] or [(i, i) for i in range(256) if t[i, 0]]
    pairs.sort(key=lambda p_: -(nch(p_[0]) + nch(p_[1])))
    for a_, b_ in pairs[:1] + pairs[-1:]:
        z = min(t[a_, 0], t[b_, 0])
        ev = []
        for w in (a_, b_):
            for c in range(6):
                for k, nm in ((1, "chunk landed (own pieces)"), (4, "barrier, next chunk requested"), (5, "S^T issued"), (6, "softmax done"), (7, "PV issued")):
                    if t[w, 10 * c + k]: ev.append((int(t[w, 10 * c + k] - z), w, f"chunk {c} {nm}"))
            ev.append((int(t[w, 8] - z), w, "exit"))
        print(f"CU pair {a_} ({nch(a_)} chunks) / {b_} ({nch(b_)} chunks):")
        for tt, w, nm in sorted(ev): print(f"   {tt:7d}  {'A' if w == a_ else '        B'} {nm}")
dev.close()
