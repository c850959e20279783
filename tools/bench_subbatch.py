"""Developer tool (round 4): goldie Q4_0 x 64 decode streams (BASELINE config 4) with the batch stepped as 1 / 2 / 4 concurrent
groups (NL_SUB_BATCHES, nl_forward_batch), short and long contexts.   python3 tools/bench_subbatch.py [tier] [wtype]"""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def one(tier, wtype, ns, pos0, steps):
    import numpy as np
    from nanollama_amd import gguf, model, synth
    path = os.path.join(os.environ.get("NL_BENCH_DIR", "/tmp"), f"nl_bench_{tier}_{wtype}_qrand.gguf")
    if not os.path.exists(path):
        synth.generate_gguf(path + ".tmp", synth.TIERS[tier], wtype, mode="qrand")
        os.replace(path + ".tmp", path)
    g = gguf.load_gguf(path)
    dev = model.load_llama_model(g, max_streams=ns)
    rng = np.random.Generator(np.random.PCG64(3))
    ids = [int(t) for t in rng.integers(3, g.meta.vocab_size, size=ns)]
    streams = list(range(ns))
    for p in range(max(0, pos0 - 6), pos0):
        ids, _ = dev.forward_batch(streams, ids, [p] * ns)
    dev.synchronize()
    dev.timer_start()
    t0 = time.perf_counter()
    for k in range(steps):
        ids, _ = dev.forward_batch(streams, ids, [pos0 + k] * ns)
    dt = time.perf_counter() - t0
    ev = dev.timer_stop()
    print(f"{tier} {wtype} x {ns} streams, NL_SUB_BATCHES={os.environ.get('NL_SUB_BATCHES', 'default')}, positions {pos0}..: "
          f"{dt / steps * 1e3:.3f} ms/step wall ({ev / steps:.3f} ms device), {ns * steps / dt:.0f} tok/s aggregate, ids[:4] {ids[:4]}")
    dev.close()

if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[1] == "one":
        one(sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]))
    else:
        tier, wtype = (sys.argv[1], sys.argv[2]) if len(sys.argv) > 2 else ("goldie", "q4_0")
        for ns in (64, 32, 16):
            for pos0, steps in ((8, 32), (1000, 16)):
                for sub in ("1", "2", "4"):
                    subprocess.run([sys.executable, os.path.abspath(__file__), "one", tier, wtype, str(ns), str(pos0), str(steps)],
                                   env=dict(os.environ, NL_SUB_BATCHES=sub))
