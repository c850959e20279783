"""Developer tool (GPU): a few minutes of continuous work on the three multi-launch paths, checking that results stay
identical run after run and that the fused plan never had to be retired (nl_last_error stays empty).
   gpurun -- python3 tools/soak.py [seconds per path]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nanollama_amd import gguf, model, synth
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_modes as b

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0

# 1. nano greedy decode: the same 1000-token continuation over and over (round 5: the persistent launch, positions 16 .. 1015)
g = b.gen("nano", "q8_0")
dev = model.load_llama_model(g)
prompt = synth.prompt_ids(16, g.meta.vocab_size)
ref_ids, runs, toks = None, 0, 0
t0 = time.time()
while time.time() - t0 < budget:
    dev.reset(); dev.prefill(prompt)
    ids = dev.decode_greedy(int(np.argmax(dev.state.logits)), len(prompt), 1000)      # (round 5: through every pass count of the persistent launch)
    ref_ids = ref_ids or ids
    assert ids == ref_ids, f"nano greedy run {runs} differs"
    runs += 1; toks += len(ids)
print(f"nano greedy: {runs} runs, {toks} tokens in {time.time() - t0:.1f} s, identical ids every run; last_error: {dev.last_error()!r}", flush=True)
dev.close()

# 2. goldie x 64 streams: steps from position 0 to 255, repeated (FIN, two-split and general attention paths)
g = b.gen("goldie", "q4_0")
ns = 64
dev = model.load_llama_model(g, max_streams=ns)
rng = np.random.Generator(np.random.PCG64(5))
first = [int(t) for t in rng.integers(3, g.meta.vocab_size, size=ns)]
ref_last, runs, steps = None, 0, 0
t0 = time.time()
while time.time() - t0 < budget:
    for s in range(ns): dev.reset(s)
    ids = list(first)
    for p in range(160):
        ids, _ = dev.forward_batch(list(range(ns)), ids, [p] * ns)
    ref_last = ref_last or list(ids)
    assert list(ids) == ref_last, f"goldie batch run {runs} differs"
    runs += 1; steps += 160
print(f"goldie x 64: {runs} runs, {steps} steps in {time.time() - t0:.1f} s, identical ids every run; last_error: {dev.last_error()!r}", flush=True)
dev.close()

# 3. mini 2047-token prefill, repeated (runs of chunks, images, merge)
g = b.gen("mini", "q4_0")
dev = model.load_llama_model(g)
toks_in = synth.prompt_ids(2047, g.meta.vocab_size)
ref_lg, runs = None, 0
t0 = time.time()
while time.time() - t0 < budget:
    dev.reset(); dev.prefill(toks_in)
    lg = dev.state.logits.copy()
    if ref_lg is None: ref_lg = lg
    assert np.array_equal(lg, ref_lg), f"mini prefill run {runs} differs"
    runs += 1
print(f"mini prefill 2047: {runs} runs in {time.time() - t0:.1f} s, bit-identical logits every run; last_error: {dev.last_error()!r}", flush=True)
dev.close()

# 4. (round 4) big greedy decode across the two-launch layers, the plan switch at position 512 and the split-attention plan,
#    and big at the default sampled settings (top-p by streamed radix selection), repeated
from nanollama_amd.engine import Engine, GenParams
g = b.gen("big", "q4_0")
dev = model.load_llama_model(g)
prompt = synth.prompt_ids(440, g.meta.vocab_size)
ref_ids, ref_s, runs, toks = None, None, 0, 0
t0 = time.time()
while time.time() - t0 < budget:
    dev.reset(); dev.prefill(prompt)
    ids = dev.decode_greedy(int(np.argmax(dev.state.logits)), len(prompt), 160)
    ref_ids = ref_ids or ids
    assert ids == ref_ids, f"big greedy run {runs} differs"
    eng = Engine(dev, eos_id=-1, rep_penalty=1.15, rep_window=64, seed=3, device_sampling=True)
    sid = eng.generate_ids(prompt[:16], GenParams(max_tokens=96, temperature=0.8, top_p=0.9))
    ref_s = ref_s or sid
    assert sid == ref_s, f"big sampled run {runs} differs"
    runs += 1; toks += len(ids) + len(sid)
print(f"big greedy (positions 440..600) + sampled: {runs} runs, {toks} tokens in {time.time() - t0:.1f} s, identical ids every run; "
      f"plan {dev.plan_info()}; last_error: {dev.last_error()!r}", flush=True)
dev.close()

# 5. (round 5) nano per-call Forward on the resident session of the persistent decode (nl_persist.h): runs of nl_forward /
#    nl_forward_argmax calls with a host argmax in between, cut by resets, chained chunks, pauses longer than the idle limit and
#    repeated positions -- the same ids every run, equal to the chained decode's, no give-up
g = b.gen("nano", "q8_0")
dev = model.load_llama_model(g)
prompt = synth.prompt_ids(16, g.meta.vocab_size)
dev.prefill(prompt)
first = int(np.argmax(dev.state.logits))
want = dev.decode_greedy(first, len(prompt), 300)
runs, toks = 0, 0
t0 = time.time()
while time.time() - t0 < budget:
    dev.reset(); dev.prefill(prompt)
    tok, pos, got = first, len(prompt), []
    for i in range(300):
        kind = (i + runs) % 7
        if kind == 3:
            tok = dev.forward_argmax(tok, pos)
        elif kind == 5 and i + 4 <= 300 and i % 50 == 5:
            ids = dev.decode_greedy(tok, pos, 4)          # a chained chunk in the middle of the session
            got += ids[:-1]; tok = ids[-1]; pos += 3
        else:
            dev.forward(tok, pos)
            if i % 97 == 11:
                dev.forward(tok, pos)                      # the same position again: a new session, the same logits
            tok = int(np.argmax(dev.state.logits))
        got.append(tok); pos += 1
        if i % 61 == 60:
            time.sleep(0.004)                              # longer than the idle limit: the launch has left
        if len(got) >= 300:
            break
    assert got[:300] == want[:300], f"session run {runs} differs at {[k for k, (a, c) in enumerate(zip(got, want)) if a != c][:3]}"
    runs += 1; toks += len(got)
info = dev.persist_info()
print(f"nano resident session: {runs} runs, {toks} tokens in {time.time() - t0:.1f} s, ids equal to the chained decode's every run; "
      f"persist {info}; last_error: {dev.last_error()!r}", flush=True)
assert info["ready"] and dev.last_error() == ""
dev.close()
