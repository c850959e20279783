"""GPU tests of the weight-stationary persistent greedy decode (nanollama_amd/csrc/nl_persist.h): one launch decodes a chunk of
tokens with every layer resident on one XCD.  Reference lines: go/main.go:173-219 (the greedy loop), go/model.go:490-620
(Forward), go/main.go:400-408 (argmax).  The oracle's ids must be reproduced one for one and its logits within the engine's
stated tolerance; the launch plans of the same handle (NL_PERSIST=0) are the second witness."""
import functools
import os
import sys

import numpy as np
import pytest

from nanollama_amd import gguf, synth

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

class _PlacementMiss(Exception):
    """the persistent launch found its workgroups placed otherwise than 32 per XCD (observed, never assumed: nl_persist.h) -- the
    chunk was redone on the launch plans and the results are right, but the run did not test the persistent path"""


def _no_warning(dev):
    err = dev.last_error()
    if "not placed 32 per XCD" in err:
        raise _PlacementMiss(err)
    return err == ""


def _retry_placement(fn):
    """A placement miss is a property of the moment on the device, not of the code under test: one retry with fresh handles (the
    note is printed); a second miss in a row fails the test."""
    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        try:
            return fn(*args, **kwargs)
        except _PlacementMiss as exc:
            print(f"\n[retry] {fn.__name__}: {exc}")
        return fn(*args, **kwargs)
    return wrapper


LOGIT_TOL = 1e-4


@pytest.fixture(scope="module")
def hip():
    from nanollama_amd import _lib, model
    if _lib.lib().nl_device_count() < 1:
        pytest.fail("no HIP device visible: the GPU tests must run on the MI355X box")
    return model


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


def _oracle_run(orc, g, prompt, n):
    """teacher-forced prompt, then n greedy steps: ids and the logits of every decode step"""
    ref = orc.OracleModel(g)
    orc.set_threads(min(16, os.cpu_count() or 1))
    lg = None
    for pos, t in enumerate(prompt):
        lg = ref.forward(t, pos)
    ids, logits = [], []
    tok = int(orc.argmax(lg))
    first = tok
    for i in range(n):
        lg = ref.forward(tok, len(prompt) + i)
        logits.append(lg.copy())
        tok = int(orc.argmax(lg))
        ids.append(tok)
    orc.set_threads(1)
    ref.close()
    return first, ids, logits


@pytest.mark.parametrize("layers", [5, 13, 16, 1])
@_retry_placement
def test_persistent_decode_matches_oracle_small_shape(hip, orc, tmp_path, monkeypatch, layers):
    # D 256 / 4 heads / I 512: the small instantiation.  5 layers: XCDs 0-4 hold one layer, 5-7 only LM-head rows; 13: nano's
    # split (two layers on XCDs 0-4); 16: two everywhere; 1: a single layer.  The handle's launch plans are the second witness.
    shape = synth.ModelShape(f"pd_small_{layers}", layers, 256, 4, 4, 2048, seq_len=256, interm=512)
    p = tmp_path / "m.gguf"
    synth.generate_gguf(str(p), shape, "q8_0", 101 + layers)
    g = gguf.load_gguf(str(p))
    prompt = synth.prompt_ids(9, shape.vocab, seed=3)
    first, want, want_logits = _oracle_run(orc, g, prompt, 60)
    dev = hip.load_llama_model(g)
    info = dev.persist_info()
    assert info["ready"] and info["max_pos"] == 256, info
    dev.prefill(prompt)
    assert int(np.argmax(dev.state.logits)) == first
    got = dev.decode_greedy(first, len(prompt), 60)
    assert _no_warning(dev), dev.last_error()
    info = dev.persist_info()
    assert info["ready"] and info["launches"] == 1 and info["tokens"] == 60, info
    assert got == want, (got[:12], want[:12])
    lg = dev.debug_read("logits", shape.vocab)
    d = float(np.abs(lg - want_logits[-1]).max()) / max(1.0, float(want_logits[-1].std()))
    print(f"\npersistent decode, {layers} layers: 60 ids equal; last-step max|gpu-oracle| = {d:.2e}")
    assert d <= LOGIT_TOL
    # the K / V rows it wrote serve the launch plans: continue per call from where it stopped, same ids as the oracle's next ones
    monkeypatch.setenv("NL_PERSIST", "0")
    plain = hip.load_llama_model(g)
    assert not plain.persist_info()["ready"]
    plain.prefill(prompt)
    assert plain.decode_greedy(first, len(prompt), 60) == want
    for which in ("k_cache", "v_cache"):
        n = shape.n_layer * shape.n_kv_head * shape.seq_len * 64
        a = dev.debug_read(which, n).reshape(-1, shape.seq_len, 64)[:, :69]
        b = plain.debug_read(which, n).reshape(-1, shape.seq_len, 64)[:, :69]
        assert np.abs(a - b).max() <= 2e-5, which
    # replay from a reset: bit-identical ids and logits (fixed summation orders)
    dev.reset()
    dev.prefill(prompt)
    assert dev.decode_greedy(first, len(prompt), 60) == got
    assert dev.debug_read("logits", shape.vocab).tobytes() == lg.tobytes()
    dev.close(); plain.close()


@_retry_placement
def test_persistent_decode_per_step_logits_and_chunking(hip, orc, tmp_path):
    # one-token chunks (every step's logits against the oracle's), then ragged chunks: the same ids whatever the chunking
    shape = synth.ModelShape("pd_steps", 13, 256, 4, 4, 1024, seq_len=192, interm=512)
    p = tmp_path / "m.gguf"
    synth.generate_gguf(str(p), shape, "q8_0", 131)
    g = gguf.load_gguf(str(p))
    prompt = synth.prompt_ids(5, shape.vocab, seed=9)
    first, want, want_logits = _oracle_run(orc, g, prompt, 150)          # positions 5 .. 154: a second attention pass from 128 on
    dev = hip.load_llama_model(g)
    assert dev.persist_info()["ready"]
    dev.prefill(prompt)
    tok, worst = first, 0.0
    for i in range(150):
        (tok,) = dev.decode_greedy(tok, len(prompt) + i, 1)
        assert tok == want[i], i
        lg = dev.debug_read("logits", shape.vocab)
        worst = max(worst, float(np.abs(lg - want_logits[i]).max()) / max(1.0, float(want_logits[i].std())))
    print(f"\npersistent decode, per-step logits over 150 positions: max|gpu-oracle| = {worst:.2e}")
    assert worst <= LOGIT_TOL
    dev.reset()
    dev.prefill(prompt)
    got, tok, pos = [], first, len(prompt)
    for n in (1, 2, 7, 16, 33, 64, 27):
        ids = dev.decode_greedy(tok, pos, n)
        got += ids; tok = ids[-1]; pos += n
    assert got == want
    assert _no_warning(dev) and dev.persist_info()["launches"] == 157
    dev.close()


@_retry_placement
def test_persistent_decode_hands_over_to_the_launch_plans_at_its_position_limit(hip, orc, tmp_path, monkeypatch):
    # a chunk that crosses the limit: the tokens below it in one persistent launch, the rest on the launch plans, one call
    shape = synth.ModelShape("pd_limit", 13, 256, 4, 4, 1024, seq_len=160, interm=512)
    p = tmp_path / "m.gguf"
    synth.generate_gguf(str(p), shape, "q8_0", 137)
    g = gguf.load_gguf(str(p))
    prompt = synth.prompt_ids(8, shape.vocab, seed=11)
    first, want, _ = _oracle_run(orc, g, prompt, 100)
    monkeypatch.setenv("NL_PERSIST_MAX_POS", "48")
    dev = hip.load_llama_model(g)
    assert dev.persist_info() == {"ready": True, "max_pos": 48, "launches": 0, "tokens": 0}
    dev.prefill(prompt)
    assert dev.decode_greedy(first, len(prompt), 100) == want
    info = dev.persist_info()
    assert info["launches"] == 1 and info["tokens"] == 40, info
    assert _no_warning(dev)
    dev.close()


@_retry_placement
def test_persistent_decode_give_up_falls_back_to_the_launch_plans(hip, orc, tmp_path, monkeypatch):
    # every hand-off poll gives up at once (spin limit 0): the call still returns the oracle's ids -- redone on the launch plans --
    # with a note in nl_last_error, and the handle keeps the plans from then on
    shape = synth.ModelShape("pd_fallback", 13, 256, 4, 4, 1024, seq_len=96, interm=512)
    p = tmp_path / "m.gguf"
    synth.generate_gguf(str(p), shape, "q8_0", 139)
    g = gguf.load_gguf(str(p))
    prompt = synth.prompt_ids(6, shape.vocab, seed=13)
    first, want, _ = _oracle_run(orc, g, prompt, 40)
    monkeypatch.setenv("NL_PERSIST_SPIN_LIMIT", "0")
    monkeypatch.setenv("NL_QUIET", "1")
    dev = hip.load_llama_model(g)
    assert dev.persist_info()["ready"]
    dev.prefill(prompt)
    assert dev.decode_greedy(first, len(prompt), 20) == want[:20]
    assert "persistent decode launch gave up" in dev.last_error()
    assert not dev.persist_info()["ready"]
    assert dev.decode_greedy(want[19], len(prompt) + 20, 20) == want[20:]
    dev.close()


def _teacher_forced(orc, g, tokens):
    """the oracle's logits of Forward(tokens[i], i)"""
    ref = orc.OracleModel(g)
    orc.set_threads(min(16, os.cpu_count() or 1))
    out = [ref.forward(t, pos).copy() for pos, t in enumerate(tokens)]
    orc.set_threads(1)
    ref.close()
    return out


def _rel(a, b):
    return float(np.abs(a - b).max()) / max(1.0, float(b.std()))


@_retry_placement
def test_resident_session_serves_per_call_forward(hip, orc, tmp_path, monkeypatch):
    # go/main.go:173-219 calls Forward once per token with a token the HOST chose: here arbitrary (not the argmax) tokens.  One
    # resident launch serves the whole run of nl_forward calls; every step's logits against the oracle's.
    shape = synth.ModelShape("pd_session", 13, 256, 4, 4, 1024, seq_len=192, interm=512)
    p = tmp_path / "m.gguf"
    synth.generate_gguf(str(p), shape, "q8_0", 151)
    g = gguf.load_gguf(str(p))
    tokens = synth.prompt_ids(150, shape.vocab, seed=21)
    want = _teacher_forced(orc, g, tokens)
    dev = hip.load_llama_model(g)
    assert dev.persist_info()["ready"]
    worst = 0.0
    for pos, t in enumerate(tokens):
        dev.forward(t, pos)
        worst = max(worst, _rel(dev.state.logits, want[pos]))
    info = dev.persist_info()
    print(f"\nresident session, 150 forced tokens: max|gpu-oracle| = {worst:.2e}; launches {info['launches']}")
    assert worst <= LOGIT_TOL and _no_warning(dev)
    assert info["launches"] == 1 and info["tokens"] == 150, info
    keep = dev.state.logits.copy()
    # the launch has left when another entry point needs the stream; what it wrote is what the launch plans read
    assert _rel(dev.debug_read("logits", shape.vocab), want[-1]) <= LOGIT_TOL
    got = dev.decode_greedy(int(np.argmax(keep)), 150, 8)
    monkeypatch.setenv("NL_PERSIST", "0")
    plain = hip.load_llama_model(g)
    plain.prefill(tokens)
    assert plain.decode_greedy(int(np.argmax(plain.state.logits)), 150, 8) == got
    monkeypatch.delenv("NL_PERSIST")
    # nl_forward_argmax on the same kind of session; then a call that needs logits starts a session that stores them
    dev.reset()
    for pos, t in enumerate(tokens[:40]):
        assert dev.forward_argmax(t, pos) == int(np.argmax(want[pos])), pos
    base = dev.persist_info()["launches"]
    dev.forward(tokens[40], 40)
    assert _rel(dev.state.logits, want[40]) <= LOGIT_TOL
    assert dev.persist_info()["launches"] == base + 1
    # ... which also serves argmax calls
    assert dev.forward_argmax(tokens[41], 41) == int(np.argmax(want[41]))
    assert dev.persist_info()["launches"] == base + 1
    # a repeated position, a reset, a jump over unwritten rows: each starts over and agrees with the launch plans
    dev.forward(tokens[41], 41)
    assert _rel(dev.state.logits, want[41]) <= LOGIT_TOL
    dev.reset(); plain.reset()
    for pos in (0, 1, 2, 9, 10):
        dev.forward(tokens[pos], pos); plain.forward(tokens[pos], pos)
        assert _rel(dev.state.logits, plain.state.logits) <= 2e-5, pos
    assert _no_warning(dev)
    dev.close(); plain.close()


@_retry_placement
def test_resident_session_idles_out_and_is_restarted(hip, orc, tmp_path, monkeypatch):
    # a caller slower than the idle limit: the launch has left by the time the next token arrives; the call starts another
    import time
    shape = synth.ModelShape("pd_idle", 5, 256, 4, 4, 512, seq_len=64, interm=512)
    p = tmp_path / "m.gguf"
    synth.generate_gguf(str(p), shape, "q8_0", 157)
    g = gguf.load_gguf(str(p))
    tokens = synth.prompt_ids(12, shape.vocab, seed=23)
    want = _teacher_forced(orc, g, tokens)
    monkeypatch.setenv("NL_PERSIST_IDLE_US", "300")
    dev = hip.load_llama_model(g)
    for pos, t in enumerate(tokens):
        dev.forward(t, pos)
        assert _rel(dev.state.logits, want[pos]) <= LOGIT_TOL, pos
        time.sleep(0.02)
    info = dev.persist_info()
    assert info["launches"] == 12 and info["tokens"] == 12 and info["ready"], info
    assert _no_warning(dev)
    dev.close()


@_retry_placement
def test_resident_session_position_limit_and_give_up(hip, orc, tmp_path, monkeypatch):
    shape = synth.ModelShape("pd_slimit", 13, 256, 4, 4, 1024, seq_len=96, interm=512)
    p = tmp_path / "m.gguf"
    synth.generate_gguf(str(p), shape, "q8_0", 163)
    g = gguf.load_gguf(str(p))
    tokens = synth.prompt_ids(60, shape.vocab, seed=27)
    want = _teacher_forced(orc, g, tokens)
    # positions below the limit on the session (it ends by itself on its last step), the rest on the launch plans
    monkeypatch.setenv("NL_PERSIST_MAX_POS", "32")
    dev = hip.load_llama_model(g)
    for pos, t in enumerate(tokens):
        dev.forward(t, pos)
        assert _rel(dev.state.logits, want[pos]) <= LOGIT_TOL, pos
    info = dev.persist_info()
    assert info["launches"] == 1 and info["tokens"] == 32 and _no_warning(dev), info
    dev.close()
    # every poll gives up at once: the call is redone on the launch plans, which the handle keeps
    monkeypatch.delenv("NL_PERSIST_MAX_POS")
    monkeypatch.setenv("NL_PERSIST_SPIN_LIMIT", "0")
    monkeypatch.setenv("NL_QUIET", "1")
    dev = hip.load_llama_model(g)
    for pos, t in enumerate(tokens[:6]):
        dev.forward(t, pos)
        assert _rel(dev.state.logits, want[pos]) <= LOGIT_TOL, pos
    assert "persistent decode launch gave up" in dev.last_error()
    assert not dev.persist_info()["ready"]
    dev.close()
    # sessions off: one launch of one step per call
    monkeypatch.delenv("NL_PERSIST_SPIN_LIMIT")
    monkeypatch.setenv("NL_PERSIST_SESSION", "0")
    dev = hip.load_llama_model(g)
    for pos, t in enumerate(tokens[:6]):
        dev.forward(t, pos)
        assert _rel(dev.state.logits, want[pos]) <= LOGIT_TOL, pos
    assert dev.persist_info()["launches"] == 6
    dev.close()


@_retry_placement
def test_persistent_decode_long_context_shares_the_attention_passes(hip, orc, tmp_path):
    # from position 128 on a head's 128-position passes are shared by up to three units of its XCD (the owner + two helpers that
    # are not heads in that layer slot; go/model.go:557-587); the owner merges their records.  Teacher-forced against the oracle
    # across every pass count 1 .. 8, at the pass edges; then a chained run up to the position limit (1024) and over it.
    shape = synth.ModelShape("pd_long", 13, 256, 4, 4, 1024, seq_len=1100, interm=512)
    p = tmp_path / "m.gguf"
    synth.generate_gguf(str(p), shape, "q8_0", 167)
    g = gguf.load_gguf(str(p))
    tokens = synth.prompt_ids(1040, shape.vocab, seed=29)
    check = {0, 5, 127, 128, 129, 200, 255, 256, 257, 383, 384, 400, 511, 512, 513, 640, 767, 768, 800, 895, 896, 1000, 1022, 1023, 1024, 1030}
    ref = orc.OracleModel(g)
    orc.set_threads(min(16, os.cpu_count() or 1))
    want = {}
    for pos, t in enumerate(tokens):
        lg = ref.forward(t, pos)
        if pos in check:
            want[pos] = lg.copy()
    dev = hip.load_llama_model(g)
    assert dev.persist_info()["max_pos"] == 1024
    worst = 0.0
    for pos, t in enumerate(tokens):
        dev.forward(t, pos)
        if pos in check:
            worst = max(worst, _rel(dev.state.logits, want[pos]))
    info = dev.persist_info()
    print(f"\npersistent decode, positions 0 .. 1039 teacher-forced: max|gpu-oracle| = {worst:.2e} at {len(check)} positions; {info}")
    assert worst <= LOGIT_TOL and _no_warning(dev)
    assert info["tokens"] == 1024 and info["launches"] <= 3, info      # (one resident launch unless the host paused longer than the idle limit)
    # chained: from position 900 over the limit, the oracle's greedy ids
    first = int(orc.argmax(ref.forward(tokens[900], 900)))       # (the oracle's cache holds positions 0 .. 1039: position 900 again)
    ref.close()
    ref = orc.OracleModel(g)
    for pos, t in enumerate(tokens[:901]):
        lg = ref.forward(t, pos)
    ids, tok = [], int(orc.argmax(lg))
    for i in range(150):
        lg = ref.forward(tok, 901 + i)
        tok = int(orc.argmax(lg)); ids.append(tok)
    orc.set_threads(1)
    ref.close()
    dev.reset()
    dev.prefill(tokens[:901])
    f0 = int(np.argmax(dev.state.logits))
    assert f0 == first
    assert dev.decode_greedy(f0, 901, 150) == ids
    assert _no_warning(dev)
    dev.close()


@_retry_placement
def test_nano_true_shape_resident_session_and_chained_decode_across_every_pass_count(hip, orc, tmp_path, monkeypatch):
    # BASELINE configs[1] at its OWN geometry (D 576 / 9 heads / I 1536, 13 layers, Q8_0): the helper assignment of the shared
    # attention passes depends on the head count (nl_persist.h: `apart = 1 + j / H`), so the D 256 / 4-head test above does not
    # cover it.  (a) 1040 teacher-forced tokens through nl_forward -- the resident session, what the patched Go loop of
    # go/main.go:173-219 calls -- against the oracle at the edges of every pass count 1 .. 8 (go/model.go:557-587) and over the
    # 1024-position limit; (b) chained nl_decode_greedy from position 900 over the limit = the oracle's ids; (c) the launch
    # plans of the same file (NL_PERSIST=0) as the second witness of the chained ids.
    shape = synth.TIERS["nano"]
    p = tmp_path / "nano.gguf"
    synth.generate_gguf(str(p), shape, "q8_0", mode="float")
    g = gguf.load_gguf(str(p))
    tokens = synth.prompt_ids(1040, shape.vocab, seed=31)
    check = {0, 127, 128, 255, 256, 383, 384, 511, 512, 640, 767, 768, 895, 896, 1000, 1023, 1024, 1030}
    ref = orc.OracleModel(g)
    orc.set_threads(min(32, os.cpu_count() or 1))
    want = {}
    at900 = None
    for pos, t in enumerate(tokens):
        lg = ref.forward(t, pos)
        if pos in check:
            want[pos] = lg.copy()
        if pos == 900:
            at900 = lg.copy()
    ref.close()
    dev = hip.load_llama_model(g)
    info = dev.persist_info()
    assert info["ready"] and info["max_pos"] == 1024, info
    worst, worst_pos = 0.0, -1
    for pos, t in enumerate(tokens):
        dev.forward(t, pos)
        if pos in check:
            d = _rel(dev.state.logits, want[pos])
            if d > worst:
                worst, worst_pos = d, pos
    info = dev.persist_info()
    print(f"\nnano true shape, resident session, positions 0 .. 1039 teacher-forced: max|gpu-oracle| = {worst:.2e} (position "
          f"{worst_pos}) over {len(check)} positions; {info}")
    assert worst <= LOGIT_TOL and _no_warning(dev)
    assert info["tokens"] == 1024 and info["launches"] <= 4, info      # (one resident launch unless the host paused past the idle limit)
    # (b) chained from position 900 over the limit
    ref = orc.OracleModel(g)
    for pos, t in enumerate(tokens[:901]):
        lg = ref.forward(t, pos)
    assert np.array_equal(lg, at900)
    top2 = []
    ids, tok = [], int(orc.argmax(lg))
    first = tok
    for i in range(150):
        lg = ref.forward(tok, 901 + i)
        tok = int(orc.argmax(lg)); ids.append(tok)
        s2 = np.partition(lg, -2)[-2:]
        top2.append(float(s2[1] - s2[0]))
    orc.set_threads(1)
    ref.close()
    dev.reset()
    dev.prefill(tokens[:901])
    f0 = int(np.argmax(dev.state.logits))
    assert f0 == first
    base = dev.persist_info()["tokens"]
    got = dev.decode_greedy(f0, 901, 150)
    assert _no_warning(dev)
    assert dev.persist_info()["tokens"] - base == 1024 - 901, dev.persist_info()      # positions 901 .. 1023 in the persistent launch
    if got != ids:
        k = next(i for i in range(150) if got[i] != ids[i])
        assert top2[k] < 5e-5, (k, got[k], ids[k], top2[k])     # only a tie within summation-order noise may differ
    # (c) the launch plans of the same file
    monkeypatch.setenv("NL_PERSIST", "0")
    plain = hip.load_llama_model(g)
    assert not plain.persist_info()["ready"]
    plain.prefill(tokens[:901])
    assert plain.decode_greedy(f0, 901, 150) == got
    dev.close(); plain.close()


@_retry_placement
def test_two_handles_on_one_device_do_not_wait_out_each_others_sessions(hip, orc, tmp_path):
    # a resident launch fills every compute unit; a second handle on the device (two models in one server) posts a quit into the
    # first one's mailbox before it queues work instead of waiting for the 2 ms idle limit: alternating calls stay correct and
    # cost far less than an idle-out each
    import time
    shape = synth.ModelShape("pd_two", 5, 256, 4, 4, 512, seq_len=64, interm=512)
    p = tmp_path / "m.gguf"
    synth.generate_gguf(str(p), shape, "q8_0", 173)
    g = gguf.load_gguf(str(p))
    tokens = synth.prompt_ids(24, shape.vocab, seed=31)
    want = _teacher_forced(orc, g, tokens)
    a, b = hip.load_llama_model(g), hip.load_llama_model(g)
    assert a.persist_info()["ready"] and b.persist_info()["ready"]
    a.forward(tokens[0], 0); b.forward(tokens[0], 0)        # (first launches: images, graphs)
    t0 = time.perf_counter()
    for pos in range(1, 24):
        a.forward(tokens[pos], pos)
        assert _rel(a.state.logits, want[pos]) <= LOGIT_TOL, pos
        b.forward(tokens[pos], pos)
        assert _rel(b.state.logits, want[pos]) <= LOGIT_TOL, pos
    dt = (time.perf_counter() - t0) / 46
    print(f"\ntwo handles alternating on one device: {dt * 1e3:.3f} ms per call")
    assert dt < 1.2e-3, dt          # (an idle-out per call would be >= 2 ms)
    assert _no_warning(a) and _no_warning(b)
    a.close(); b.close()


@pytest.mark.parametrize("wtype", ["q4_0", "q5_0"])
@_retry_placement
def test_persistent_decode_takes_q4_0_and_q5_0_files(hip, orc, tmp_path, monkeypatch, wtype):
    # a Q4_0 block is 32 values (nibble - 8) x d, a Q5_0 block (5-bit value - 16) x d (go/quant.go:45-94, :405-420): int8-valued
    # quants with an fp16 scale, i.e. exactly what the register images hold -- the packer expands them, the kernel is the Q8_0 one;
    # the embedding table is re-blocked as Q8_0 for the in-launch lookup.  Ids and logits against the oracle, the resident
    # session too, the launch plans of the same file as second witness.
    shape = synth.ModelShape(f"pd_{wtype}", 13, 256, 4, 4, 1024, seq_len=256, interm=512)
    p = tmp_path / "m.gguf"
    synth.generate_gguf(str(p), shape, wtype, 181)
    g = gguf.load_gguf(str(p))
    prompt = synth.prompt_ids(7, shape.vocab, seed=5)
    first, want, want_logits = _oracle_run(orc, g, prompt, 140)       # (positions 7 .. 146: a second attention pass from 128 on)
    dev = hip.load_llama_model(g)
    assert dev.persist_info()["ready"], dev.persist_info()
    dev.prefill(prompt)
    assert int(np.argmax(dev.state.logits)) == first
    got = dev.decode_greedy(first, len(prompt), 140)
    assert _no_warning(dev) and dev.persist_info()["tokens"] == 140
    assert got == want, (got[:10], want[:10])
    d = _rel(dev.debug_read("logits", shape.vocab), want_logits[-1])
    # per-call Forward on the resident session: teacher-forced with the oracle's ids
    dev.reset()
    dev.prefill(prompt)
    worst, tok = 0.0, first
    for i in range(30):
        dev.forward(tok, len(prompt) + i)
        worst = max(worst, _rel(dev.state.logits, want_logits[i]))
        tok = want[i]
    print(f"\npersistent decode, {wtype}: 140 ids equal; last-step max|gpu-oracle| = {d:.2e}, session {worst:.2e}")
    assert d <= LOGIT_TOL and worst <= LOGIT_TOL
    monkeypatch.setenv("NL_PERSIST", "0")
    plain = hip.load_llama_model(g)
    assert not plain.persist_info()["ready"]
    plain.prefill(prompt)
    assert plain.decode_greedy(first, len(prompt), 140) == want
    dev.close(); plain.close()


@pytest.mark.parametrize("which", ["small_kv2", "nano_kv3"])
@_retry_placement
def test_persistent_decode_with_grouped_query_heads(hip, orc, tmp_path, monkeypatch, which):
    # GQA (go/model.go:557-587: query head h attends over kv head h / (H / KV)): [Q; K; V] has D + 2 KV 64 rows, every head
    # unit gathers its group's k | v rows, the group's first head stores them.  The small shape with 2 kv heads and nano's
    # width with 3 (D 576, 9 heads): ids and logits against the oracle across a second attention pass (helpers included), the
    # resident session, the launch plans as second witness.
    shape = (synth.ModelShape("pd_gqa2", 13, 256, 4, 2, 1024, seq_len=256, interm=512) if which == "small_kv2"
             else synth.ModelShape("pd_nano_gqa3", 13, 576, 9, 3, 4096, seq_len=256))
    p = tmp_path / "m.gguf"
    synth.generate_gguf(str(p), shape, "q8_0" if which == "small_kv2" else "q4_0", 191)
    g = gguf.load_gguf(str(p))
    prompt = synth.prompt_ids(6, shape.vocab, seed=7)
    first, want, want_logits = _oracle_run(orc, g, prompt, 150)       # positions 6 .. 155
    dev = hip.load_llama_model(g)
    assert dev.persist_info()["ready"], dev.persist_info()
    dev.prefill(prompt)
    assert int(np.argmax(dev.state.logits)) == first
    got = dev.decode_greedy(first, len(prompt), 150)
    assert _no_warning(dev) and dev.persist_info()["tokens"] == 150
    assert got == want, (got[:10], want[:10])
    d = _rel(dev.debug_read("logits", shape.vocab), want_logits[-1])
    dev.reset()
    dev.prefill(prompt)
    worst, tok = 0.0, first
    for i in range(150):
        dev.forward(tok, len(prompt) + i)
        if i % 7 == 0 or i > 118:
            worst = max(worst, _rel(dev.state.logits, want_logits[i]))
        tok = want[i]
    print(f"\npersistent decode, GQA {which}: 150 ids equal; last-step max|gpu-oracle| = {d:.2e}, session {worst:.2e}")
    assert d <= LOGIT_TOL and worst <= LOGIT_TOL
    monkeypatch.setenv("NL_PERSIST", "0")
    plain = hip.load_llama_model(g)
    plain.prefill(prompt)
    assert plain.decode_greedy(first, len(prompt), 150) == want
    # the K / V rows it wrote are the launch plans'
    for which_c in ("k_cache", "v_cache"):
        n = shape.n_layer * shape.n_kv_head * shape.seq_len * 64
        a = dev.debug_read(which_c, n).reshape(-1, shape.seq_len, 64)[:, :150]
        b = plain.debug_read(which_c, n).reshape(-1, shape.seq_len, 64)[:, :150]
        assert np.abs(a - b).max() <= 2e-5, which_c
    dev.close(); plain.close()


@_retry_placement
def test_shapes_outside_the_instantiations_keep_the_launch_plans(hip, tmp_path):
    # kv-head counts without an instantiation, other widths, weight types that are not 32 int8-valued quants x an fp16 scale: not
    # candidates (the launch plans serve them)
    for shape, wt in ((synth.ModelShape("pd_gqa1", 2, 256, 4, 1, 512, seq_len=64, interm=512), "q8_0"),
                      (synth.ModelShape("pd_f16", 2, 256, 4, 4, 512, seq_len=64, interm=512), "f16"),
                      (synth.ModelShape("pd_q4k", 2, 256, 4, 4, 512, seq_len=64, interm=512), "q4_k"),
                      (synth.ModelShape("pd_wide", 2, 512, 8, 8, 512, seq_len=64, interm=1024), "q8_0")):
        p = tmp_path / f"{shape.name}.gguf"
        synth.generate_gguf(str(p), shape, wt, 141)
        dev = hip.load_llama_model(gguf.load_gguf(str(p)))
        assert not dev.persist_info()["ready"], shape.name
        assert len(dev.decode_greedy(3, 0, 8)) == 8
        dev.close()


def test_a_census_miss_is_forgiven_twice(hip, orc, tmp_path, monkeypatch):
    # where a launch landed is not how it exchanged: a launch whose census was not 8 x 32 has its chunk redone on the launch plans
    # (the oracle's ids either way) and the handle tries the persistent launch again; the third miss retires the path.  The misses
    # are simulated on the host side (NL_PERSIST_FAKE_CENSUS_MISS): the launches themselves place well
    shape = synth.ModelShape("pd_census", 13, 256, 4, 4, 1024, seq_len=96, interm=512)
    p = tmp_path / "m.gguf"
    synth.generate_gguf(str(p), shape, "q8_0", 163)
    g = gguf.load_gguf(str(p))
    prompt = synth.prompt_ids(6, shape.vocab, seed=15)
    first, want, _ = _oracle_run(orc, g, prompt, 50)
    monkeypatch.setenv("NL_QUIET", "1")
    for misses, ready_after in ((2, True), (3, False)):
        monkeypatch.setenv("NL_PERSIST_FAKE_CENSUS_MISS", str(misses))
        dev = hip.load_llama_model(g)
        dev.prefill(prompt)
        got, tok = [], first
        for k in range(5):
            ids = dev.decode_greedy(tok, len(prompt) + 10 * k, 10)
            got += ids
            tok = ids[-1]
            note = dev.last_error()
            if k < misses:
                assert "not placed 32 per XCD" in note and ("tries the persistent launch again" in note) == (k < 2), (misses, k, note)
        assert got == want[:50], misses
        info = dev.persist_info()
        assert info["ready"] == ready_after and info["launches"] == (5 if ready_after else 3), (misses, info)
        dev.close()
