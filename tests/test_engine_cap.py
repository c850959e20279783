"""The id-level generation loop (nanollama_amd/engine.py, mirror of go/main.go:152-230) stops where the Go loop's
`len(output) < 8192` condition stops it: no further token is sampled, and the generator, the recent window and the
token counter are the same on the greedy, device-sampling and host-sampling paths.  Driven with a stand-in model (no
GPU): every step has one overwhelmingly likely token, so the three paths generate the same ids."""
from types import SimpleNamespace

import numpy as np

from nanollama_amd.engine import Engine, GenParams

V = 64


class FakeModel:
    def __init__(self):
        self.config = SimpleNamespace(vocab_size=V, seq_len=256)
        self.state = SimpleNamespace(logits=np.zeros(V, np.float32))
        self.forwards = 0

    def _set(self, tok, pos):
        self.state.logits[:] = -50.0
        self.state.logits[(tok * 7 + pos * 3 + 1) % V] = 50.0

    def reset(self):
        pass

    def prefill(self, toks):
        self._set(toks[-1], len(toks) - 1)

    def forward(self, tok, pos):
        self.forwards += 1
        self._set(tok, pos)

    def decode_greedy(self, tok, pos, n):
        out = []
        for k in range(n):
            self.forward(tok, pos + k)
            tok = int(np.argmax(self.state.logits))
            out.append(tok)
        return out

    def sample_decode(self, pos, n, temp, top_p, top_k, pen, window, us, recent):
        ids, rec = [], list(recent)
        for k in range(n):
            tok = int(np.argmax(self.state.logits))     # (the dominant token survives any penalty / any uniform)
            ids.append(tok)
            rec = (rec + [tok])[-window:] if window else []
            self.forward(tok, pos + k)
        return ids, rec


def _run(mode, cap_after, max_tokens=40):
    m = FakeModel()
    kw = dict(eos_id=-1, rep_window=6, seed=11)
    if mode == "greedy":
        eng, p = Engine(m, rep_penalty=1.0, **kw), GenParams(max_tokens=max_tokens, temperature=0.0)
    else:
        eng = Engine(m, rep_penalty=1.15, device_sampling=mode == "device", sample_chunk=8, **kw)
        p = GenParams(max_tokens=max_tokens, temperature=0.8, top_p=0.9)
    seen = []

    def on_token(t):
        seen.append(t)
        return cap_after is not None and len(seen) >= cap_after

    ids = eng.generate_ids([1, 2, 3], p, on_token=on_token)
    return ids, seen, eng.last_tokens, float(eng.rng.random(dtype=np.float32))


def test_output_cap_stops_every_path_after_the_same_token():
    full = _run("greedy", None)[0]
    assert len(full) == 40
    for cap in (1, 5, 8, 13):                      # inside a chunk, on a chunk boundary (sample_chunk = 8), across chunks
        runs = {mode: _run(mode, cap) for mode in ("greedy", "device", "host")}
        for mode, (ids, seen, last, nxt) in runs.items():
            assert ids == full[:cap] and seen == full[:cap], (mode, cap)
            assert last == min(cap, 6), (mode, cap)               # len(recentTokens), capped by --rep-window (go/main.go:198-200,223)
        # the sampling paths leave the generator where the per-token Go loop would: `cap` draws
        assert runs["device"][3] == runs["host"][3]
        ref = np.random.default_rng(11)
        ref.random(cap, dtype=np.float32)
        assert runs["host"][3] == float(ref.random(dtype=np.float32))


def test_without_a_cap_all_tokens_are_emitted():
    for mode in ("greedy", "device", "host"):
        ids, seen, last, _ = _run(mode, None, max_tokens=20)
        assert len(ids) == 20 and seen == ids and last == 6


def test_greedy_path_stops_the_device_within_a_chunk_of_the_cap():
    # (advisor, round 3) the greedy fast path used to run all max_tokens steps and truncate afterwards: the rate was
    # understated and the device state ran past where the Go loop stops
    m = FakeModel()
    eng = Engine(m, rep_penalty=1.0, eos_id=-1, rep_window=6, seed=11)
    eng.greedy_chunk = 16
    seen = []
    ids = eng.generate_ids([1, 2, 3], GenParams(max_tokens=200, temperature=0.0),
                           on_token=lambda t: seen.append(t) or len(seen) >= 20)
    assert len(ids) == 20 and seen == ids
    assert m.forwards <= 19 + 16                # sample_0 from the prefill, then at most one chunk beyond the 19 kept steps
    full = Engine(FakeModel(), rep_penalty=1.0, eos_id=-1, rep_window=6, seed=11).generate_ids(
        [1, 2, 3], GenParams(max_tokens=200, temperature=0.0))
    assert len(full) == 200 and full[:20] == ids
