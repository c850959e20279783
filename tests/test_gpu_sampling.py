"""On-device sampling (nl_op_sample / nl_sample_decode) against the host mirrors of go/main.go:177-200, :294-408."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sampling_kat as kat  # noqa: E402
import sampling_mirror as sm  # noqa: E402
from nanollama_amd import gguf  # noqa: E402

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def hip():
    from nanollama_amd import _lib, model
    if _lib.lib().nl_device_count() < 1:
        pytest.fail("no HIP device visible: the GPU tests must run on the MI355X box")
    return model


def _case(rng, V):
    lg = (rng.standard_normal(V) * rng.choice([0.7, 1.0, 3.0, 8.0])).astype(np.float32)
    window = int(rng.choice([0, 1, 8, 64]))
    recent = [int(t) for t in rng.integers(0, V, size=int(rng.integers(0, window + 1)))]
    if len(recent) >= 3:
        recent[-1] = recent[0]                 # a repeated token is penalised once per occurrence
    return lg, window, recent, float(rng.random(dtype=np.float32))


def test_device_answers_the_first_principles_kat(hip):
    # hand-computed picks (tests/sampling_kat.py: eight logits, every pick >= 9e-3 from a cdf boundary); the literal-Go
    # mirror answers the same table in tests/test_sampling_kat.py
    lg = np.array(kat.LOGITS, np.float32)
    for temp, top_p, u, want in kat.TOP_P:
        pick, after, rec = hip.op_sample(lg, temp, top_p, 50, 1.0, 4, u, [7])
        assert pick == want, (temp, top_p, u, pick)
        assert np.array_equal(after, lg) and rec == [7, want]        # penalty 1.0 leaves the logits alone
    for temp, k, u, want in kat.TOP_K:
        assert hip.op_sample(lg, temp, 1.0, k, 1.0, 0, u, [])[0] == want, (temp, k, u)
    tie = np.array(kat.TIE_LOGITS, np.float32)
    for temp, k, u, want in kat.TIE_TOP_K:
        assert hip.op_sample(tie, temp, 1.0, k, 1.0, 0, u, [])[0] == want, (temp, k, u)
    pick, after, rec = hip.op_sample(lg, 0.0, 0.9, 50, kat.PENALTY, 8, 0.5, kat.RECENT)
    assert after.tolist() == kat.LOGITS_AFTER_PENALTY
    assert pick == 6 and rec == kat.RECENT + [6]                      # argmax of the penalised logits: id 6 (1.5)


@pytest.mark.parametrize("V", [512, 4096, 32000, 96000])
def test_top_p_matches_the_go_chain(hip, V):
    # Primary oracle: the literal Go chain (sampling_mirror.go_top_p -- float32 throughout, one left-to-right sum).  The
    # device adds the V terms in fixed chunks instead, so a draw whose r (or the top-p cut) lies within float32 rounding
    # of a cumulative-probability boundary may land on the neighbouring candidate; device_top_p -- the same algorithm
    # in the device's summation order -- is consulted ONLY to show that a disagreement is such a boundary case
    # (margin < 5e-4 relative) and that the device did what its summation order implies.
    rng = np.random.default_rng(V)
    go_mismatch = 0
    trials = 24 if V <= 32000 else 8
    for _ in range(trials):
        lg, window, recent, u = _case(rng, V)
        temp, top_p, pen = float(rng.choice([0.8, 1.5])), float(rng.choice([0.5, 0.9, 0.95])), 1.15
        pick, lg_after, rec_after = hip.op_sample(lg, temp, top_p, 50, pen, window, u, recent)
        want_lg = sm.apply_penalty(lg, recent, pen, V)
        assert np.array_equal(lg_after, want_lg)                       # in-place penalty, bit for bit
        assert rec_after == sm.push_recent(recent, pick, window)
        if pick != sm.go_top_p(want_lg, temp, top_p, u):
            go_mismatch += 1
            explained, margin = sm.device_top_p(want_lg, temp, top_p, u)
            assert margin < 5e-4 and pick == explained, (V, temp, top_p, u, margin)
    assert go_mismatch <= max(2, trials // 8)


@pytest.mark.parametrize("V", [300, 4096, 32000, 50000, 128256])
def test_top_p_selection_without_a_sort_equals_its_integer_restatement(hip, V):
    # top-p takes no sort: weighted radix selection on exact integer weights (nl_sample.h) -- the candidates in the registers of
    # one workgroup up to 32768 of them (samp_select_radix_kernel), streamed out of L2 pass by pass beyond (the 7.9B tier's
    # 128256: samp_select_radix_stream_kernel).  sampling_mirror.device_top_p restates it with Python integers; the device must agree EXACTLY -- every
    # case, not only the ones the Go chain disagrees on -- including the shapes that leave the common path:
    #   * thousands of candidates with one identical p: ties go by ascending id, the cut / the pick need the k-th of them
    #     (64-bit division, the in-order rank search), and their level-1 bucket overflows the LDS candidate list (the
    #     lower levels then walk the registers);
    #   * all logits equal (one key: no levels at all), two distinct values, a single dominant token;
    #   * u = 0, u just below 1, top_p tiny and top_p just below 1.
    rng = np.random.default_rng(77 + V)
    cases = []
    for _ in range(10):
        lg, window, recent, u = _case(rng, V)
        cases.append((lg, float(rng.choice([0.8, 1.5])), float(rng.choice([0.5, 0.9, 0.95])), u))
    flat = np.zeros(V, np.float32)
    two = np.where(rng.random(V) < 0.5, np.float32(0.25), np.float32(-0.5)).astype(np.float32)
    ties = (rng.standard_normal(V) * 3.0).astype(np.float32)
    ties[rng.permutation(V)[: (V * 2) // 3]] = np.float32(1.0)              # two thirds of the vocabulary share one logit
    peak = (rng.standard_normal(V)).astype(np.float32)
    peak[V // 3] = 40.0
    for lg in (flat, two, ties, peak):
        for u in (0.0, 0.37, float(np.float32(1.0) - np.float32(2.0 ** -24))):
            for top_p in (0.9, 0.02, float(np.float32(1.0) - np.float32(2.0 ** -24))):
                cases.append((lg, 0.8, top_p, u))
    for lg, temp, top_p, u in cases:
        pick = hip.op_sample(lg, temp, top_p, 50, 1.0, 0, u, [])[0]
        want, _ = sm.device_top_p(lg, temp, top_p, u)
        assert pick == want, (V, temp, top_p, u, pick, want)


@pytest.mark.parametrize("V", [64, 512, 32000, 50000])        # (50000: the keys are streamed out of L2, samp_topk_kernel<true>)
def test_top_k_is_the_go_loop_exactly(hip, V):
    rng = np.random.default_rng(1000 + V)
    for _ in range(16):
        lg, window, recent, u = _case(rng, V)
        if rng.random() < 0.3:
            lg[int(rng.integers(0, V))] = lg.max()                     # tie for the largest logit: earlier index first
        temp, k, pen = float(rng.choice([0.5, 1.0])), int(rng.choice([1, 5, 50, 200])), float(rng.choice([1.0, 1.3]))
        pick, lg_after, rec_after = hip.op_sample(lg, temp, 1.0, k, pen, window, u, recent)
        want_lg = sm.apply_penalty(lg, recent, pen, V)
        assert np.array_equal(lg_after, want_lg)
        assert pick == sm.go_top_k(want_lg, temp, k, u)
        assert rec_after == sm.push_recent(recent, pick, window)


@pytest.mark.parametrize("V", [2048, 40000])
def test_top_k_selection_edge_cases(hip, V):
    # the selection without a sort (samp_topk_kernel): every logit equal (the Go list keeps the FIRST top_k indices), a plateau of
    # equal logits straddling the k-th place, top_k = 1024 (the device limit) and top_k > V, negative and denormal logits, -inf
    rng = np.random.default_rng(77 + V)
    cases = []
    cases.append((np.full(V, 0.25, np.float32), 50))
    lg = rng.normal(0, 2, V).astype(np.float32)
    lg[rng.choice(V, 300, replace=False)] = np.float32(lg.max() - 0.5)       # 300 equal logits around the 50th place
    cases.append((lg, 50)); cases.append((lg, 7)); cases.append((lg, 1024))
    neg = -np.abs(rng.normal(0, 3, V)).astype(np.float32)
    neg[::97] = np.float32(-1e-41)                                           # denormals
    neg[5::101] = -np.inf
    cases.append((neg, 200))
    cases.append((rng.normal(0, 1, V).astype(np.float32), 1))
    for lg, k in cases:
        for u in (0.0, 0.31, 0.77, float(np.float32(1.0) - np.float32(2.0 ** -24))):
            pick = hip.op_sample(lg.copy(), 0.9, 1.0, k, 1.0, 0, u, [])[0]
            assert pick == sm.go_top_k(lg, 0.9, k, u), (V, k, u)
    small = rng.normal(0, 1, 64).astype(np.float32)
    assert hip.op_sample(small.copy(), 0.9, 1.0, 500, 1.0, 0, 0.4, [])[0] == sm.go_top_k(small, 0.9, 500, 0.4)     # top_k > V


def test_zero_temperature_is_argmax_of_the_penalised_logits(hip):
    rng = np.random.default_rng(5)
    for _ in range(8):
        lg, window, recent, u = _case(rng, 4096)
        top = int(np.argmax(lg))
        recent = (recent + [top])[-window:] if window else []
        pick, lg_after, _ = hip.op_sample(lg, 0.0, 0.9, 50, 1.5, window, u, recent)
        want_lg = sm.apply_penalty(lg, recent, 1.5, 4096)
        assert np.array_equal(lg_after, want_lg) and pick == int(np.argmax(want_lg))


def test_sampler_argument_errors(hip):
    from nanollama_amd._lib import NlError
    lg = np.zeros(64, np.float32)
    with pytest.raises(NlError):
        hip.op_sample(lg, 0.8, 0.9, 50, 1.1, 2048, 0.5, [])       # window beyond the on-device limit
    with pytest.raises(NlError):
        hip.op_sample(lg, 0.8, 0.0, 50, 1.1, 8, 0.5, [])          # top_p must be positive


@pytest.mark.parametrize("mode", ["top_p", "top_k", "greedy_penalty"])
def test_sample_decode_loop_matches_host_loop(hip, mode):
    # The whole loop of go/main.go:173-219 on the device vs the same loop driven from the host with the logits read
    # back every step and the mirror sampler: identical ids, window and final logits.
    g = gguf.load_gguf(os.path.join(GOLDEN, "tiny_q8_0.gguf"))
    v = np.load(os.path.join(GOLDEN, "tiny_q8_0.npz"))
    prompt = [int(t) for t in v["prompt"]]
    temp, top_p, top_k, pen, window = {"top_p": (0.9, 0.9, 50, 1.15, 8), "top_k": (0.7, 1.0, 20, 1.15, 8),
                                       "greedy_penalty": (0.0, 0.9, 50, 1.3, 4)}[mode]
    n = 24
    us = np.random.default_rng(3).random(n, dtype=np.float32)
    V = g.meta.vocab_size
    host = hip.load_llama_model(g)
    host.prefill(prompt)
    recent, want_ids, pos = [], [], len(prompt)
    for i in range(n):
        lg = sm.apply_penalty(host.state.logits[:V].copy(), recent, pen, V)
        if temp <= 0:
            tok = int(np.argmax(lg))
        elif top_p < 1.0:
            tok = sm.device_top_p(lg, temp, top_p, float(us[i]))[0]
        else:
            tok = sm.go_top_k(lg, temp, top_k, float(us[i]))
        recent = sm.push_recent(recent, tok, window)
        want_ids.append(tok)
        host.forward(tok, pos)
        pos += 1
    dev = hip.load_llama_model(g)
    dev.prefill(prompt)
    ids1, rec1 = dev.sample_decode(len(prompt), 10, temp, top_p, top_k, pen, window, us[:10], [])
    ids2, rec2 = dev.sample_decode(len(prompt) + 10, n - 10, temp, top_p, top_k, pen, window, us[10:], rec1)   # chunked
    assert ids1 + ids2 == want_ids
    assert rec2 == recent
    dev.forward(want_ids[-1], pos - 1)          # same state afterwards: re-running the last position gives the host's logits
    assert np.array_equal(dev.state.logits, host.state.logits)
    # stops at seq_len like go/main.go:216
    ids3, _ = dev.sample_decode(g.meta.seq_len - 2, 10, temp, top_p, top_k, pen, window, us[:10], [])
    assert len(ids3) == 2
    host.close(); dev.close()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_engine_device_sampling_equals_the_host_loop(hip, seed):
    # Engine.generate_ids with the loop on the device vs the literal per-token host loop (logits read back every
    # token): same generator seed -> same ids, same token counter, and the generator ends in the same state.
    from nanollama_amd.engine import Engine, GenParams
    g = gguf.load_gguf(os.path.join(GOLDEN, "tiny_q8_0.gguf"))
    dev = hip.load_llama_model(g)
    outs = []
    for on_device in (True, False):
        # an EOS id that actually occurs mid-stream exercises the chunk truncation + generator rewind
        probe = Engine(dev, eos_id=-1, rep_penalty=1.15, rep_window=16, seed=seed, device_sampling=on_device, sample_chunk=8)
        ids = probe.generate_ids([1, 5, 6, 9], GenParams(max_tokens=40, temperature=0.9, top_p=0.9))
        eos = ids[21]
        eng = Engine(dev, eos_id=eos, rep_penalty=1.15, rep_window=16, seed=seed, device_sampling=on_device, sample_chunk=8)
        cut = eng.generate_ids([1, 5, 6, 9], GenParams(max_tokens=40, temperature=0.9, top_p=0.9))
        nxt = float(eng.rng.random(dtype=np.float32))
        topk = Engine(dev, eos_id=-1, rep_penalty=1.2, rep_window=4, seed=seed, device_sampling=on_device).generate_ids(
            [1, 5, 6, 9], GenParams(max_tokens=30, temperature=0.7, top_p=1.0, top_k=12))
        outs.append((ids, cut, eng.last_tokens, nxt, topk))
    assert outs[0] == outs[1]
    ids, cut = outs[0][0], outs[0][1]
    assert len(ids) == 40 and cut == ids[:len(cut)] and cut[-1] == ids[21] and len(cut) <= 22
    dev.close()
