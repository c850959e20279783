"""The literal-Go sampling mirror (tests/sampling_mirror.py go_*: go/main.go:177-200, :294-398) pinned against
first-principles known-answer vectors (tests/sampling_kat.py).  The same vectors are asserted against the device in
tests/test_gpu_sampling.py, so the Go chain -- not the device-order mirror -- is what the device answers to."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sampling_kat as kat  # noqa: E402
import sampling_mirror as sm  # noqa: E402


def test_kat_table_is_the_plain_python_recomputation():
    for temp, top_p, u, want in kat.TOP_P:
        assert kat.expected_top_p(kat.LOGITS, temp, top_p, u) == want
    for temp, k, u, want in kat.TOP_K:
        assert kat.expected_top_k(kat.LOGITS, temp, k, u) == want
    for temp, k, u, want in kat.TIE_TOP_K:
        assert kat.expected_top_k(kat.TIE_LOGITS, temp, k, u) == want


def test_go_mirror_answers_the_kat():
    lg = np.array(kat.LOGITS, np.float32)
    for temp, top_p, u, want in kat.TOP_P:
        assert sm.go_top_p(lg, temp, top_p, u) == want, (temp, top_p, u)
        assert sm.device_top_p(lg, temp, top_p, u)[0] == want, (temp, top_p, u)
    for temp, k, u, want in kat.TOP_K:
        assert sm.go_top_k(lg, temp, k, u) == want, (temp, k, u)
    tie = np.array(kat.TIE_LOGITS, np.float32)
    for temp, k, u, want in kat.TIE_TOP_K:
        assert sm.go_top_k(tie, temp, k, u) == want, (temp, k, u)
    after = sm.apply_penalty(lg, kat.RECENT, kat.PENALTY, lg.size)
    assert after.tolist() == kat.LOGITS_AFTER_PENALTY
    assert sm.push_recent([1, 2, 3], 9, 3) == [2, 3, 9] and sm.push_recent([1], 9, 3) == [1, 9]


def test_integer_restatement_of_top_p_equals_the_go_chain_away_from_cdf_boundaries():
    # nl_sample.h selects by exact integer weights (device_top_p restates it with Python integers; the GPU tests hold the device
    # to that restatement bit for bit).  Here, without a GPU: over random distributions of several spreads and sizes the integer
    # selection picks what the literal Go float32 chain picks, except where u (or the top-p cut) lies within float32 rounding
    # of a cumulative-probability boundary -- and there it picks the neighbouring candidate.
    rng = np.random.default_rng(2024)
    total = differ = 0
    for V in (64, 1000, 4096):
        for _ in range(60):
            lg = (rng.standard_normal(V) * rng.choice([0.3, 1.0, 4.0])).astype(np.float32)
            temp, top_p = float(rng.choice([0.7, 1.0, 1.4])), float(rng.choice([0.3, 0.9, 0.97]))
            u = float(rng.random(dtype=np.float32))
            a = sm.go_top_p(lg, temp, top_p, u)
            b, margin = sm.device_top_p(lg, temp, top_p, u)
            total += 1
            if a != b:
                differ += 1
                assert margin < 5e-4, (V, temp, top_p, u, margin)
    assert differ <= max(2, total // 50), (differ, total)


def test_integer_restatement_orders_ties_by_id_and_needs_no_normalisation():
    # thousands of candidates with one p: ascending id within the tie; scaling all logits by a constant shift changes nothing
    lg = np.full(3000, np.float32(0.5))
    lg[7], lg[1999] = np.float32(3.0), np.float32(3.0)
    for u, top_p in ((0.0, 0.9), (0.49, 0.9), (0.999, 0.05), (0.73, 0.999)):
        pick, _ = sm.device_top_p(lg, 0.8, top_p, u)
        shifted, _ = sm.device_top_p((lg + np.float32(11.0)).astype(np.float32), 0.8, top_p, u)
        assert pick == shifted
    assert sm.device_top_p(lg, 0.8, 0.9, 0.0)[0] == 7                       # the first of the two largest
    assert sm.device_top_p(lg, 0.8, 1e-6, 0.9999)[0] == 7                   # a one-candidate nucleus
    flat = np.zeros(500, np.float32)
    assert sm.device_top_p(flat, 1.0, 0.5, 0.0)[0] == 0 and sm.device_top_p(flat, 1.0, 0.5, 0.9999)[0] == 249
