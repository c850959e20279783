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
