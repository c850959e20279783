"""nanollama_amd.quantize against the reference's own re-quantiser output (hash recorded by
tests/golden/make_quantize_golden.py) and against the oracle's dequantiser."""
import hashlib
import json
import os

import numpy as np

from nanollama_amd import gguf, quantize
from oracle import oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_q8_0_requantisation_is_byte_identical_to_the_reference_tool(tmp_path):
    want = json.load(open(os.path.join(GOLDEN, "quantize_golden.json")))
    out = tmp_path / "q8.gguf"
    assert quantize.main([os.path.join(GOLDEN, want["input"]), str(out)]) == 0
    blob = out.read_bytes()
    assert len(blob) == want["size"] and hashlib.sha256(blob).hexdigest() == want["sha256"]
    g = gguf.load_gguf(str(out))
    assert g.meta.vocab_size == 512 and g.tensors["blk.0.attn_q.weight"].type == gguf.GGML_Q8_0
    assert g.tensors["blk.0.attn_norm.weight"].type == gguf.GGML_F32


def test_q4_0_requantisation_round_trips_within_half_a_step(tmp_path):
    out = tmp_path / "q4.gguf"
    quantize.requantize(os.path.join(GOLDEN, "tiny_f16.gguf"), str(out), "q4_0", verbose=False)
    src, dst = gguf.load_gguf(os.path.join(GOLDEN, "tiny_f16.gguf")), gguf.load_gguf(str(out))
    a, ia = src.get_tensor("blk.1.ffn_up.weight")
    b, ib = dst.get_tensor("blk.1.ffn_up.weight")
    assert ib.type == gguf.GGML_Q4_0 and ib.dims == ia.dims
    w = oracle.dequant(a, ia.type, ia.nel).reshape(-1, 32)
    wq = oracle.dequant(b, ib.type, ib.nel).reshape(-1, 32)
    step = np.abs(w).max(axis=1, keepdims=True) / 8
    # inside the representable range the error is half a step; +amax itself clamps to 7*d (ref rule, d > 0)
    assert np.all(np.abs(w - wq) <= step * 1.001 + 1e-7)
    assert np.median(np.abs(w - wq) / step.clip(1e-9)) < 0.3
