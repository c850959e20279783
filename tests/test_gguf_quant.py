"""GGUF reader/writer + quantisers against fixtures written by the reference's
own writer and quantisers (tests/golden/make_goldens.py)."""
import os
import struct
from dataclasses import replace

import numpy as np
import pytest

from nanollama_amd import gguf, quant, synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_quantisers_match_reference_bytes():
    k = np.load(os.path.join(GOLDEN, "quant_kat.npz"))
    for name in ("randn", "edge", "rows576"):
        x = k[f"{name}_in"]
        assert quant.quantize_q8_0(x).tobytes() == k[f"{name}_q8"].tobytes(), name
        assert quant.quantize_q4_0(x).tobytes() == k[f"{name}_q4"].tobytes(), name
        assert quant.to_f16_bytes(x).tobytes() == k[f"{name}_f16"].tobytes(), name
    assert bytes(k["randn_q8"][:6]).hex() == "4424bcbbf1e6"  # SURVEY 8c captured prefixes
    assert bytes(k["randn_q4"][:6]).hex() == "3b343424a7b6"


CASES = [("tiny_f16", "tiny", "f16", 11, {}), ("tiny_q8_0", "tiny", "q8_0", 11, {}),
         ("tiny_q4_0", "tiny", "q4_0", 11, {}),
         ("tiny_qknorm_q8_0", "tiny", "q8_0", 12, dict(name="tiny_qknorm", qk_norm=True)),
         ("tiny_conj_q4_0", "tiny", "q4_0", 13, dict(name="tiny_conj", rope_conjugate=True)),
         ("tiny_tied_q8_0", "tiny", "q8_0", 14, dict(name="tiny_tied", tied=True)),
         ("tiny_mha_q4_0", "tiny_mha", "q4_0", 15, {})]


@pytest.mark.parametrize("tag,tier,wtype,seed,over", CASES)
def test_generator_is_byte_identical_to_reference_writer(tmp_path, tag, tier, wtype, seed, over):
    shape = replace(synth.TIERS[tier], **over)
    out = tmp_path / "x.gguf"
    synth.generate_gguf(str(out), shape, wtype, seed)
    assert out.read_bytes() == open(os.path.join(GOLDEN, tag + ".gguf"), "rb").read()


def test_loader_parses_reference_layout():
    g = gguf.load_gguf(os.path.join(GOLDEN, "tiny_q8_0.gguf"))
    m = g.meta
    assert (m.num_layers, m.embed_dim, m.num_heads, m.num_kv_heads, m.head_dim) == (2, 128, 4, 2, 32)
    assert (m.vocab_size, m.seq_len, m.interm_size) == (512, 64, 512)
    assert m.bos_id == 1 and m.eos_id == -1 and m.add_space_prefix is False
    assert abs(m.rms_norm_eps - 1e-5) < 1e-12 and m.rope_theta == 10000.0
    data, info = g.get_tensor("blk.0.attn_q.weight")
    assert info.type == gguf.GGML_Q8_0 and info.dims == (128, 128) and data.nbytes == 128 * 128 // 32 * 34
    data, info = g.get_tensor("blk.1.attn_norm.weight")
    assert info.type == gguf.GGML_F32 and data.nbytes == 128 * 4
    assert g.data_offset % 32 == 0 and all(t.offset % 32 == 0 for t in g.tensors.values())
    with pytest.raises(KeyError):
        g.get_tensor("nope")
    assert "output.weight" not in gguf.load_gguf(os.path.join(GOLDEN, "tiny_tied_q8_0.gguf")).tensors


def test_loader_defaults_and_quirks(tmp_path):
    # VocabSize comes only from the token list; llama.vocab_size is ignored (go/gguf.go:489-498);
    # HeadDim = D/H regardless of key_length (:461-463); kv heads default to heads (:464-466).
    w = gguf.GGUFWriter(str(tmp_path / "a.gguf"))
    w.add_string("general.architecture", "llama")
    w.add_uint32("llama.embedding_length", 64)
    w.add_uint32("llama.attention.head_count", 4)
    w.add_uint32("llama.attention.key_length", 999)
    w.add_uint32("llama.vocab_size", 77)
    w.add_string_array("tokenizer.ggml.tokens", ["a", "b", "c"])
    w.add_tensor_raw("t", np.zeros(32, np.uint8), gguf.GGML_F32, (8,))
    w.write()
    m = gguf.load_gguf(w.path).meta
    assert m.vocab_size == 3 and m.head_dim == 16 and m.num_kv_heads == 4
    assert m.bos_id == 1 and m.eos_id == 2 and m.add_space_prefix is True and m.tokenizer_model == "llama"


def test_loader_errors(tmp_path):
    p = tmp_path / "bad.gguf"
    p.write_bytes(struct.pack("<IIQQ", 0x12345678, 3, 0, 0) + b"\0" * 64)
    with pytest.raises(gguf.GGUFError, match="bad magic"):
        gguf.load_gguf(str(p))
    p.write_bytes(struct.pack("<IIQQ", gguf.GGUF_MAGIC, 1, 0, 0) + b"\0" * 64)
    with pytest.raises(gguf.GGUFError, match="unsupported GGUF version"):
        gguf.load_gguf(str(p))
    p.write_bytes(struct.pack("<IIQQ", gguf.GGUF_MAGIC, 3, 0, 0))
    with pytest.raises(gguf.GGUFError, match="no tensor data"):
        gguf.load_gguf(str(p))
    with pytest.raises(gguf.GGUFError, match="open GGUF"):
        gguf.load_gguf(str(tmp_path / "missing.gguf"))
    # tensor that runs past the end of the data section (go/gguf.go:569-572)
    w = gguf.GGUFWriter(str(tmp_path / "oob.gguf"))
    w.add_tensor_raw("t", np.zeros(32, np.uint8), gguf.GGML_F32, (64,))
    w.write()
    with pytest.raises(gguf.GGUFError, match="out of bounds"):
        gguf.load_gguf(w.path).get_tensor("t")


def test_tier_shapes_match_survey_table():
    t = synth.TIERS
    assert (t["nano"].ffn, t["mini"].ffn, t["goldie"].ffn, t["big"].ffn) == (1536, 2048, 4096, 11008)
    assert [round(t[k].matrix_params() / 1e6, 1) for k in ("nano", "mini", "goldie", "big")] == [70.2, 148.4, 767.4, 7481.6]
    # SURVEY 8 table quotes the matrices term alone; weight_bytes_per_token adds norms + one embedding row (8d formula)
    assert round(t["nano"].matrix_params() * 34 / 32 / 1e6, 2) == 74.58
    assert round(t["big"].matrix_params() * 18 / 32 / 1e6, 2) == 4208.39
    assert 74.58e6 < synth.weight_bytes_per_token(t["nano"], "q8_0") < 74.70e6
    assert round(synth.kv_bytes_per_token(t["nano"], 127) / 1e6, 1) == 7.7


def test_qrand_mode_is_deterministic_and_well_formed(tmp_path):
    a, b = tmp_path / "a.gguf", tmp_path / "b.gguf"
    synth.generate_gguf(str(a), synth.TIERS["tiny"], "q4_0", 5, mode="qrand")
    synth.generate_gguf(str(b), synth.TIERS["tiny"], "q4_0", 5, mode="qrand")
    assert a.read_bytes() == b.read_bytes()
    g = gguf.load_gguf(str(a))
    data, info = g.get_tensor("blk.0.ffn_up.weight")
    d = data.reshape(-1, 18)[:, :2].copy().view(np.float16).astype(np.float32)
    assert np.all(d > 0) and np.all(np.isfinite(d))
