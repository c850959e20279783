"""Tokenizer port (nanollama_amd/tokenizer.py == go/tokenizer.go) against goldens produced by independent
implementations of the same algorithms (tests/golden/make_tokenizer_goldens.py)."""
import json
import os

import pytest

from nanollama_amd.gguf import GGUFMetadata
from nanollama_amd.tokenizer import Tokenizer, build_gpt2_byte_table

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tokenizer_golden.json"),
                      encoding="utf-8"))


def _sp():
    g = GOLD["sentencepiece"]
    m = GGUFMetadata(token_list=g["tokens"], token_scores=g["scores"], token_types=g["token_types"],
                     vocab_size=len(g["tokens"]), bos_id=g["bos_id"], eos_id=g["eos_id"], tokenizer_model="llama",
                     add_space_prefix=True)
    return Tokenizer(m), g


def _gpt2():
    g = GOLD["gpt2"]
    m = GGUFMetadata(token_list=g["tokens"], token_scores=[], token_types=g["token_types"], token_merges=g["merges"],
                     vocab_size=len(g["tokens"]), bos_id=g["bos_id"], eos_id=g["eos_id"], tokenizer_model="gpt2")
    return Tokenizer(m), g


def test_sentencepiece_bpe_matches_sentencepiece_library():
    tok, g = _sp()
    for case in g["cases"]:
        ids = tok.encode(case["text"], add_bos=False)
        assert ids == case["ids"], (case["text"], ids, case["ids"])
        assert tok.decode(ids) == case["decoded"]
    assert tok.encode("Hello world", add_bos=True)[0] == g["bos_id"]
    assert tok.encode("", add_bos=True) == [g["bos_id"]] and tok.encode("", add_bos=False) == []


def test_sentencepiece_control_tokens_are_atomic():
    tok, g = _sp()
    n_sp = g["n_sp"]
    specials = {t: i for i, t in enumerate(g["tokens"]) if i >= n_sp}
    name = next(iter(specials))
    ids = tok.encode(f"hi{name}there", add_bos=False)
    assert specials[name] in ids
    i = ids.index(specials[name])
    assert tok.decode(ids[:i]) == "hi"
    # control tokens vanish from Decode (go/tokenizer.go:347-349) but DecodeToken still shows the piece
    assert name not in tok.decode(ids) and tok.decode_token(specials[name]) == name
    assert tok.find_special_token(name.strip("<|>")) in (specials[name], -1)


def test_byte_fallback_roundtrip_and_streaming():
    tok, _ = _sp()
    text = "日本 🙂"
    ids = tok.encode(text, add_bos=False)
    assert tok.decode(ids) == text
    # streaming DecodeToken pieces are raw bytes that only form UTF-8 together (go/tokenizer.go:380-390)
    joined = b"".join(tok.decode_token_bytes(i) for i in ids)
    assert joined.decode("utf-8").lstrip(" ") == text
    assert tok.decode_token(-1) == "" and tok.decode_token(10**6) == ""


def test_gpt2_byte_level_bpe_matches_tokenizers_library():
    tok, g = _gpt2()
    for case in g["cases"]:
        ids = tok.encode(case["text"], add_bos=False)
        assert ids == case["ids"], (case["text"], ids, case["ids"])
        assert tok.decode(ids) == case["decoded"]


def test_gpt2_byte_table_is_a_bijection():
    b2u, u2b = build_gpt2_byte_table()
    assert len(set(b2u)) == 256 and all(u2b[b2u[b]] == b for b in range(256))
    assert b2u[ord("A")] == "A" and b2u[32] == "Ġ" and b2u[10] == "Ċ"
