#!/usr/bin/env python3
"""Golden vectors for the tokenizer port (nanollama_amd/tokenizer.py mirrors go/tokenizer.go).

The Go tokenizer cannot be run (no toolchain) and the reference ships no tokenizer test vectors, so the
pins come from INDEPENDENT implementations of the same algorithms, run in the build container:

  * SentencePiece mode: a small BPE model (byte fallback on) is trained with the `sentencepiece` package on this
    repository's own documentation; its vocabulary / scores / types are extracted with the REFERENCE's
    scripts/export_gguf.py::load_tokenizer_metadata (imported, not copied) -- exactly what ends up in a GGUF --
    padded with the reference's SPECIAL_TOKENS as control tokens; expected ids come from sentencepiece's own
    encoder (piece-level BPE by score is what go/tokenizer.go:267-295 implements).
  * GPT-2 byte-level mode: a byte-level BPE is trained with the `tokenizers` package using Qwen2's
    pre-tokenizer regex (the one hard-coded at go/tokenizer.go:83-89); expected ids come from that library.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_tokenizer_goldens.py
"""
import json
import os
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))

import sentencepiece as spm  # noqa: E402
from scripts import export_gguf as ref_exp  # noqa: E402  (reference)

CASES = [
    "Hello world", "hello", " leading space", "two  spaces", "The quick brown fox jumps over the lazy dog.",
    "kernel launch latency, 1.56 us", "MI355X gfx950 HBM3E 8 TB/s", "naive cafe résumé", "日本語のテキスト",
    "emoji 🙂 test", "tabs\tand\nnewlines", "", "a", "UPPER lower MiXeD", "x=y+z*(a-b)/c", "don't we'll I'm",
    "<|user_start|>hi there<|user_end|>", "pre<|bos|>post", "12345 67 8", "  ",
]


def corpus_text():
    parts = []
    for name in ("SURVEY.md", "DESIGN.md", "INTEGRATION.md", "BASELINE.md"):
        with open(os.path.join(REPO, name), encoding="utf-8") as f:
            parts.append(f.read())
    return "\n".join(parts)


def sentencepiece_goldens(tmp):
    corpus = os.path.join(tmp, "corpus.txt")
    with open(corpus, "w", encoding="utf-8") as f:
        f.write(corpus_text())
    prefix = os.path.join(tmp, "tok")
    spm.SentencePieceTrainer.train(input=corpus, model_prefix=prefix, vocab_size=700, model_type="bpe",
                                   byte_fallback=True, character_coverage=0.995, bos_id=1, eos_id=2, unk_id=0,
                                   pad_id=-1, add_dummy_prefix=True, normalization_rule_name="identity",
                                   remove_extra_whitespaces=False, split_digits=False, minloglevel=2)
    sp = spm.SentencePieceProcessor(model_file=prefix + ".model")
    meta = ref_exp.load_tokenizer_metadata(prefix + ".model", model_vocab_size=sp.get_piece_size() + 6)
    cases = []
    for text in CASES:
        if "<|" in text:
            continue  # control tokens are a GGUF-level notion; covered by the special-token cases below
        # go/tokenizer.go:206 prepends the space marker only when the text does not already start with a space;
        # sentencepiece always prepends its dummy prefix.  So Go(" x") == sentencepiece("x").
        sp_text = text[1:] if text.startswith(" ") else text
        ids = [int(i) for i in sp.encode(sp_text)]
        cases.append({"text": text, "ids": ids, "decoded": sp.decode(ids)})
    return {"model": "llama", "tokens": meta["tokens"], "scores": meta["scores"], "token_types": meta["token_types"],
            "bos_id": meta["bos_id"], "eos_id": meta["eos_id"], "cases": cases,
            "n_sp": sp.get_piece_size()}


def gpt2_goldens():
    from tokenizers import Regex, Tokenizer, decoders, models, pre_tokenizers, trainers
    pat = (r"(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}{1,3}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|\s+")
    tok = Tokenizer(models.BPE())
    tok.pre_tokenizer = pre_tokenizers.Sequence([pre_tokenizers.Split(Regex(pat), behavior="isolated"),
                                                 pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=False)])
    tok.decoder = decoders.ByteLevel()
    trainer = trainers.BpeTrainer(vocab_size=600, special_tokens=["<|endoftext|>", "<|im_start|>", "<|im_end|>"],
                                  initial_alphabet=pre_tokenizers.ByteLevel.alphabet(), show_progress=False)
    tok.train_from_iterator(corpus_text().split("\n"), trainer)
    blob = json.loads(tok.to_str())
    vocab = blob["model"]["vocab"]
    merges = [m if isinstance(m, str) else " ".join(m) for m in blob["model"]["merges"]]
    tokens = [None] * len(vocab)
    for piece, i in vocab.items():
        tokens[i] = piece
    types = [3 if t in ("<|endoftext|>", "<|im_start|>", "<|im_end|>") else 1 for t in tokens]
    cases = []
    for text in CASES:
        enc = tok.encode(text)
        cases.append({"text": text, "ids": enc.ids, "decoded": tok.decode(enc.ids, skip_special_tokens=True)})
    return {"model": "gpt2", "tokens": tokens, "merges": merges, "token_types": types, "bos_id": 0, "eos_id": 0,
            "cases": cases}


def main():
    with tempfile.TemporaryDirectory() as tmp:
        out = {"sentencepiece": sentencepiece_goldens(tmp), "gpt2": gpt2_goldens()}
    with open(os.path.join(OUT, "tokenizer_golden.json"), "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=False)
    print("sentencepiece vocab", len(out["sentencepiece"]["tokens"]), "cases", len(out["sentencepiece"]["cases"]),
          "| gpt2 vocab", len(out["gpt2"]["tokens"]), "merges", len(out["gpt2"]["merges"]))


if __name__ == "__main__":
    main()
