"""BPE tokenizer from GGUF metadata -- host-side mirror of go/tokenizer.go.

Two modes, auto-detected from tokenizer.ggml.model (go/tokenizer.go:67-136):
  * SentencePiece BPE ("llama"): U+2581 space marker, greedy highest-SCORE merges (:267-295), <0xNN>
    byte fallback (:298-336)
  * GPT-2 byte-level BPE ("gpt2", Qwen-style): bytes_to_unicode table (:49-64), Qwen2 pre-tokenizer regex
    (:83-89), lowest-RANK merges (:237-264)
Control tokens (token_type 3, longer than 2 bytes) are matched as whole units before BPE (:165-202).
Go strings are byte strings; byte-fallback pieces decode to raw bytes, so decode() works on bytes and only
turns them into str at the end.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import regex  # \p{L} / \p{N} classes, like Go's regexp

from .gguf import GGUFMetadata

_SPACE = "▁"
_GPT2_PRETOKEN = regex.compile(
    r"(?i:'s|'t|'re|'ve|'m|'ll|'d)"
    r"|[^\r\n\p{L}\p{N}]?\p{L}+"
    r"|\p{N}{1,3}"
    r"| ?[^\s\p{L}\p{N}]+[\r\n]*"
    r"|\s*[\r\n]+"
    r"|\s+")


def build_gpt2_byte_table():
    """buildGPT2ByteTable go/tokenizer.go:49-64"""
    byte_to_uni, uni_to_byte, n = [""] * 256, {}, 0
    for b in range(256):
        if 33 <= b <= 126 or 161 <= b <= 172 or 174 <= b <= 255:
            r = chr(b)
        else:
            r = chr(256 + n)
            n += 1
        byte_to_uni[b] = r
        uni_to_byte[r] = b
    return byte_to_uni, uni_to_byte


def _is_byte_piece(piece: str) -> bool:
    # go/tokenizer.go:351: len(piece) == 6 && "<0x" prefix && '>' suffix (byte length: ASCII only)
    return len(piece) == 6 and piece.isascii() and piece.startswith("<0x") and piece.endswith(">")


class Tokenizer:
    def __init__(self, meta: GGUFMetadata, verbose: bool = False):
        self.vocab: List[str] = meta.token_list
        self.scores: List[float] = meta.token_scores
        self.types: Optional[List[int]] = meta.token_types or None
        self.vocab_size = meta.vocab_size
        self.bos_id, self.eos_id = meta.bos_id, meta.eos_id
        self.is_gpt2 = meta.tokenizer_model == "gpt2"
        self.add_space_prefix = False if self.is_gpt2 else meta.add_space_prefix
        self.token_to_id: Dict[str, int] = {}
        for i, tok in enumerate(self.vocab):
            self.token_to_id[tok] = i      # later duplicates win, like the Go map build
        self.byte_tokens = [self.token_to_id.get("<0x%02X>" % i, -1) for i in range(256)]
        self.special_tokens: Dict[str, int] = {}
        if self.types is not None:
            for i, typ in enumerate(self.types):
                if typ == 3 and i < len(self.vocab) and len(self.vocab[i].encode("utf-8")) > 2:
                    self.special_tokens[self.vocab[i]] = i
        self.byte_to_unicode, self.unicode_to_byte = build_gpt2_byte_table() if self.is_gpt2 else ([], {})
        self.merge_rank: Dict[str, int] = {}
        if self.is_gpt2:
            for i, merge in enumerate(meta.token_merges):
                self.merge_rank[merge] = i
        if verbose:
            print(f"[tokenizer] vocab={self.vocab_size} bos={self.bos_id} eos={self.eos_id} gpt2={self.is_gpt2} "
                  f"add_space_prefix={self.add_space_prefix}")

    # ---- Encode go/tokenizer.go:139-162 ----
    def encode(self, text: str, add_bos: bool = True) -> List[int]:
        tokens: List[int] = []
        if add_bos and self.bos_id >= 0:
            tokens.append(self.bos_id)
        if not text:
            return tokens
        for seg in self._split_on_special_tokens(text):
            if seg in self.special_tokens:
                tokens.append(self.special_tokens[seg])
            elif self.is_gpt2:
                tokens.extend(self._encode_gpt2(seg))
            else:
                tokens.extend(self._encode_sentencepiece(seg))
        return tokens

    def _split_on_special_tokens(self, text: str) -> List[str]:
        """go/tokenizer.go:165-202: earliest match wins, longest on ties (byte offsets / byte lengths)."""
        if not self.special_tokens:
            return [text]
        segments, remaining = [], text.encode("utf-8")
        specials = [(t, t.encode("utf-8")) for t in self.special_tokens]
        while remaining:
            best_pos, best_len, best_tok = -1, 0, ""
            for tok, tb in specials:
                pos = remaining.find(tb)
                if pos >= 0 and (best_pos < 0 or pos < best_pos or (pos == best_pos and len(tb) > best_len)):
                    best_pos, best_len, best_tok = pos, len(tb), tok
            if best_pos < 0:
                segments.append(remaining.decode("utf-8", errors="replace"))
                break
            if best_pos > 0:
                segments.append(remaining[:best_pos].decode("utf-8", errors="replace"))
            segments.append(best_tok)
            remaining = remaining[best_pos + best_len:]
        return segments

    def _encode_sentencepiece(self, text: str) -> List[int]:
        if self.add_space_prefix and text and text[0] != " ":
            text = " " + text
        text = text.replace(" ", _SPACE)
        symbols = []
        for ch in text:                      # initialTokenizeSP :315-336
            if ch in self.token_to_id:
                symbols.append(ch)
            else:
                symbols.extend("<0x%02X>" % b for b in ch.encode("utf-8"))
        return self._symbols_to_ids(self._bpe_merge(symbols))

    def _bpe_merge(self, symbols: List[str]) -> List[str]:
        """go/tokenizer.go:267-295: merge the adjacent pair whose merged piece has the highest score (first wins ties)."""
        while True:
            best_score, best_idx = -1e30, -1
            for i in range(len(symbols) - 1):
                tid = self.token_to_id.get(symbols[i] + symbols[i + 1])
                if tid is not None:
                    sc = self.scores[tid]
                    if sc > best_score:
                        best_score, best_idx = sc, i
            if best_idx < 0:
                return symbols
            symbols = symbols[:best_idx] + [symbols[best_idx] + symbols[best_idx + 1]] + symbols[best_idx + 2:]

    def _encode_gpt2(self, text: str) -> List[int]:
        out: List[int] = []
        for chunk in _GPT2_PRETOKEN.findall(text):
            symbols = [self.byte_to_unicode[b] for b in chunk.encode("utf-8")]
            out.extend(self._symbols_to_ids(self._bpe_merge_gpt2(symbols)))
        return out

    def _bpe_merge_gpt2(self, symbols: List[str]) -> List[str]:
        """go/tokenizer.go:237-264: lowest merge rank first."""
        while True:
            best_rank, best_idx = len(self.merge_rank) + 1, -1
            for i in range(len(symbols) - 1):
                rank = self.merge_rank.get(symbols[i] + " " + symbols[i + 1])
                if rank is not None and rank < best_rank:
                    best_rank, best_idx = rank, i
            if best_idx < 0:
                return symbols
            symbols = symbols[:best_idx] + [symbols[best_idx] + symbols[best_idx + 1]] + symbols[best_idx + 2:]

    def _symbols_to_ids(self, symbols: List[str]) -> List[int]:
        """go/tokenizer.go:298-312: unknown symbols fall back to their bytes; bytes without a token are dropped."""
        tokens: List[int] = []
        for sym in symbols:
            tid = self.token_to_id.get(sym)
            if tid is not None:
                tokens.append(tid)
            else:
                tokens.extend(self.byte_tokens[b] for b in sym.encode("utf-8") if self.byte_tokens[b] >= 0)
        return tokens

    # ---- Decode go/tokenizer.go:339-406 ----
    def decode_token_bytes(self, tid: int) -> bytes:
        """DecodeToken (:380-406) as raw bytes (a byte-fallback token is one byte of a UTF-8 sequence)."""
        if tid < 0 or tid >= self.vocab_size:
            return b""
        piece = self.vocab[tid]
        if _is_byte_piece(piece):
            try:
                return bytes([int(piece[3:5], 16)])
            except ValueError:
                return b"\x00"
        if self.is_gpt2:
            out = bytearray()
            for r in piece:
                if r in self.unicode_to_byte:
                    out.append(self.unicode_to_byte[r])
                else:
                    out.extend(r.encode("utf-8"))
            return bytes(out)
        return piece.replace(_SPACE, " ").encode("utf-8")

    def decode_token(self, tid: int) -> str:
        return self.decode_token_bytes(tid).decode("utf-8", errors="replace")

    def decode(self, ids: List[int]) -> str:
        out = bytearray()
        for tid in ids:
            if tid < 0 or tid >= self.vocab_size:
                continue
            if self.types is not None and tid < len(self.types) and self.types[tid] == 3:
                continue                      # control tokens are skipped (:347-349)
            out.extend(self.decode_token_bytes(tid))
        if not self.is_gpt2 and self.add_space_prefix and out[:1] == b" ":
            out = out[1:]
        return bytes(out).decode("utf-8", errors="replace")

    def find_special_token(self, name: str) -> int:
        """go/tokenizer.go:409-421"""
        for v in (name, "<|" + name + "|>", "<" + name + ">"):
            if v in self.token_to_id:
                return self.token_to_id[v]
        return -1
