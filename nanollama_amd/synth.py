"""Deterministic random-weight GGUF generator.

There is no network for checkpoints, so every parity and bench input is a
random-weight GGUF written here in the reference exporter's exact layout
(scripts/export_gguf.py:520-610: KV order, tensors sorted by the checkpoint's
parameter names, 1-D tensors F32, 2-D tensors in the requested type).

Weights: numpy Generator(PCG64(seed)); matrices U(-s, s) with s = sqrt(3/D) --
the reference's own init scale (nanollama/llama.py:246-255) -- for ALL
matrices (the reference zero-inits c_proj/down_proj, which would make every
layer a no-op); embeddings N(0,1); norm weights U(0.5, 1.5).

mode="float": draw float32 weights, quantise with the reference rules
(nanollama_amd.quant).  mode="qrand": draw the quantised blocks directly
(uniform quants + a scale near s/127 or s/8) -- statistically the same model,
~20x faster to produce; used for the 7.9B tier where float generation takes
minutes.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np

from . import quant
from .gguf import (GGML_F16, GGML_F32, GGML_Q4_0, GGML_Q4_K, GGML_Q5_0, GGML_Q6_K, GGML_Q8_0, GGUFWriter,
                   ggml_block_elements, ggml_block_size)


@dataclass(frozen=True)
class ModelShape:
    name: str
    n_layer: int
    dim: int
    n_head: int
    n_kv_head: int
    vocab: int
    seq_len: int = 2048
    eps: float = 1e-5
    rope_theta: float = 10000.0
    qk_norm: bool = False
    rope_conjugate: bool = False
    tied: bool = False
    interm: int = 0  # 0 -> SwiGLU rule
    attn_bias: bool = False  # Qwen-style attention biases (go/model.go:244-247); nanollama models have none

    @property
    def head_dim(self) -> int:
        return self.dim // self.n_head

    @property
    def kv_dim(self) -> int:
        return self.n_kv_head * self.head_dim

    @property
    def ffn(self) -> int:
        if self.interm:
            return self.interm
        # nanollama/llama.py:178-179, scripts/export_gguf.py:348-351
        hidden = int(2 * (4 * self.dim) / 3)
        return 256 * ((hidden + 255) // 256)

    def matrix_params(self) -> int:
        """2-D parameters read per decoded token (layers + LM head)."""
        d, i = self.dim, self.ffn
        per_layer = d * d * 2 + 2 * self.kv_dim * d + 3 * d * i
        return self.n_layer * per_layer + self.vocab * d


# NAMED_CONFIGS nanollama/llama.py:40-51; vocab tiers README.md:49-54.
TIERS: Dict[str, ModelShape] = {
    "nano": ModelShape("nano", 13, 576, 9, 9, 32000),
    "mini": ModelShape("mini", 20, 768, 12, 3, 32000),
    "goldie": ModelShape("goldie", 28, 1536, 24, 6, 48000),
    "big": ModelShape("big", 40, 4096, 64, 16, 96000),
    # test-sized shapes
    "tiny": ModelShape("tiny", 2, 128, 4, 2, 512, seq_len=64, interm=512),
    "tiny_mha": ModelShape("tiny_mha", 2, 128, 2, 2, 512, seq_len=64, interm=512),
    "small_test": ModelShape("small_test", 3, 192, 3, 3, 1024, seq_len=256),
}

TIER_SEED = {"nano": 1000, "mini": 1001, "goldie": 1002, "big": 1003}

WTYPES = {"f32": GGML_F32, "f16": GGML_F16, "q8_0": GGML_Q8_0, "q4_0": GGML_Q4_0,
          # the remaining formats the Go engine reads (go/quant.go:171-484); no reference quantiser exists for them,
          # so these are always drawn directly as random blocks (mode "qrand")
          "q5_0": GGML_Q5_0, "q4_k": GGML_Q4_K, "q6_k": GGML_Q6_K}
QRAND_ONLY = (GGML_Q5_0, GGML_Q4_K, GGML_Q6_K)


def token_list(vocab: int) -> List[str]:
    """V single-code-point pieces so prompts map 1:1 to ids."""
    toks = ["<unk>", "<s>", "</s>"]
    for i in range(3, vocab):
        cp = 0x100 + i
        if cp >= 0xD800:
            cp += 0x800  # skip the surrogate range
        toks.append(chr(cp))
    return toks[:vocab]


def tensor_plan(shape: ModelShape) -> List[Tuple[str, str, Tuple[int, ...], str]]:
    """(checkpoint name, gguf name, shape, kind) in the exporter's order:
    sorted() over the checkpoint's parameter names (export_gguf.py:587)."""
    d, i, kv = shape.dim, shape.ffn, shape.kv_dim
    plan = {}
    for l in range(shape.n_layer):
        p, g = f"layers.{l}.", f"blk.{l}."
        plan[p + "attn.c_q.weight"] = (g + "attn_q.weight", (shape.n_head * shape.head_dim, d), "matrix")
        plan[p + "attn.c_k.weight"] = (g + "attn_k.weight", (kv, d), "matrix")
        plan[p + "attn.c_v.weight"] = (g + "attn_v.weight", (kv, d), "matrix")
        plan[p + "attn.c_proj.weight"] = (g + "attn_output.weight", (d, d), "matrix")
        plan[p + "attn_norm.weight"] = (g + "attn_norm.weight", (d,), "norm")
        if shape.attn_bias:
            plan[p + "attn.c_q.bias"] = (g + "attn_q.bias", (shape.n_head * shape.head_dim,), "bias")
            plan[p + "attn.c_k.bias"] = (g + "attn_k.bias", (kv,), "bias")
            plan[p + "attn.c_v.bias"] = (g + "attn_v.bias", (kv,), "bias")
            plan[p + "attn.c_proj.bias"] = (g + "attn_output.bias", (d,), "bias")
        plan[p + "ffn.gate_proj.weight"] = (g + "ffn_gate.weight", (i, d), "matrix")
        plan[p + "ffn.up_proj.weight"] = (g + "ffn_up.weight", (i, d), "matrix")
        plan[p + "ffn.down_proj.weight"] = (g + "ffn_down.weight", (d, i), "matrix")
        plan[p + "ffn_norm.weight"] = (g + "ffn_norm.weight", (d,), "norm")
    plan["norm.weight"] = ("output_norm.weight", (d,), "norm")
    if not shape.tied:
        plan["output.weight"] = ("output.weight", (shape.vocab, d), "matrix")
    plan["tok_embeddings.weight"] = ("token_embd.weight", (shape.vocab, d), "embedding")
    return [(k,) + plan[k] for k in sorted(plan)]


def _tensor_rng(seed: int, ckpt_name: str) -> np.random.Generator:
    # independent stream per tensor so generation order / chunking cannot change values
    h = np.frombuffer(ckpt_name.encode(), dtype=np.uint8).astype(np.uint64)
    key = int((h * np.arange(1, len(h) + 1, dtype=np.uint64)).sum() % (1 << 31))
    return np.random.Generator(np.random.PCG64([seed, key]))


def draw_float(shape: ModelShape, seed: int, ckpt_name: str, tshape, kind: str) -> np.ndarray:
    rng = _tensor_rng(seed, ckpt_name)
    if kind == "norm":
        return rng.uniform(0.5, 1.5, size=tshape).astype(np.float32)
    if kind == "bias":
        return rng.uniform(-0.5, 0.5, size=tshape).astype(np.float32)
    if kind == "embedding":
        return rng.standard_normal(size=tshape, dtype=np.float32)
    s = np.float32((3.0 ** 0.5) * (shape.dim ** -0.5))
    return ((rng.random(size=tshape, dtype=np.float32) * np.float32(2.0) - np.float32(1.0)) * s).astype(np.float32)


def _draw_qrand(shape: ModelShape, seed: int, ckpt_name: str, tshape, kind: str, wtype: int) -> np.ndarray:
    """Random quantised blocks drawn directly: every quant byte uniform, every fp16 scale
    d = 2^e * (1 + m/1024) with random mantissa m and e = floor(log2(amax / qmax)), i.e. the same
    magnitude the reference quantiser would produce for U(-amax, amax) weights."""
    rng = _tensor_rng(seed, ckpt_name)
    nel = int(np.prod(tshape))
    amax = 4.0 if kind == "embedding" else (3.0 ** 0.5) * (shape.dim ** -0.5)

    def f16_bits(target: float, n: int) -> np.ndarray:
        """n fp16 values 2^e * (1 + m/1024), random mantissa, e = floor(log2(target))."""
        efield = max(1, min(30, int(np.floor(np.log2(target))) + 15))
        return ((rng.integers(0, 1 << 16, size=n, dtype=np.int64).astype(np.uint16) & np.uint16(0x03FF))
                | np.uint16(efield << 10))

    if wtype in (GGML_Q4_K, GGML_Q6_K):
        nsb = nel // 256
        bsz = 144 if wtype == GGML_Q4_K else 210
        raw = rng.integers(0, 1 << 63, size=(nsb * bsz + 7) // 8, dtype=np.int64).view(np.uint8)[:nsb * bsz]
        blocks = raw.reshape(nsb, bsz)
        if wtype == GGML_Q4_K:   # w = d*sc*q - dmin*m, sc,m in 0..63, q in 0..15
            blocks[:, 0:2] = f16_bits(amax / 470.0, nsb).view(np.uint8).reshape(nsb, 2)
            blocks[:, 2:4] = f16_bits(amax / 64.0, nsb).view(np.uint8).reshape(nsb, 2)
        else:                    # w = d*sc*(q-32), sc int8, q in 0..63
            blocks[:, 208:210] = f16_bits(amax / 4096.0, nsb).view(np.uint8).reshape(nsb, 2)
        return raw
    nb = nel // 32
    bsz = {GGML_Q8_0: 34, GGML_Q4_0: 18, GGML_Q5_0: 22}[wtype]
    target = amax / {GGML_Q8_0: 127.0, GGML_Q4_0: 8.0, GGML_Q5_0: 16.0}[wtype]
    nbytes = nb * bsz
    raw = rng.integers(0, 1 << 63, size=(nbytes + 7) // 8, dtype=np.int64).view(np.uint8)[:nbytes]
    blocks = raw.reshape(nb, bsz)
    blocks[:, 0:2] = f16_bits(target, nb).view(np.uint8).reshape(nb, 2)
    return raw


def generate_gguf(path: str, shape: ModelShape, wtype: str = "q8_0", seed: Optional[int] = None,
                  mode: str = "float", keep_float: bool = False, stream_scale: float = 1.0) -> Dict[str, np.ndarray]:
    """Write a random-weight GGUF.  Returns {gguf_name: float32 weights} when
    keep_float (test-sized models only).  stream_scale (float mode): the embedding and the two projections that write the
    residual stream (attention output, down) are multiplied by it -- a model whose residual stream lives at that magnitude
    while everything behind an RMSNorm stays O(1)."""
    t = WTYPES[wtype]
    if seed is None:
        seed = TIER_SEED.get(shape.name, 4242)
    w = GGUFWriter(path)
    # KV order == scripts/export_gguf.py:520-537,556-561
    w.add_string("general.architecture", "llama")
    w.add_string("general.name", f"nanollama-synth-{shape.name}")
    w.add_uint32("llama.block_count", shape.n_layer)
    w.add_uint32("llama.embedding_length", shape.dim)
    w.add_uint32("llama.attention.head_count", shape.n_head)
    w.add_uint32("llama.attention.head_count_kv", shape.n_kv_head)
    w.add_uint32("llama.attention.key_length", shape.head_dim)
    w.add_uint32("llama.attention.value_length", shape.head_dim)
    w.add_uint32("llama.feed_forward_length", shape.ffn)
    w.add_uint32("llama.context_length", shape.seq_len)
    w.add_float32("llama.attention.layer_norm_rms_epsilon", shape.eps)
    w.add_float32("llama.rope.freq_base", shape.rope_theta)
    w.add_uint32("llama.vocab_size", shape.vocab)
    w.add_bool("nanollama.qk_norm", shape.qk_norm)
    w.add_bool("nanollama.rope_conjugate", shape.rope_conjugate)
    toks = token_list(shape.vocab)
    w.add_string("tokenizer.ggml.model", "llama")
    w.add_string_array("tokenizer.ggml.tokens", toks)
    w.add_float32_array("tokenizer.ggml.scores", [0.0] * shape.vocab)
    w.add_int32_array("tokenizer.ggml.token_type", [2, 3, 3] + [1] * (shape.vocab - 3))
    w.add_uint32("tokenizer.ggml.bos_token_id", 1)
    w.add_int32("tokenizer.ggml.eos_token_id", -1)  # random models must not stop early
    w.add_bool("tokenizer.ggml.add_space_prefix", False)

    floats: Dict[str, np.ndarray] = {}
    for ckpt, gname, tshape, kind in tensor_plan(shape):
        if kind in ("norm", "bias"):
            f32 = draw_float(shape, seed, ckpt, tshape, kind)
            w.add_tensor_raw(gname, quant.to_f32_bytes(f32), GGML_F32, tshape)
            if keep_float:
                floats[gname] = f32
            continue
        if (mode == "qrand" and t in (GGML_Q8_0, GGML_Q4_0)) or t in QRAND_ONLY:
            w.add_tensor_raw(gname, _draw_qrand(shape, seed, ckpt, tshape, kind, t), t, tshape)
            continue
        f32 = draw_float(shape, seed, ckpt, tshape, kind)
        if stream_scale != 1.0 and (kind == "embedding" or ckpt.endswith(("attn.c_proj.weight", "ffn.down_proj.weight"))):
            f32 = (f32 * np.float32(stream_scale)).astype(np.float32)
        w.add_tensor_raw(gname, quant.encode(f32, t), t, tshape)
        if keep_float:
            floats[gname] = f32
    os.makedirs(os.path.dirname(os.path.abspath(path)) or ".", exist_ok=True)
    w.write()
    return floats


def prompt_ids(n: int, vocab: int, seed: int = 7) -> List[int]:
    """Fixed prompt ids from PCG64(seed); BOS=1 first, as Encode(addBos) does."""
    rng = np.random.Generator(np.random.PCG64(seed))
    ids = rng.integers(3, vocab, size=max(n - 1, 0))
    return [1] + [int(x) for x in ids]


def weight_bytes_per_token(shape: ModelShape, wtype: str) -> float:
    """Algorithmic weight bytes per decoded token: every 2-D tensor
    in the layers + LM head once, norms, one embedding row."""
    t = WTYPES[wtype]
    bpe = ggml_block_size(t) / ggml_block_elements(t)
    return shape.matrix_params() * bpe + (2 * shape.n_layer + 1) * shape.dim * 4 + shape.dim * bpe


def kv_bytes_per_token(shape: ModelShape, pos: int) -> float:
    """f32 KV read once per element at position pos + the write of one position."""
    return shape.n_layer * (pos + 1) * shape.kv_dim * 2 * 4 + shape.n_layer * shape.kv_dim * 2 * 4
