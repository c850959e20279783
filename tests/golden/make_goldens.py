#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE's own
importable Python.  Runs only in the build container (needs /root/reference);
its outputs (small .gguf files + .npz vectors) are committed, this script is
the provenance record.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_goldens.py

What comes from the reference (imported, never copied):
  * scripts/export_gguf.py  GGUFWriter (file layout), tensor_to_q8_0 /
    tensor_to_q4_0 (quantisers)                       -> the .gguf fixtures
  * nanollama/llama.py      Llama, LlamaConfig        -> golden logits / ids
  * nanollama/engine.py     KVCache                   -> incremental decode path
What is ours: the random weights (nanollama_amd.synth.draw_float), the
dequantisation used to load quantised weights into the PyTorch model (plain
numpy below, independent of oracle/), and the float64-computed RoPE tables the
Go engine uses (go/model.go:346-358; PyTorch's default tables are bf16).
"""
import os
import struct
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from scripts import export_gguf as ref_exp  # noqa: E402  (reference)
from nanollama.llama import Llama, LlamaConfig  # noqa: E402  (reference)
from nanollama.engine import KVCache  # noqa: E402  (reference)

from nanollama_amd import synth  # noqa: E402
from nanollama_amd.gguf import GGML_F16, GGML_F32, GGML_Q4_0, GGML_Q8_0  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.manual_seed(0)
torch.set_num_threads(4)


def np_dequant(raw: bytes, t: int, shape) -> np.ndarray:
    """Independent numpy dequantiser (block formats per go/quant.go:22-31,103-108)."""
    a = np.frombuffer(raw, dtype=np.uint8)
    if t == GGML_F32:
        return a.view(np.float32).reshape(shape).copy()
    if t == GGML_F16:
        return a.view(np.float16).astype(np.float32).reshape(shape)
    if t == GGML_Q8_0:
        b = a.reshape(-1, 34)
        d = b[:, :2].copy().view(np.float16).astype(np.float32)
        q = b[:, 2:].copy().view(np.int8).astype(np.float32)
        return (q * d).reshape(shape)
    if t == GGML_Q4_0:
        b = a.reshape(-1, 18)
        d = b[:, :2].copy().view(np.float16).astype(np.float32)
        lo = (b[:, 2:] & 0x0F).astype(np.int32) - 8
        hi = (b[:, 2:] >> 4).astype(np.int32) - 8
        q = np.concatenate([lo, hi], axis=1).astype(np.float32)
        return (q * d).reshape(shape)
    raise ValueError(t)


def write_with_reference_writer(path, shape: synth.ModelShape, wtype: str, seed: int):
    """Same KV sequence as nanollama_amd.synth.generate_gguf, but every byte is
    produced by the reference's GGUFWriter + quantisers."""
    t = synth.WTYPES[wtype]
    w = ref_exp.GGUFWriter(path)
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
    w.add_string("tokenizer.ggml.model", "llama")
    w.add_string_array("tokenizer.ggml.tokens", synth.token_list(shape.vocab))
    w.add_float32_array("tokenizer.ggml.scores", [0.0] * shape.vocab)
    w.add_int32_array("tokenizer.ggml.token_type", [2, 3, 3] + [1] * (shape.vocab - 3))
    w.add_uint32("tokenizer.ggml.bos_token_id", 1)
    w.add_int32("tokenizer.ggml.eos_token_id", -1)
    w.add_bool("tokenizer.ggml.add_space_prefix", False)
    deq = {}
    for ckpt, gname, tshape, kind in synth.tensor_plan(shape):
        f32 = torch.from_numpy(synth.draw_float(shape, seed, ckpt, tshape, kind))
        tt = GGML_F32 if kind == "norm" else t
        w.add_tensor(gname, f32, tt)  # reference quantiser / converter
        name, raw, rt, rshape = w.tensors[-1]
        deq[ckpt] = np_dequant(raw, rt, rshape)
    with open(os.devnull, "w") as devnull:
        so = sys.stdout
        sys.stdout = devnull
        try:
            w.write()
        finally:
            sys.stdout = so
    return deq


def rope_tables_f64(hd: int, n: int, theta: float):
    """go/model.go:346-358: float64 math, cast to float32."""
    half = hd // 2
    th = float(np.float32(theta))
    i = np.arange(half, dtype=np.float64)
    freq = 1.0 / np.power(th, (2.0 * i) / float(hd))
    ang = np.arange(n, dtype=np.float64)[:, None] * freq[None, :]
    return np.cos(ang).astype(np.float32), np.sin(ang).astype(np.float32)


def build_ref_model(shape: synth.ModelShape, deq) -> Llama:
    cfg = LlamaConfig(sequence_len=shape.seq_len, vocab_size=shape.vocab, n_layer=shape.n_layer,
                      n_head=shape.n_head, n_kv_head=shape.n_kv_head, n_embd=shape.dim, norm_eps=shape.eps,
                      rope_theta=shape.rope_theta, tie_embeddings=shape.tied, use_qk_norm=shape.qk_norm)
    model = Llama(cfg)
    assert model.layers[0].ffn.gate_proj.weight.shape[0] == shape.ffn, "tiny shapes must follow the SwiGLU rule"
    sd = {k: torch.from_numpy(v.copy()) for k, v in deq.items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(m in ("output.weight",) and shape.tied for m in missing) or not missing, missing
    cos, sin = rope_tables_f64(shape.head_dim, model.rotary_seq_len, shape.rope_theta)
    if shape.rope_conjugate:  # (x0 c + x1 s, -x0 s + x1 c) == standard rotation with s -> -s
        sin = -sin
    model.cos = torch.from_numpy(cos)[None, :, None, :]
    model.sin = torch.from_numpy(sin)[None, :, None, :]
    return model.eval().float()


@torch.no_grad()
def golden_for(shape: synth.ModelShape, wtype: str, seed: int, tag: str, n_prompt=12, n_gen=32):
    path = os.path.join(OUT, f"{tag}.gguf")
    deq = write_with_reference_writer(path, shape, wtype, seed)
    model = build_ref_model(shape, deq)
    prompt = synth.prompt_ids(n_prompt, shape.vocab, seed=7)
    ids = torch.tensor([prompt], dtype=torch.long)
    full = model(ids)[0].numpy().astype(np.float32)  # [T, V] teacher-forced, no cache

    # incremental decode through the reference KVCache: prefill token-at-a-time, then greedy
    kv = KVCache(1, shape.n_kv_head, shape.seq_len, shape.head_dim, shape.n_layer, torch.device("cpu"), torch.float32)
    step_logits = []
    for t in prompt:
        lg = model(torch.tensor([[t]]), kv_cache=kv)[0, -1]
        step_logits.append(lg.numpy().astype(np.float32))
    inc = np.stack(step_logits)
    gen_ids, gen_logits, margins = [], [], []
    cur = inc[-1]
    for _ in range(n_gen):
        nxt = int(np.argmax(cur))
        top2 = np.partition(cur, -2)[-2:]
        margins.append(float(top2[1] - top2[0]))
        gen_ids.append(nxt)
        gen_logits.append(cur)
        cur = model(torch.tensor([[nxt]]), kv_cache=kv)[0, -1].numpy().astype(np.float32)
    d = float(np.abs(full - inc).max())
    np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), prompt=np.asarray(prompt, np.int32),
                        logits_full=full, logits_incremental=inc, greedy_ids=np.asarray(gen_ids, np.int32),
                        greedy_logits=np.stack(gen_logits), greedy_margins=np.asarray(margins, np.float32))
    print(f"{tag:28s} gguf={os.path.getsize(path)/1024:7.1f} KiB  |full-inc|={d:.2e}  "
          f"min margin={min(margins):.3e}  logit std={full.std():.3f}")


def quantiser_kats():
    """Bytes of the reference quantisers on fixed inputs (incl. edge cases)."""
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4, 64, generator=g)
    edge = torch.zeros(6, 32)
    edge[1] = torch.linspace(-1, 1, 32)                      # symmetric ramp
    edge[2, 5] = -3.0                                        # max is negative (ref Q4_0 keeps d > 0)
    edge[3] = torch.tensor([0.5, 1.5, 2.5, -0.5, -1.5, -2.5] * 5 + [127.0, -127.0])  # exact .5 ties, d = 1
    edge[4] = torch.full((32,), 1e-8)                        # scale is an fp16 subnormal / flushes to 0
    edge[5] = torch.tensor([8.0, -8.0, 7.5, -7.5, 0.5, -0.5, 3.49, 3.51] * 4)
    big = torch.randn(16, 576, generator=g) * 0.07
    out = {}
    for name, t in (("randn", x), ("edge", edge), ("rows576", big)):
        out[f"{name}_in"] = t.numpy().astype(np.float32)
        out[f"{name}_q8"] = np.frombuffer(ref_exp.tensor_to_q8_0(t), dtype=np.uint8)
        out[f"{name}_q4"] = np.frombuffer(ref_exp.tensor_to_q4_0(t), dtype=np.uint8)
        out[f"{name}_f16"] = np.frombuffer(ref_exp.tensor_to_bytes(t, torch.float16), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, "quant_kat.npz"), **out)
    print("quant_kat: q8 prefix", bytes(out["randn_q8"][:6]).hex(), " q4 prefix", bytes(out["randn_q4"][:6]).hex())


def block_kats():
    """Hand-built blocks with expected float32 outputs computed from first
    principles (pure Python, IEEE binary16 via struct 'e'), independent of both
    the oracle and numpy's dequant above."""
    def h(bits):
        return struct.unpack("<e", struct.pack("<H", bits))[0]
    scales = [0x0000, 0x8000, 0x0001, 0x03FF, 0x0400, 0x3C00, 0xBC00, 0x2E66, 0x7BFF, 0xFBFF, 0x1234, 0x8001]
    rng = np.random.Generator(np.random.PCG64(99))
    q8_blocks, q8_exp, q4_blocks, q4_exp = [], [], [], []
    for s in scales:
        q = rng.integers(-128, 128, size=32).astype(np.int8)
        q[0], q[1], q[2] = -128, 127, 0
        q8_blocks.append(struct.pack("<H", s) + q.tobytes())
        q8_exp.append([np.float32(np.float32(int(v)) * np.float32(h(s))) for v in q])
        nib = rng.integers(0, 16, size=32)
        nib[0], nib[1], nib[16] = 0, 15, 8
        packed = bytes(int(nib[j]) | (int(nib[j + 16]) << 4) for j in range(16))
        q4_blocks.append(struct.pack("<H", s) + packed)
        q4_exp.append([np.float32(np.float32(int(v) - 8) * np.float32(h(s))) for v in nib])
    half_bits = np.arange(0, 65536, 257, dtype=np.uint16)
    half_bits = np.concatenate([half_bits, np.array([0x0001, 0x03FF, 0x0400, 0x7BFF, 0x7C00, 0xFC00, 0x8000], np.uint16)])
    half_vals = np.array([h(int(b)) for b in half_bits if (int(b) & 0x7C00) != 0x7C00 or (int(b) & 0x3FF) == 0],
                         dtype=np.float32)
    half_bits = np.array([b for b in half_bits if (int(b) & 0x7C00) != 0x7C00 or (int(b) & 0x3FF) == 0], np.uint16)
    np.savez_compressed(os.path.join(OUT, "block_kat.npz"),
                        q8_blocks=np.frombuffer(b"".join(q8_blocks), np.uint8), q8_expected=np.asarray(q8_exp, np.float32),
                        q4_blocks=np.frombuffer(b"".join(q4_blocks), np.uint8), q4_expected=np.asarray(q4_exp, np.float32),
                        half_bits=half_bits, half_values=half_vals)
    print("block_kat:", len(scales), "blocks per format,", len(half_bits), "fp16 values")


def main():
    from dataclasses import replace
    tiny = synth.TIERS["tiny"]
    golden_for(tiny, "f16", 11, "tiny_f16")
    golden_for(tiny, "q8_0", 11, "tiny_q8_0")
    golden_for(tiny, "q4_0", 11, "tiny_q4_0")
    golden_for(replace(tiny, name="tiny_qknorm", qk_norm=True), "q8_0", 12, "tiny_qknorm_q8_0")
    golden_for(replace(tiny, name="tiny_conj", rope_conjugate=True), "q4_0", 13, "tiny_conj_q4_0")
    golden_for(replace(tiny, name="tiny_tied", tied=True), "q8_0", 14, "tiny_tied_q8_0")
    golden_for(synth.TIERS["tiny_mha"], "q4_0", 15, "tiny_mha_q4_0")
    quantiser_kats()
    block_kats()


if __name__ == "__main__":
    main()
