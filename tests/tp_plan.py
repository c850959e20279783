"""TEST INFRASTRUCTURE: the tensor-parallel shard plan restated on the host, so the CPU tests can
check with the oracle's GEMVs that such slices reproduce the full products.  The product slices on the device
(nl_upload_tensor -> matrix_upload call sites in nanollama_amd/csrc/nl_engine.hip); its results are checked on the GPU
against the single-GPU engine and the oracle (tests/test_gpu_parity.py, tests/test_gpu_p2p.py).

  attn_q / attn_k / attn_v : rows of this rank's heads (GQA groups stay local)
  attn_output              : COLUMNS of this rank's heads   -> partial [D], all-reduce(sum)
  ffn_gate / ffn_up        : rows [rank*I/G, (rank+1)*I/G)
  ffn_down                 : COLUMNS of the same range      -> partial [D], all-reduce(sum)
  output (LM head)         : rows [rank*V/G, ...)           -> logits slice, all-gather
  token_embd, norms        : replicated
A Q4_0/Q8_0 row is cols/32 independent blocks, so a column slice is a pure byte slice.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Tuple

import numpy as np

from nanollama_amd.gguf import ggml_block_elements, ggml_block_size


@dataclass(frozen=True)
class Slice:
    row0: int
    nrows: int
    col0: int
    ncols: int


def check_divisible(n_heads: int, n_kv_heads: int, interm: int, vocab: int, tp: int) -> None:
    if n_heads % tp or n_kv_heads % tp or vocab % tp or interm % (32 * tp):
        raise ValueError("heads / kv heads / vocab / interm(32-blocks) must divide by tp_size")


def shard_slices(dim: int, n_heads: int, n_kv_heads: int, head_dim: int, interm: int, vocab: int, tp: int,
                 rank: int) -> Dict[str, Slice]:
    check_divisible(n_heads, n_kv_heads, interm, vocab, tp)
    hs, kvs, is_, vs = n_heads // tp * head_dim, n_kv_heads // tp * head_dim, interm // tp, vocab // tp
    return {
        "attn_q.weight": Slice(rank * hs, hs, 0, dim),
        "attn_k.weight": Slice(rank * kvs, kvs, 0, dim),
        "attn_v.weight": Slice(rank * kvs, kvs, 0, dim),
        "attn_output.weight": Slice(0, dim, rank * hs, hs),
        "ffn_gate.weight": Slice(rank * is_, is_, 0, dim),
        "ffn_up.weight": Slice(rank * is_, is_, 0, dim),
        "ffn_down.weight": Slice(0, dim, rank * is_, is_),
        "output.weight": Slice(rank * vs, vs, 0, dim),
    }


def slice_raw(raw: np.ndarray, ggml_type: int, rows: int, cols: int, s: Slice) -> np.ndarray:
    """Cut a [rows, cols] tensor given as raw GGUF block bytes down to slice s (still raw bytes)."""
    be, bb = ggml_block_elements(ggml_type), ggml_block_size(ggml_type)
    if s.col0 % be or s.ncols % be:
        raise ValueError("column slices must fall on block boundaries")
    a = np.ascontiguousarray(raw).reshape(rows, cols // be * bb)
    return np.ascontiguousarray(a[s.row0:s.row0 + s.nrows, s.col0 // be * bb:(s.col0 + s.ncols) // be * bb]).reshape(-1)


def collectives_per_token(n_layers: int, dim: int, vocab: int, tp: int) -> Dict[str, Tuple[int, int]]:
    """(count, bytes each) of the collectives one decoded token needs."""
    if tp == 1:
        return {}
    return {"all_reduce_f32": (2 * n_layers, dim * 4), "all_gather_f32": (1, vocab // tp * 4)}
