"""Gamma essence loader -- mirror of LoadGamma (go/gamma.go:37-270): a sparse NPZ with
indices.npy (int32 / int64 token ids) and values.npy ([n, embed_dim] float16 or float32), optionally
vocab_size.npy / embed_dim.npy scalars.  Applied on the device by nl_set_gamma (embed[token] += gamma[token])."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class GammaEssence:
    vocab_size: int
    embed_dim: int
    num_tokens: int
    indices: np.ndarray   # int32 [n]
    values: np.ndarray    # [n, embed_dim] float16 or float32
    is_f16: bool


def load_gamma(path: str) -> GammaEssence:
    try:
        z = np.load(path)
    except Exception as exc:
        raise ValueError(f"open gamma npz: {exc}") from exc
    if "indices" not in z or "values" not in z:
        raise ValueError("gamma npz needs indices.npy and values.npy")
    idx = np.asarray(z["indices"]).astype(np.int32).reshape(-1)
    vals = np.asarray(z["values"])
    if vals.dtype not in (np.float16, np.float32):
        vals = vals.astype(np.float32)
    if vals.ndim != 2 or vals.shape[0] != idx.size:
        raise ValueError(f"gamma values shape {vals.shape} does not match {idx.size} indices")
    vocab = int(np.asarray(z["vocab_size"]).reshape(-1)[0]) if "vocab_size" in z else int(idx.max()) + 1 if idx.size else 0
    return GammaEssence(vocab, int(vals.shape[1]), int(idx.size), idx, np.ascontiguousarray(vals), vals.dtype == np.float16)
