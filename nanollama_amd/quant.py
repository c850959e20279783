"""Vectorised block quantisers producing the reference's exact bytes.

Rules follow the reference exporter, not ggml:
  * Q8_0: scripts/export_gguf.py:124-159 -- d = amax/127 (float32), zero
    blocks get d = 1, q = clamp(round_half_even(x / d), -128, 127), the stored
    scale is float16(d) while the division uses the float32 d.
  * Q4_0: scripts/export_gguf.py:85-121 -- POSITIVE d = amax/8 (unlike ggml's
    signed-max rule), q = clamp(round(x / d) + 8, 0, 15), low nibble =
    element j, high nibble = element j+16.
Pinned by tests/test_gguf_quant.py against bytes captured from the reference
functions (tests/golden/quant_kat.npz).
"""
from __future__ import annotations

import numpy as np

from .gguf import GGML_F16, GGML_F32, GGML_Q4_0, GGML_Q8_0

QK = 32


def quantize_q8_0(w: np.ndarray) -> np.ndarray:
    """float array (any shape, size % 32 == 0) -> uint8 array of 34-byte blocks."""
    t = np.ascontiguousarray(w, dtype=np.float32).reshape(-1, QK)
    amax = np.abs(t).max(axis=1)
    scales = amax / np.float32(127.0)
    scales[scales == 0] = np.float32(1.0)
    q = np.clip(np.rint(t / scales[:, None]), -128, 127).astype(np.int8)
    out = np.empty((t.shape[0], 34), dtype=np.uint8)
    out[:, 0:2] = scales.astype(np.float16).view(np.uint8).reshape(-1, 2)
    out[:, 2:] = q.view(np.uint8)
    return out.reshape(-1)


def quantize_q4_0(w: np.ndarray) -> np.ndarray:
    """float array (size % 32 == 0) -> uint8 array of 18-byte blocks."""
    t = np.ascontiguousarray(w, dtype=np.float32).reshape(-1, QK)
    amax = np.abs(t).max(axis=1)
    scales = amax / np.float32(8.0)
    scales[scales == 0] = np.float32(1.0)
    q = np.clip(np.rint(t / scales[:, None]) + np.float32(8.0), 0, 15).astype(np.uint8)
    out = np.empty((t.shape[0], 18), dtype=np.uint8)
    out[:, 0:2] = scales.astype(np.float16).view(np.uint8).reshape(-1, 2)
    out[:, 2:] = q[:, :16] | (q[:, 16:] << 4)
    return out.reshape(-1)


def to_f16_bytes(w: np.ndarray) -> np.ndarray:
    """scripts/export_gguf.py:78-82 with target float16."""
    return np.ascontiguousarray(w, dtype=np.float32).astype(np.float16).view(np.uint8).reshape(-1)


def to_f32_bytes(w: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(w, dtype=np.float32).view(np.uint8).reshape(-1)


def encode(w: np.ndarray, ggml_type: int) -> np.ndarray:
    if ggml_type == GGML_Q8_0:
        return quantize_q8_0(w)
    if ggml_type == GGML_Q4_0:
        return quantize_q4_0(w)
    if ggml_type == GGML_F16:
        return to_f16_bytes(w)
    if ggml_type == GGML_F32:
        return to_f32_bytes(w)
    raise ValueError(f"no quantiser for ggml type {ggml_type}")
