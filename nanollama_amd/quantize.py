"""GGUF re-quantiser -- the job of scripts/quantize_gguf.py (F16/F32 GGUF -> Q8_0 GGUF, 1-D tensors to F32),
vectorised, plus Q4_0 which the reference CLI cannot produce.

    python -m nanollama_amd.quantize in-f16.gguf out-q8_0.gguf [--type q8_0|q4_0]

--type q8_0 is byte-identical to the reference tool (pinned by tests/test_quantize_cli.py):
  * its Q8_0 rule is NOT the exporter's: scale = amax/127 and q = round(v * (1/scale)) are evaluated in Python
    floats (float64) with banker's rounding, the scale is then packed to fp16 (scripts/quantize_gguf.py:183-215);
  * it re-serialises metadata arrays with an element type inferred from the values -- an int32 array without
    negative entries comes back as uint32 (:400-440).  We reproduce that so the files match.
--type q4_0 uses the exporter's Q4_0 rule (scripts/export_gguf.py:85-121) on the float32 values.
"""
from __future__ import annotations

import argparse
import struct
import sys
from typing import List

import numpy as np

from . import quant
from .gguf import (GGML_F16, GGML_F32, GGML_Q4_0, GGML_Q8_0, GGUF_ALIGNMENT, GGUF_MAGIC, T_ARRAY, T_BOOL, T_STRING,
                   _SCALAR_FMT, load_gguf)


def quantize_q8_0_pyfloat(values_f32: np.ndarray) -> np.ndarray:
    """scripts/quantize_gguf.py:183-215 in float64 (what CPython's floats are)."""
    t = np.ascontiguousarray(values_f32, dtype=np.float32).astype(np.float64).reshape(-1, 32)
    amax = np.abs(t).max(axis=1)
    scale = np.where(amax == 0, 1.0, amax / 127.0)
    q = np.clip(np.rint(t * (1.0 / scale)[:, None]), -128, 127).astype(np.int8)
    out = np.empty((t.shape[0], 34), dtype=np.uint8)
    out[:, 0:2] = scale.astype(np.float16).view(np.uint8).reshape(-1, 2)
    out[:, 2:] = q.view(np.uint8)
    return out.reshape(-1)


def _wstr(f, s: str):
    b = s.encode("utf-8")
    f.write(struct.pack("<Q", len(b)))
    f.write(b)


def _write_array_inferred(f, arr: List):
    """scripts/quantize_gguf.py:400-440: element type inferred from the content."""
    if not arr:
        f.write(struct.pack("<IQ", 4, 0))
        return
    first = arr[0]
    if isinstance(first, str):
        et = 8
    elif isinstance(first, float):
        et = 6
    elif isinstance(first, bool):
        et = 4
    elif isinstance(first, int):
        et = 5 if any(v < 0 for v in arr) else 4
    else:
        et = 4
    f.write(struct.pack("<IQ", et, len(arr)))
    if et == 8:
        for e in arr:
            _wstr(f, e)
    else:
        fmt = {6: "f", 5: "i", 4: "I"}[et]
        f.write(struct.pack("<%d%s" % (len(arr), fmt), *[int(e) if et != 6 else e for e in arr]))


def requantize(src: str, dst: str, wtype: str = "q8_0", verbose: bool = True) -> None:
    g = load_gguf(src)
    target = {"q8_0": GGML_Q8_0, "q4_0": GGML_Q4_0}[wtype]
    new = []
    for name in g.tensor_order:
        data, info = g.get_tensor(name)
        nel = info.nel
        if info.type == GGML_F16:
            vals = data.view(np.float16).astype(np.float32)
        elif info.type == GGML_F32:
            vals = data.view(np.float32)
        else:
            raise ValueError(f"Cannot quantize type {info.type}")
        if info.ndims == 1 or nel % 32 != 0:
            new.append((name, np.ascontiguousarray(vals, dtype=np.float32).view(np.uint8), GGML_F32, info.dims))
            kind = "F32"
        else:
            raw = quantize_q8_0_pyfloat(vals) if target == GGML_Q8_0 else quant.quantize_q4_0(vals)
            new.append((name, raw, target, info.dims))
            kind = wtype.upper()
        if verbose:
            print(f"  {name:40s} [{'x'.join(str(d) for d in reversed(info.dims))}] -> {kind}")
    with open(dst, "wb") as f:
        f.write(struct.pack("<IIQQ", GGUF_MAGIC, g.version, len(new), len(g.meta.kv)))
        for key, value in g.meta.kv.items():
            vtype = g.meta.kv_types[key]
            _wstr(f, key)
            f.write(struct.pack("<I", vtype))
            if vtype == T_ARRAY:
                _write_array_inferred(f, value)
            elif vtype == T_STRING:
                _wstr(f, value)
            elif vtype == T_BOOL:
                f.write(struct.pack("<B", 1 if value else 0))
            else:
                f.write(struct.pack(_SCALAR_FMT[vtype], value))
        off, offsets = 0, []
        for i, (_, raw, _, _) in enumerate(new):
            if i > 0:
                off = (off + GGUF_ALIGNMENT - 1) // GGUF_ALIGNMENT * GGUF_ALIGNMENT
            offsets.append(off)
            off += raw.nbytes
        for i, (name, raw, t, dims) in enumerate(new):
            _wstr(f, name)
            f.write(struct.pack("<I", len(dims)))
            for d in dims:   # already innermost-first
                f.write(struct.pack("<Q", d))
            f.write(struct.pack("<IQ", t, offsets[i]))
        f.write(b"\0" * ((-f.tell()) % GGUF_ALIGNMENT))
        for _, raw, _, _ in new:
            f.write(b"\0" * ((-f.tell()) % GGUF_ALIGNMENT))
            f.write(raw.data)


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="nanollama_amd.quantize")
    ap.add_argument("input")
    ap.add_argument("output")
    ap.add_argument("--type", default="q8_0", choices=["q8_0", "q4_0"])
    a = ap.parse_args(argv)
    requantize(a.input, a.output, a.type)
    return 0


if __name__ == "__main__":
    sys.exit(main())
