"""ctypes front-end of the CPU oracle (oracle/nl_oracle.c).

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never from nanollama_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libnl_oracle.so")


class NloConfig(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("dim", C.c_int32), ("n_heads", C.c_int32), ("n_kv_heads", C.c_int32),
                ("head_dim", C.c_int32), ("interm", C.c_int32), ("vocab", C.c_int32), ("seq_len", C.c_int32),
                ("eps", C.c_float), ("rope_theta", C.c_float), ("qk_norm", C.c_int32), ("rope_conjugate", C.c_int32)]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "nl_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libnl_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        fp, vp, i32, u32, i64 = C.POINTER(C.c_float), C.c_void_p, C.c_int32, C.c_uint32, C.c_int64
        L.nlo_half2float.restype = C.c_float
        L.nlo_half2float.argtypes = [C.c_uint16]
        L.nlo_dequant.argtypes = [fp, vp, u32, i64]
        L.nlo_matmul.argtypes = [fp, vp, u32, fp, i32, i32]
        L.nlo_rmsnorm.argtypes = [fp, fp, i32, C.c_float]
        L.nlo_rmsnorm_bare.argtypes = [fp, i32, C.c_float]
        L.nlo_rmsnorm_into.argtypes = [fp, fp, fp, i32, C.c_float]
        L.nlo_softmax.argtypes = [fp, i32]
        L.nlo_silu.restype = C.c_float
        L.nlo_silu.argtypes = [C.c_float]
        L.nlo_argmax.argtypes = [fp, i32]
        L.nlo_embed_lookup.argtypes = [fp, vp, u32, i32, i32]
        L.nlo_create.restype = vp
        L.nlo_create.argtypes = [C.POINTER(NloConfig)]
        L.nlo_last_error.restype = C.c_char_p
        L.nlo_last_error.argtypes = [vp]
        L.nlo_set_tensor.argtypes = [vp, C.c_char_p, u32, vp]
        L.nlo_finalize.argtypes = [vp]
        L.nlo_destroy.argtypes = [vp]
        L.nlo_forward.argtypes = [vp, i32, i32]
        L.nlo_reset.argtypes = [vp]
        L.nlo_logits.restype = fp
        L.nlo_logits.argtypes = [vp]
        L.nlo_get_config.restype = C.POINTER(NloConfig)
        L.nlo_get_config.argtypes = [vp]
        L.nlo_state_buffer.restype = fp
        L.nlo_state_buffer.argtypes = [vp, C.c_char_p]
        L.nlo_generate_greedy.argtypes = [vp, C.POINTER(i32), i32, i32, i32, C.POINTER(i32), fp]
        L.nlo_set_threads.argtypes = [i32]
        L.nlo_set_gamma.argtypes = [vp, C.POINTER(i32), i32, fp]
        _lib = L
    return _lib


def _fp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def set_threads(n: int):
    lib().nlo_set_threads(int(n))


def half2float(h: int) -> float:
    return float(lib().nlo_half2float(h))


def dequant(data: np.ndarray, ggml_type: int, n: int) -> np.ndarray:
    data = np.ascontiguousarray(data)
    out = np.empty(n, dtype=np.float32)
    rc = lib().nlo_dequant(_fp(out), data.ctypes.data, ggml_type, n)
    if rc != 0:
        raise ValueError(f"unsupported type {ggml_type}")
    return out


def matmul(w: np.ndarray, ggml_type: int, x: np.ndarray, rows: int, cols: int) -> np.ndarray:
    w = np.ascontiguousarray(w)
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.zeros(rows, dtype=np.float32)
    rc = lib().nlo_matmul(_fp(out), w.ctypes.data, ggml_type, _fp(x), rows, cols)
    if rc != 0:
        raise ValueError(f"unsupported matmul type {ggml_type}")
    return out


def rmsnorm_into(x: np.ndarray, w: np.ndarray, eps: float) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    out = np.empty_like(x)
    lib().nlo_rmsnorm_into(_fp(out), _fp(x), _fp(w), len(x), C.c_float(eps))
    return out


def rmsnorm_bare(x: np.ndarray, eps: float) -> np.ndarray:
    out = np.array(x, dtype=np.float32, copy=True)
    lib().nlo_rmsnorm_bare(_fp(out), len(out), C.c_float(eps))
    return out


def softmax(x: np.ndarray) -> np.ndarray:
    out = np.array(x, dtype=np.float32, copy=True)
    lib().nlo_softmax(_fp(out), len(out))
    return out


def silu(x: float) -> float:
    return float(lib().nlo_silu(C.c_float(x)))


def argmax(logits: np.ndarray) -> int:
    logits = np.ascontiguousarray(logits, dtype=np.float32)
    return int(lib().nlo_argmax(_fp(logits), len(logits)))


def embed_lookup(data: np.ndarray, ggml_type: int, token: int, dim: int) -> np.ndarray:
    out = np.empty(dim, dtype=np.float32)
    lib().nlo_embed_lookup(_fp(out), np.ascontiguousarray(data).ctypes.data, ggml_type, token, dim)
    return out


class OracleModel:
    """CPU restatement of LlamaModel (go/model.go): Forward / Reset / Logits."""

    def __init__(self, gguf_file):
        """gguf_file: nanollama_amd.gguf.GGUFFile (tensor bytes are aliased, so it is kept alive here)."""
        L = lib()
        m = gguf_file.meta
        self.gguf = gguf_file
        cfg = NloConfig(m.num_layers, m.embed_dim, m.num_heads, m.num_kv_heads, m.head_dim, m.interm_size,
                        m.vocab_size, m.seq_len, m.rms_norm_eps, m.rope_theta, int(m.qk_norm), int(m.rope_conjugate))
        self.h = L.nlo_create(C.byref(cfg))
        self._keep = []
        for name in gguf_file.tensor_order:
            data, info = gguf_file.get_tensor(name)
            self._keep.append(data)
            rc = L.nlo_set_tensor(self.h, name.encode(), info.type, data.ctypes.data)
            if rc != 0:
                raise ValueError(L.nlo_last_error(self.h).decode())
        if L.nlo_finalize(self.h) != 0:
            raise ValueError(L.nlo_last_error(self.h).decode())
        c = L.nlo_get_config(self.h).contents
        self.vocab, self.seq_len, self.dim = c.vocab, c.seq_len, c.dim
        self.eos_id = m.eos_id

    def forward(self, token: int, pos: int) -> np.ndarray:
        lib().nlo_forward(self.h, token, pos)
        return self.logits()

    def logits(self) -> np.ndarray:
        return np.ctypeslib.as_array(lib().nlo_logits(self.h), shape=(self.vocab,))

    def state(self, which: str, n: int) -> np.ndarray:
        p = lib().nlo_state_buffer(self.h, which.encode())
        return np.ctypeslib.as_array(p, shape=(n,))

    def set_gamma(self, indices, values):
        idx = np.ascontiguousarray(indices, dtype=np.int32)
        vals = np.ascontiguousarray(values, dtype=np.float32)   # f16 -> f32 is exact, as half2float is
        lib().nlo_set_gamma(self.h, idx.ctypes.data_as(C.POINTER(C.c_int32)), int(idx.size), _fp(vals))

    def reset(self):
        lib().nlo_reset(self.h)

    def generate_greedy(self, prompt: List[int], max_tokens: int, want_logits: bool = False
                        ) -> Tuple[List[int], Optional[np.ndarray]]:
        p = np.asarray(prompt, dtype=np.int32)
        out = np.zeros(max_tokens, dtype=np.int32)
        lg = np.zeros((max_tokens, self.vocab), dtype=np.float32) if want_logits else None
        n = lib().nlo_generate_greedy(self.h, p.ctypes.data_as(C.POINTER(C.c_int32)), len(p), max_tokens, self.eos_id,
                                      out.ctypes.data_as(C.POINTER(C.c_int32)), _fp(lg) if want_logits else None)
        return [int(v) for v in out[:n]], (lg[:n] if want_logits else None)

    def close(self):
        if self.h:
            lib().nlo_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
