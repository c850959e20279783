"""GGUF v3 loader and writer -- host-side mirror of the reference's file surface.

Reader: mirrors ``go/gguf.go`` (LoadGGUF :289-414, parseMetadata :417-558,
GetTensor :561-574, block sizes :239-286) with the same names, defaults and
error behaviour, so host code above it reads like the Go engine's.

Writer: produces files byte-identical to the reference's
``scripts/export_gguf.py`` ``GGUFWriter.write`` (:266-311): KV pairs in
insertion order, tensor infos with reversed dims, per-tensor 32-byte alignment
inside the data section.  Pinned by tests/test_gguf_quant.py against fixtures written
by the reference's own writer.
"""
from __future__ import annotations

import io
import mmap
import os
import struct
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Tuple

import numpy as np

GGUF_MAGIC = 0x46554747  # go/gguf.go:25
GGUF_VERSION = 3
GGUF_ALIGNMENT = 32

# GGUF value types (go/gguf.go:28-41)
T_UINT8, T_INT8, T_UINT16, T_INT16, T_UINT32, T_INT32, T_FLOAT32, T_BOOL = range(8)
T_STRING, T_ARRAY, T_UINT64, T_INT64, T_FLOAT64 = 8, 9, 10, 11, 12

# GGML tensor types (go/gguf.go:43-57)
GGML_F32, GGML_F16, GGML_Q4_0, GGML_Q4_1 = 0, 1, 2, 3
GGML_Q5_0, GGML_Q5_1, GGML_Q8_0, GGML_Q8_1 = 6, 7, 8, 9
GGML_Q4_K, GGML_Q6_K = 12, 14

TYPE_NAMES = {GGML_F32: "f32", GGML_F16: "f16", GGML_Q4_0: "q4_0", GGML_Q5_0: "q5_0",
              GGML_Q8_0: "q8_0", GGML_Q4_K: "q4_k", GGML_Q6_K: "q6_k"}

_SCALAR_FMT = {T_UINT8: "<B", T_INT8: "<b", T_UINT16: "<H", T_INT16: "<h", T_UINT32: "<I",
               T_INT32: "<i", T_FLOAT32: "<f", T_UINT64: "<Q", T_INT64: "<q", T_FLOAT64: "<d"}


class GGUFError(Exception):
    pass


def ggml_block_size(t: int) -> int:
    """Bytes per block (go/gguf.go:239-260); 0 for unsupported types."""
    return {GGML_F32: 4, GGML_F16: 2, GGML_Q4_0: 18, GGML_Q4_1: 20, GGML_Q8_0: 34,
            GGML_Q5_0: 22, GGML_Q6_K: 210, GGML_Q4_K: 144}.get(t, 0)


def ggml_block_elements(t: int) -> int:
    """Elements per block (go/gguf.go:263-272)."""
    if t in (GGML_F32, GGML_F16):
        return 1
    if t in (GGML_Q4_K, GGML_Q6_K):
        return 256
    return 32


@dataclass
class GGUFTensorInfo:
    name: str
    ndims: int
    dims: Tuple[int, ...]  # GGML order: innermost first
    type: int
    offset: int

    @property
    def nel(self) -> int:
        n = 1
        for d in self.dims[: self.ndims]:
            n *= d
        return n

    def nbytes(self) -> int:
        """tensorBytes go/gguf.go:275-286."""
        be = ggml_block_elements(self.type)
        return (self.nel // be) * ggml_block_size(self.type)


@dataclass
class GGUFMetadata:
    """go/gguf.go:60-90 with the defaults of parseMetadata :417-424."""
    num_layers: int = 0
    embed_dim: int = 0
    num_heads: int = 0
    num_kv_heads: int = 0
    head_dim: int = 0
    vocab_size: int = 0
    seq_len: int = 0
    interm_size: int = 0
    rms_norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    qk_norm: bool = False
    rope_conjugate: bool = False
    token_list: List[str] = field(default_factory=list)
    token_scores: List[float] = field(default_factory=list)
    token_types: List[int] = field(default_factory=list)
    token_merges: List[str] = field(default_factory=list)
    tokenizer_model: str = "llama"
    bos_id: int = 1
    eos_id: int = 2
    add_space_prefix: bool = True
    kv: Dict[str, Any] = field(default_factory=dict)
    kv_types: Dict[str, int] = field(default_factory=dict)


class _Reader:
    def __init__(self, buf):
        self.buf = buf
        self.pos = 0

    def take(self, n: int) -> bytes:
        if self.pos + n > len(self.buf):
            raise GGUFError("unexpected EOF")
        b = self.buf[self.pos:self.pos + n]
        self.pos += n
        return b

    def scalar(self, fmt: str):
        return struct.unpack(fmt, self.take(struct.calcsize(fmt)))[0]

    def string(self) -> str:
        n = self.scalar("<Q")
        if n > 1 << 24:  # go/gguf.go:114
            raise GGUFError(f"string too long: {n}")
        return bytes(self.take(n)).decode("utf-8", errors="replace")

    def value(self, vtype: int):
        if vtype in _SCALAR_FMT:
            return self.scalar(_SCALAR_FMT[vtype])
        if vtype == T_BOOL:
            return self.scalar("<B") != 0
        if vtype == T_STRING:
            return self.string()
        if vtype == T_ARRAY:
            et = self.scalar("<I")
            cnt = self.scalar("<Q")
            if cnt > 1 << 24:  # go/gguf.go:181
                raise GGUFError(f"array too large: {cnt}")
            if et in _SCALAR_FMT:  # fast path
                f = _SCALAR_FMT[et]
                sz = struct.calcsize(f)
                raw = self.take(sz * cnt)
                return list(struct.unpack("<%d%s" % (cnt, f[1]), raw))
            return [self.value(et) for _ in range(cnt)]
        raise GGUFError(f"unknown GGUF type: {vtype}")


def _to_int(v) -> int:
    """toInt go/gguf.go:199-220 (non-integers -> 0)."""
    if isinstance(v, bool) or not isinstance(v, int):
        return 0
    return int(v)


def _to_f32(v) -> float:
    """toFloat32 go/gguf.go:223-236."""
    if isinstance(v, bool):
        return 0.0
    if isinstance(v, (int, float)):
        return float(np.float32(v))
    return 0.0


def parse_metadata(kv: Dict[str, Any], kv_types: Optional[Dict[str, int]] = None) -> GGUFMetadata:
    """parseMetadata go/gguf.go:417-558."""
    m = GGUFMetadata(kv=kv, kv_types=kv_types or {})
    m.rms_norm_eps = float(np.float32(1e-5))
    arch = kv.get("general.architecture", "llama")
    if not isinstance(arch, str):
        arch = "llama"

    def geti(key):
        return _to_int(kv[key]) if key in kv else None

    for attr, key in (("num_layers", ".block_count"), ("embed_dim", ".embedding_length"),
                      ("num_heads", ".attention.head_count"), ("num_kv_heads", ".attention.head_count_kv"),
                      ("interm_size", ".feed_forward_length"), ("seq_len", ".context_length")):
        v = geti(arch + key)
        if v is not None:
            setattr(m, attr, v)
    if arch + ".attention.layer_norm_rms_epsilon" in kv:
        m.rms_norm_eps = _to_f32(kv[arch + ".attention.layer_norm_rms_epsilon"])
    if arch + ".rope.freq_base" in kv:
        m.rope_theta = _to_f32(kv[arch + ".rope.freq_base"])
    if m.num_heads > 0 and m.embed_dim > 0:
        m.head_dim = m.embed_dim // m.num_heads  # key_length is ignored (:461-463)
    if m.num_kv_heads == 0:
        m.num_kv_heads = m.num_heads
    if isinstance(kv.get("nanollama.qk_norm"), bool):
        m.qk_norm = kv["nanollama.qk_norm"]
    if isinstance(kv.get("nanollama.rope_conjugate"), bool):
        m.rope_conjugate = kv["nanollama.rope_conjugate"]
    if isinstance(kv.get("tokenizer.ggml.model"), str):
        m.tokenizer_model = kv["tokenizer.ggml.model"]
    toks = kv.get("tokenizer.ggml.tokens")
    if isinstance(toks, list):
        m.token_list = [t if isinstance(t, str) else "" for t in toks]
        m.vocab_size = len(m.token_list)  # the ONLY source of VocabSize (:489-498)
    sc = kv.get("tokenizer.ggml.scores")
    if isinstance(sc, list):
        m.token_scores = [_to_f32(s) for s in sc]
    tt = kv.get("tokenizer.ggml.token_type")
    if isinstance(tt, list):
        m.token_types = [_to_int(t) for t in tt]
    if "tokenizer.ggml.bos_token_id" in kv:
        m.bos_id = _to_int(kv["tokenizer.ggml.bos_token_id"])
    if "tokenizer.ggml.eos_token_id" in kv:
        m.eos_id = _to_int(kv["tokenizer.ggml.eos_token_id"])
    mg = kv.get("tokenizer.ggml.merges")
    if isinstance(mg, list):
        m.token_merges = [s if isinstance(s, str) else "" for s in mg]
    v = kv.get("tokenizer.ggml.add_space_prefix")
    if isinstance(v, bool):
        m.add_space_prefix = v
    elif isinstance(v, int) and (kv_types or {}).get("tokenizer.ggml.add_space_prefix") in (T_UINT8, T_UINT32):
        m.add_space_prefix = v != 0
    return m


class GGUFFile:
    """Parsed GGUF file (go/gguf.go:101-107).  TensorData is a read-only
    memory map of the data section instead of a heap copy."""

    def __init__(self, meta: GGUFMetadata, tensors: Dict[str, GGUFTensorInfo], tensor_data, data_offset: int,
                 version: int, order: List[str]):
        self.meta = meta
        self.tensors = tensors
        self.tensor_data = tensor_data
        self.data_offset = data_offset
        self.version = version
        self.tensor_order = order

    def get_tensor(self, name: str) -> Tuple[np.ndarray, GGUFTensorInfo]:
        """GetTensor go/gguf.go:561-574: raw bytes (uint8 view) + info."""
        info = self.tensors.get(name)
        if info is None:
            raise KeyError(f"tensor not found: {name}")
        size = info.nbytes()
        start, end = info.offset, info.offset + size
        if end > len(self.tensor_data):
            raise GGUFError(f"tensor {name} out of bounds: {start} + {size} > {len(self.tensor_data)}")
        return self.tensor_data[start:end], info


def load_gguf(path: str, verbose: bool = False) -> GGUFFile:
    """LoadGGUF go/gguf.go:289-414."""
    try:
        f = open(path, "rb")
    except OSError as e:
        raise GGUFError(f"open GGUF: {e}") from e
    with f:
        size = os.fstat(f.fileno()).st_size
        if size < 24:
            raise GGUFError("read magic: unexpected EOF")
        mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
    r = _Reader(memoryview(mm))
    magic = r.scalar("<I")
    if magic != GGUF_MAGIC:
        raise GGUFError("bad magic: 0x%08X (expected 0x%08X)" % (magic, GGUF_MAGIC))
    version = r.scalar("<I")
    if version < 2 or version > 3:
        raise GGUFError(f"unsupported GGUF version: {version}")
    tensor_count = r.scalar("<Q")
    kv_count = r.scalar("<Q")
    if verbose:
        print(f"[gguf] version={version} tensors={tensor_count} metadata={kv_count}")
    kv: Dict[str, Any] = {}
    kv_types: Dict[str, int] = {}
    for _ in range(kv_count):
        key = r.string()
        vtype = r.scalar("<I")
        kv[key] = r.value(vtype)
        kv_types[key] = vtype
    tensors: Dict[str, GGUFTensorInfo] = {}
    order: List[str] = []
    for _ in range(tensor_count):
        name = r.string()
        ndims = r.scalar("<I")
        if ndims > 4:
            raise GGUFError(f"tensor {name}: ndims {ndims} > 4")
        dims = tuple(r.scalar("<Q") for _ in range(ndims))
        ttype = r.scalar("<I")
        offset = r.scalar("<Q")
        tensors[name] = GGUFTensorInfo(name, ndims, dims, ttype, offset)
        order.append(name)
    header_end = r.pos
    data_offset = ((header_end + GGUF_ALIGNMENT - 1) // GGUF_ALIGNMENT) * GGUF_ALIGNMENT
    data_size = size - data_offset
    if data_size <= 0:
        raise GGUFError(f"no tensor data (dataOffset={data_offset}, fileSize={size})")
    if verbose:
        print("[gguf] data offset=%d size=%.1f MB" % (data_offset, data_size / 1024 / 1024))
    data = np.frombuffer(mm, dtype=np.uint8, count=data_size, offset=data_offset)
    meta = parse_metadata(kv, kv_types)
    if verbose:
        print(f"[gguf] arch={kv.get('general.architecture', 'llama')} layers={meta.num_layers} dim={meta.embed_dim} "
              f"heads={meta.num_heads} kv_heads={meta.num_kv_heads} head_dim={meta.head_dim}")
        print("[gguf] vocab=%d seq_len=%d ffn=%d rope_theta=%.1f tokenizer=%s" % (
            meta.vocab_size, meta.seq_len, meta.interm_size, meta.rope_theta, meta.tokenizer_model))
    return GGUFFile(meta, tensors, data, data_offset, version, order)


# --------------------------------------------------------------------- writer

class GGUFWriter:
    """Byte-compatible with scripts/export_gguf.py GGUFWriter (:164-311), but
    streams tensor payloads (bytes / numpy arrays) instead of holding
    Python-level per-block loops."""

    def __init__(self, path: str):
        self.path = path
        self.kv_pairs: List[Tuple[str, int, Any]] = []
        self.tensors: List[Tuple[str, Any, int, Tuple[int, ...]]] = []

    def add_uint32(self, k, v): self.kv_pairs.append((k, T_UINT32, v))
    def add_int32(self, k, v): self.kv_pairs.append((k, T_INT32, v))
    def add_float32(self, k, v): self.kv_pairs.append((k, T_FLOAT32, v))
    def add_bool(self, k, v): self.kv_pairs.append((k, T_BOOL, v))
    def add_string(self, k, v): self.kv_pairs.append((k, T_STRING, v))
    def add_string_array(self, k, v): self.kv_pairs.append((k, T_ARRAY, (T_STRING, v)))
    def add_float32_array(self, k, v): self.kv_pairs.append((k, T_ARRAY, (T_FLOAT32, v)))
    def add_int32_array(self, k, v): self.kv_pairs.append((k, T_ARRAY, (T_INT32, v)))

    def add_tensor_raw(self, name: str, raw, ggml_type: int, shape: Tuple[int, ...]):
        """raw: bytes-like or a C-contiguous uint8 numpy array; shape in row-major (PyTorch) order."""
        self.tensors.append((name, raw, ggml_type, tuple(int(s) for s in shape)))

    @staticmethod
    def _wstr(f, s: str):
        e = s.encode("utf-8")
        f.write(struct.pack("<Q", len(e)))
        f.write(e)

    def _wkv(self, f, key, vtype, value):
        self._wstr(f, key)
        f.write(struct.pack("<I", vtype))
        if vtype in _SCALAR_FMT:
            f.write(struct.pack(_SCALAR_FMT[vtype], value))
        elif vtype == T_BOOL:
            f.write(struct.pack("<B", 1 if value else 0))
        elif vtype == T_STRING:
            self._wstr(f, value)
        elif vtype == T_ARRAY:
            et, elems = value
            f.write(struct.pack("<I", et))
            f.write(struct.pack("<Q", len(elems)))
            if et == T_STRING:
                buf = io.BytesIO()
                for e in elems:
                    b = e.encode("utf-8")
                    buf.write(struct.pack("<Q", len(b)))
                    buf.write(b)
                f.write(buf.getvalue())
            elif et in _SCALAR_FMT:
                f.write(struct.pack("<%d%s" % (len(elems), _SCALAR_FMT[et][1]), *elems))
        else:
            raise GGUFError(f"cannot write value type {vtype}")

    def write(self):
        with open(self.path, "wb") as f:
            f.write(struct.pack("<I", GGUF_MAGIC))
            f.write(struct.pack("<I", GGUF_VERSION))
            f.write(struct.pack("<Q", len(self.tensors)))
            f.write(struct.pack("<Q", len(self.kv_pairs)))
            for key, vtype, value in self.kv_pairs:
                self._wkv(f, key, vtype, value)
            off = 0
            offsets = []
            for i, (_, raw, _, _) in enumerate(self.tensors):
                if i > 0:
                    off = ((off + GGUF_ALIGNMENT - 1) // GGUF_ALIGNMENT) * GGUF_ALIGNMENT
                offsets.append(off)
                off += len(memoryview(raw).cast("B")) if not isinstance(raw, np.ndarray) else raw.nbytes
            for i, (name, raw, t, shape) in enumerate(self.tensors):
                self._wstr(f, name)
                f.write(struct.pack("<I", len(shape)))
                for d in reversed(shape):
                    f.write(struct.pack("<Q", d))
                f.write(struct.pack("<I", t))
                f.write(struct.pack("<Q", offsets[i]))
            self._align(f)
            for _, raw, _, _ in self.tensors:
                self._align(f)
                if isinstance(raw, np.ndarray):
                    f.write(np.ascontiguousarray(raw).view(np.uint8).reshape(-1).data)
                else:
                    f.write(raw)

    @staticmethod
    def _align(f):
        pad = (-f.tell()) % GGUF_ALIGNMENT
        if pad:
            f.write(b"\x00" * pad)
