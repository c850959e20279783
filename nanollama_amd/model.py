"""Host-side mirror of go/model.go's LlamaModel over the HIP C ABI.

Same names and call pattern as the reference so call sites read alike:

    gguf  = load_gguf(path)                 # go/main.go:51
    model = load_llama_model(gguf)          # go/main.go:63  LoadLlamaModel
    model.reset()                           # go/main.go:156 Reset
    model.forward(tok, pos)                 # go/main.go:161 Forward
    model.state.logits                      # go/main.go:174 State.Logits (host float32, mutable)

All arithmetic happens in libnanollama_hip.so; this file only moves bytes.
"""
from __future__ import annotations

import ctypes as C
import re
from dataclasses import dataclass
from typing import List, Optional

import numpy as np

from . import _lib
from .gguf import GGUFFile


@dataclass
class LlamaConfig:
    """go/model.go:27-42"""
    num_layers: int
    embed_dim: int
    num_heads: int
    num_kv_heads: int
    head_dim: int
    vocab_size: int
    seq_len: int
    interm_size: int
    rms_norm_eps: float
    rope_theta: float
    qk_norm: bool = False
    rope_conjugate: bool = False


# the tensors loadWeights reads (go/model.go:177-265)
_KNOWN_TENSOR = re.compile(r"^(token_embd\.weight|output_norm\.weight|output\.weight|blk\.\d+\.("
                           r"(attn_norm|ffn_norm|attn_q|attn_k|attn_v|attn_output|ffn_gate|ffn_up|ffn_down)\.weight|"
                           r"attn_(q|k|v|output)\.bias))$")


class LlamaState:
    """The slice of go/model.go:93-118 the host touches: Logits (and Pos)."""

    def __init__(self, vocab: int):
        self.logits = np.zeros(vocab, dtype=np.float32)
        self.pos = 0


class LlamaModel:
    def __init__(self, config: LlamaConfig, handle, max_streams: int):
        self.config = config
        self.state = LlamaState(config.vocab_size)
        self.gamma = None  # go/model.go:23 (gamma injection is out of scope)
        self._h = handle
        self.max_streams = max_streams

    # --- Forward go/model.go:490 ---
    def forward(self, token: int, pos: int, stream: int = 0) -> None:
        L = _lib.lib()
        _lib.check(self._h, L.nl_forward(self._h, stream, int(token), int(pos),
                                         self.state.logits.ctypes.data_as(C.POINTER(C.c_float))))

    def forward_argmax(self, token: int, pos: int, stream: int = 0) -> int:
        out = C.c_int(0)
        _lib.check(self._h, _lib.lib().nl_forward_argmax(self._h, stream, int(token), int(pos), C.byref(out)))
        return out.value

    def decode_greedy(self, token: int, pos: int, n_steps: int, stream: int = 0) -> List[int]:
        ids = (C.c_int * max(n_steps, 1))()
        done = C.c_int(0)
        _lib.check(self._h, _lib.lib().nl_decode_greedy(self._h, stream, int(token), int(pos), int(n_steps), ids,
                                                        C.byref(done)))
        return [int(ids[i]) for i in range(done.value)]

    def sample_decode(self, pos: int, n_steps: int, temperature: float, top_p: float, top_k: int, rep_penalty: float,
                      rep_window: int, uniforms, recent: List[int], stream: int = 0):
        """The sampling loop of Generate (go/main.go:173-219) on the device: from the logits the last forward /
        prefill left there, n_steps x { penalty, sample, Forward(sampled, pos++) }.  uniforms: one float32 per
        step (the host generator's Float32 stream).  Returns (ids, recent window afterwards)."""
        p = _lib.NlSampleParams(float(temperature), float(top_p), int(top_k), float(rep_penalty), int(rep_window))
        u = np.ascontiguousarray(uniforms, dtype=np.float32)
        if u.size < n_steps:
            raise ValueError("one uniform per step is required")
        rec = (C.c_int * max(int(rep_window), 1))(*[int(t) for t in recent])
        nrec = C.c_int(len(recent))
        ids = (C.c_int * max(n_steps, 1))()
        done = C.c_int(0)
        _lib.check(self._h, _lib.lib().nl_sample_decode(self._h, stream, int(pos), int(n_steps), C.byref(p),
                                                        u.ctypes.data_as(C.POINTER(C.c_float)), rec, C.byref(nrec), ids,
                                                        C.byref(done)))
        return [int(ids[i]) for i in range(done.value)], [int(rec[i]) for i in range(nrec.value)]

    def prefill(self, tokens: List[int], pos0: int = 0, stream: int = 0, want_logits: bool = True) -> None:
        """The prompt loop of Generate (go/main.go:160-166) queued on the device in one call."""
        arr = (C.c_int * len(tokens))(*[int(t) for t in tokens])
        out = self.state.logits.ctypes.data_as(C.POINTER(C.c_float)) if want_logits else None
        _lib.check(self._h, _lib.lib().nl_prefill(self._h, stream, arr, len(tokens), int(pos0), out))

    def forward_batch(self, streams: List[int], tokens: List[int], pos: List[int], want_logits: bool = False):
        n = len(streams)
        ia = lambda v: (C.c_int * n)(*[int(x) for x in v])
        ids = (C.c_int * n)()
        lg = np.zeros((n, self.config.vocab_size), dtype=np.float32) if want_logits else None
        _lib.check(self._h, _lib.lib().nl_forward_batch(
            self._h, ia(streams), ia(tokens), ia(pos), n,
            lg.ctypes.data_as(C.POINTER(C.c_float)) if want_logits else None, ids))
        return [int(ids[i]) for i in range(n)], lg

    def set_gamma(self, indices, values) -> None:
        """model.Gamma = gamma (go/main.go:79): indices int32 [n], values [n, dim] float32 or float16."""
        idx = np.ascontiguousarray(indices, dtype=np.int32)
        vals = np.ascontiguousarray(values)
        if vals.dtype not in (np.float32, np.float16):
            vals = vals.astype(np.float32)
        if idx.size and (vals.ndim != 2 or vals.shape != (idx.size, self.config.embed_dim)):
            raise ValueError(f"gamma embed_dim {vals.shape} != model dim {self.config.embed_dim}")  # go/main.go:75-77
        _lib.check(self._h, _lib.lib().nl_set_gamma(self._h, idx.ctypes.data_as(C.POINTER(C.c_int32)), int(idx.size),
                                                    vals.ctypes.data, int(vals.dtype == np.float16)))
        self.gamma = (idx, vals) if idx.size else None

    # --- Reset go/model.go:623 ---
    def reset(self, stream: int = 0) -> None:
        _lib.check(self._h, _lib.lib().nl_reset(self._h, stream))
        self.state.pos = 0

    # --- measurement / introspection ---
    def synchronize(self):
        _lib.check(self._h, _lib.lib().nl_synchronize(self._h))

    def timer_start(self):
        _lib.check(self._h, _lib.lib().nl_timer_start(self._h))

    def timer_stop(self) -> float:
        ms = C.c_float(0)
        _lib.check(self._h, _lib.lib().nl_timer_stop(self._h, C.byref(ms)))
        return ms.value

    def profile_forward(self, token: int, pos: int, iters: int = 10, stream: int = 0):
        L = _lib.lib()
        ms = (C.c_float * _lib.NL_NUM_KINDS)()
        calls = (C.c_int * _lib.NL_NUM_KINDS)()
        _lib.check(self._h, L.nl_profile_forward(self._h, stream, token, pos, iters, ms, calls))
        return {L.nl_kernel_kind_name(k).decode(): (ms[k], calls[k]) for k in range(_lib.NL_NUM_KINDS)}

    def memory_usage(self):
        w, kv, st = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        _lib.check(self._h, _lib.lib().nl_memory_usage(self._h, C.byref(w), C.byref(kv), C.byref(st)))
        return {"weights": w.value, "kv_cache": kv.value, "state": st.value}

    def p2p_info(self):
        """Tensor-parallel transport of this handle: push all-reduce set up? receive area uncached?"""
        on, unc = C.c_int(0), C.c_int(0)
        _lib.check(self._h, _lib.lib().nl_p2p_info(self._h, C.byref(on), C.byref(unc)))
        return {"push_allreduce": bool(on.value), "uncached_receive_area": bool(unc.value)}

    def plan_info(self):
        """nl_plan_info: which launch plans the handle holds (fused mode, its position limit, launches per token)."""
        v = [C.c_int(0) for _ in range(4)]
        _lib.check(self._h, _lib.lib().nl_plan_info(self._h, *[C.byref(x) for x in v]))
        return {"fused_mode": v[0].value, "fused_max_pos": v[1].value, "launches_fused": v[2].value, "launches_general": v[3].value}

    def persist_info(self):
        """nl_persist_info: is the one-launch-per-chunk persistent decode (nl_persist.h) serving greedy chains of this handle."""
        ready, max_pos = C.c_int(0), C.c_int(0)
        launches, tokens = C.c_longlong(0), C.c_longlong(0)
        _lib.check(self._h, _lib.lib().nl_persist_info(self._h, C.byref(ready), C.byref(max_pos), C.byref(launches), C.byref(tokens)))
        return {"ready": bool(ready.value), "max_pos": max_pos.value, "launches": launches.value, "tokens": tokens.value}

    def last_error(self) -> str:
        """nl_last_error: the message of the last failed call -- or the one-time note of a call that succeeded after
        retiring the fused launch plan (a cluster exchange timed out and the step was redone on the general plan)."""
        return (_lib.lib().nl_last_error(self._h) or b"").decode()

    def debug_read(self, which: str, n: int, stream: int = 0) -> np.ndarray:
        out = np.zeros(n, dtype=np.float32)
        got = _lib.lib().nl_debug_read(self._h, which.encode(), stream, out.ctypes.data_as(C.POINTER(C.c_float)), n)
        if got < 0:
            _lib.check(self._h, int(got))
        return out[:got]

    def close(self):
        if self._h:
            _lib.lib().nl_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def load_llama_model(gguf: GGUFFile, device: int = 0, max_streams: int = 1, tp_rank: int = 0, tp_size: int = 1,
                     comm_id: Optional[bytes] = None, flags: int = 0, verbose: bool = False,
                     p2p_allgather=None, p2p_loopback: bool = False, devices: Optional[List[int]] = None) -> LlamaModel:
    """LoadLlamaModel go/model.go:121-174: config from GGUF metadata, every
    tensor handed to the device library, state allocated there.

    Tensor-parallel runs (one process per GPU, tp_size 2/4/8): `p2p_allgather(bytes) -> [bytes per rank]` (e.g.
    Rendezvous.allgather_bytes) selects the push all-reduce over xGMI -- it carries the hipIpc handles of the ranks'
    receive areas; `comm_id` selects RCCL instead.

    `devices` = [d0, d1, ...] (2, 4 or 8 entries): ONE process, the model sharded tensor-parallel over those GPUs behind
    one handle (nl_create_group) -- what `--gpus N` of the CLI / server uses; the returned model is used like any other."""
    L = _lib.lib()
    m = gguf.meta
    head_dim = m.head_dim
    if head_dim == 0 and m.num_heads > 0:
        head_dim = m.embed_dim // m.num_heads
    seq_len = m.seq_len
    if seq_len > 2048:  # go/model.go:145-148
        if verbose:
            print(f"[model] capping seq_len from {seq_len} to 2048")
        seq_len = 2048
    cfg = _lib.NlConfig(m.num_layers, m.embed_dim, m.num_heads, m.num_kv_heads, head_dim, m.interm_size, m.vocab_size,
                        seq_len, m.rms_norm_eps, m.rope_theta, int(m.qk_norm), int(m.rope_conjugate), max_streams,
                        device, tp_rank, tp_size, flags)
    h = C.c_void_p()
    if devices is not None:
        if tp_size != 1 or comm_id is not None or p2p_allgather is not None or p2p_loopback:
            raise ValueError("devices=[...] is the one-process group: no tp_rank / tp_size / communicator arguments")
        ids = (C.c_int * len(devices))(*[int(d) for d in devices])
        rc = L.nl_create_group(C.byref(cfg), ids, len(devices), C.byref(h))
    else:
        rc = L.nl_create(C.byref(cfg), C.byref(h))
    if rc != 0:
        raise _lib.NlError(rc, (L.nl_last_error(None) or b"").decode())
    try:
        if devices is not None:
            pass                                  # the group wires its ranks itself
        elif tp_size > 1 and p2p_loopback:
            # measurement only (bench.py --shard-of): this rank alone, tensor-parallel plan, own area in place of the peers'
            _lib.check(h, L.nl_p2p_loopback(h))
        elif tp_size > 1 and p2p_allgather is not None and not (flags & _lib.NL_FLAG_LOCAL_GROUP):
            mine = C.create_string_buffer(_lib.NL_P2P_HANDLE_BYTES)
            _lib.check(h, L.nl_p2p_export(h, mine))
            handles = p2p_allgather(mine.raw)
            if len(handles) != tp_size or any(len(x) != _lib.NL_P2P_HANDLE_BYTES for x in handles):
                raise ValueError("p2p_allgather must return one handle per rank, in rank order")
            _lib.check(h, L.nl_p2p_import(h, C.c_char_p(b"".join(handles))))
        elif (tp_size > 1 or comm_id is not None) and not (flags & _lib.NL_FLAG_LOCAL_GROUP):
            if comm_id is None or len(comm_id) != _lib.NL_COMM_ID_BYTES:
                raise ValueError("tp_size > 1 needs the communicator id from nl_comm_get_unique_id")
            _lib.check(h, L.nl_comm_init(h, C.c_char_p(comm_id)))
        for name in gguf.tensor_order:
            if not _KNOWN_TENSOR.match(name):
                # the Go loader fetches tensors by name and never looks at the rest (go/model.go:177-265): files that
                # carry extras such as rope_freqs.weight load there, so they are skipped here
                if verbose:
                    print(f"[model] skipping tensor {name}")
                continue
            data, info = gguf.get_tensor(name)
            if info.ndims == 1:
                rows, cols = 1, info.dims[0]
            else:
                cols, rows = info.dims[0], info.dims[1]
            data = np.ascontiguousarray(data)
            _lib.check(h, L.nl_upload_tensor(h, name.encode(), info.type, data.ctypes.data, data.nbytes, rows, cols))
        _lib.check(h, L.nl_finalize(h))
    except Exception:
        L.nl_destroy(h)
        raise
    config = LlamaConfig(m.num_layers, m.embed_dim, m.num_heads, m.num_kv_heads, head_dim, m.vocab_size, seq_len,
                         m.interm_size, m.rms_norm_eps, m.rope_theta, m.qk_norm, m.rope_conjugate)
    if verbose:
        print(f"[model] loaded: {config.num_layers} layers, {config.embed_dim} dim, {config.num_heads} heads, "
              f"{config.num_kv_heads} kv_heads, {config.vocab_size} vocab, bias=False")
    return LlamaModel(config, h, max_streams)


class LocalTPGroup:
    """n tensor-parallel shards of one model in THIS process (nl_group_forward): the same sharding
    arithmetic as the one-process-per-GPU RCCL path, with the all-reduce / all-gather seams done
    in-process so it can be checked on a single GPU."""

    def __init__(self, gguf: GGUFFile, n: int, device: int = 0, fused: bool = False):
        # fused: short contexts step the two-launches-per-layer plan of a push-group rank (nl_tp.h) where the shapes allow
        # it; the group then adds the shards' partial vectors itself, in rank order
        flags = _lib.NL_FLAG_LOCAL_GROUP | (_lib.NL_FLAG_GROUP_FUSED if fused else 0)
        self.shards = [load_llama_model(gguf, device=device, tp_rank=r, tp_size=n, flags=flags) for r in range(n)]
        self.n = n
        self.logits = np.zeros(self.shards[0].config.vocab_size, dtype=np.float32)

    def forward(self, token: int, pos: int, stream: int = 0) -> np.ndarray:
        hs = (C.c_void_p * self.n)(*[s._h for s in self.shards])
        rc = _lib.lib().nl_group_forward(hs, self.n, stream, int(token), int(pos),
                                         self.logits.ctypes.data_as(C.POINTER(C.c_float)))
        _lib.check(self.shards[0]._h, rc)
        return self.logits

    def close(self):
        for s in self.shards:
            s.close()


def comm_unique_id() -> bytes:
    buf = C.create_string_buffer(_lib.NL_COMM_ID_BYTES)
    rc = _lib.lib().nl_comm_get_unique_id(buf)
    if rc != 0:
        raise _lib.NlError(rc, (_lib.lib().nl_last_error(None) or b"").decode())
    return buf.raw


def op_matmul(w_raw: np.ndarray, ggml_type: int, x: np.ndarray, rows: int, cols: int, device: int = 0) -> np.ndarray:
    """matmulDispatch go/model.go:361-386 on the device (single GEMV, host in/out)."""
    w_raw = np.ascontiguousarray(w_raw)
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.zeros(rows, dtype=np.float32)
    rc = _lib.lib().nl_op_matmul(device, ggml_type, w_raw.ctypes.data, w_raw.nbytes,
                                 x.ctypes.data_as(C.POINTER(C.c_float)), out.ctypes.data_as(C.POINTER(C.c_float)),
                                 rows, cols)
    if rc != 0:
        raise _lib.NlError(rc, "nl_op_matmul")
    return out


def op_matmul_batch(w_raw: np.ndarray, ggml_type: int, x: np.ndarray, rows: int, cols: int, device: int = 0) -> np.ndarray:
    """W @ x[n] for every row of x ([n_tokens, cols]) through the MFMA multi-token path."""
    w_raw = np.ascontiguousarray(w_raw)
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.zeros((x.shape[0], rows), dtype=np.float32)
    fp = C.POINTER(C.c_float)
    rc = _lib.lib().nl_op_matmul_batch(device, ggml_type, w_raw.ctypes.data, w_raw.nbytes, x.ctypes.data_as(fp),
                                       out.ctypes.data_as(fp), rows, cols, x.shape[0])
    if rc != 0:
        raise _lib.NlError(rc, "nl_op_matmul_batch")
    return out


def op_sample(logits: np.ndarray, temperature: float, top_p: float, top_k: int, rep_penalty: float, rep_window: int,
              uniform: float, recent: List[int], device: int = 0):
    """One on-device sampling decision (go/main.go:177-200, :294-408) on host logits.
    Returns (picked id, logits after the in-place penalty, recent window afterwards)."""
    lg = np.array(logits, dtype=np.float32, copy=True)
    p = _lib.NlSampleParams(float(temperature), float(top_p), int(top_k), float(rep_penalty), int(rep_window))
    rec = (C.c_int * max(int(rep_window), 1))(*[int(t) for t in recent])
    nrec = C.c_int(len(recent))
    picked = C.c_int(-1)
    rc = _lib.lib().nl_op_sample(device, lg.ctypes.data_as(C.POINTER(C.c_float)), int(lg.size), C.byref(p),
                                 C.c_float(uniform), rec, C.byref(nrec), C.byref(picked))
    if rc != 0:
        raise _lib.NlError(rc, "nl_op_sample")
    return picked.value, lg, [int(rec[i]) for i in range(nrec.value)]


def op_exp(x: np.ndarray, device: int = 0) -> np.ndarray:
    """float32(exp(float64(x))) as the forward kernels compute it (go/quant.go:619, :629-631)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.zeros_like(x)
    fp = C.POINTER(C.c_float)
    rc = _lib.lib().nl_op_exp(device, x.ctypes.data_as(fp), out.ctypes.data_as(fp), x.size)
    if rc != 0:
        raise _lib.NlError(rc, "nl_op_exp")
    return out


def op_rmsnorm(x: np.ndarray, w: np.ndarray, eps: float, device: int = 0) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    out = np.zeros_like(x)
    fp = C.POINTER(C.c_float)
    rc = _lib.lib().nl_op_rmsnorm(device, x.ctypes.data_as(fp), w.ctypes.data_as(fp), C.c_float(eps),
                                  out.ctypes.data_as(fp), len(x))
    if rc != 0:
        raise _lib.NlError(rc, "nl_op_rmsnorm")
    return out
