"""ctypes binding of libnanollama_hip.so (include/nanollama_hip.h).

The product path has no CPU fallback: if the HIP library is missing or no GPU is
visible, loading / creating a model raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NL_LIB_PATH") or os.path.join(_HERE, "libnanollama_hip.so")  # NL_LIB_PATH: A/B builds (tools/)
NL_NUM_KINDS = 11
NL_COMM_ID_BYTES = 128
NL_P2P_HANDLE_BYTES = 64
NL_FLAG_NO_GRAPH = 1
NL_FLAG_LOCAL_GROUP = 2
NL_FLAG_GROUP_FUSED = 4

STATUS = {0: "NL_OK", -1: "NL_ERR_INVALID", -2: "NL_ERR_UNSUPPORTED", -3: "NL_ERR_HIP", -4: "NL_ERR_STATE",
          -5: "NL_ERR_MISSING", -6: "NL_ERR_COMM"}


class NlConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_layers", "dim", "n_heads", "n_kv_heads", "head_dim", "interm", "vocab",
                                         "seq_len")] + \
               [("rms_eps", C.c_float), ("rope_theta", C.c_float)] + \
               [(n, C.c_int32) for n in ("qk_norm", "rope_conjugate", "max_streams", "device", "tp_rank", "tp_size",
                                         "flags")]


class NlSampleParams(C.Structure):
    """nl_sample_params: the knobs of Engine.Generate's sampling branch (go/main.go:29-34, :135-140)."""
    _fields_ = [("temperature", C.c_float), ("top_p", C.c_float), ("top_k", C.c_int32), ("rep_penalty", C.c_float),
                ("rep_window", C.c_int32)]


class NlError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"{STATUS.get(code, code)}: {msg}")
        self.code = code


def source_files():
    """The library's sources in the order csrc/Makefile hashes them."""
    src_dir = os.path.join(_HERE, "csrc")
    hdrs = sorted(f for f in os.listdir(src_dir) if f.endswith(".h"))
    return [os.path.join(src_dir, "nl_engine.hip")] + [os.path.join(src_dir, f) for f in hdrs] + \
           [os.path.join(os.path.dirname(_HERE), "include", "nanollama_hip.h")]


def source_sha(files=None) -> str:
    import hashlib
    h = hashlib.sha256()
    for f in files or source_files():
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def build_info() -> dict:
    """What the loaded binary says it was built from, next to the sources present now."""
    L = lib()
    L.nl_build_info.restype = C.c_char_p
    info = dict(kv.split("=", 1) for kv in L.nl_build_info().decode().split())
    tree = source_sha()
    return {"lib_src_sha16": info.get("src"), "lib_git_head_at_build": info.get("git"), "tree_src_sha16": tree,
            "lib_matches_tree": info.get("src") == tree}


def build(force: bool = False) -> str:
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU).  Rebuilds when the binary is
    missing, older than a source, or carries another source hash than the tree (mtimes do not survive every copy)."""
    src_dir = os.path.join(_HERE, "csrc")
    srcs = source_files() + [os.path.join(src_dir, "Makefile")]
    stale = not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if not stale and not force:
        with open(LIB_PATH, "rb") as fh:       # read the embedded string without loading the library
            stale = f"src={source_sha()} git=".encode() not in fh.read()
    if force or stale:
        subprocess.check_call(["make", "-C", src_dir, "-B"])
    return LIB_PATH


_lib = None

EXPORTS = ["nl_build_info", "nl_set_gamma", "nl_abi_version", "nl_device_count", "nl_create", "nl_upload_tensor", "nl_finalize", "nl_destroy",
           "nl_last_error", "nl_reset", "nl_forward", "nl_forward_argmax", "nl_decode_greedy", "nl_prefill", "nl_forward_batch", "nl_get_config",
           "nl_synchronize", "nl_timer_start", "nl_timer_stop", "nl_kernel_kind_name", "nl_profile_forward",
           "nl_memory_usage", "nl_debug_read", "nl_op_matmul", "nl_op_matmul_batch", "nl_op_rmsnorm", "nl_comm_get_unique_id",
           "nl_comm_init", "nl_group_forward", "nl_debug_stamps", "nl_sample_decode", "nl_op_sample", "nl_p2p_export",
           "nl_p2p_import", "nl_p2p_info", "nl_p2p_loopback", "nl_op_exp", "nl_plan_info", "nl_persist_info", "nl_create_group", "nl_host_logits",
           "nl_op_gran16_soak"]


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(the engine has no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    vp, i32, fp, ip = C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int)
    L.nl_abi_version.restype = i32
    L.nl_device_count.restype = i32
    L.nl_create.argtypes = [C.POINTER(NlConfig), C.POINTER(vp)]
    L.nl_upload_tensor.argtypes = [vp, C.c_char_p, C.c_uint32, vp, C.c_uint64, C.c_uint64, C.c_uint64]
    L.nl_finalize.argtypes = [vp]
    L.nl_destroy.argtypes = [vp]
    L.nl_set_gamma.argtypes = [vp, C.POINTER(C.c_int32), i32, vp, i32]
    L.nl_last_error.restype = C.c_char_p
    L.nl_last_error.argtypes = [vp]
    L.nl_reset.argtypes = [vp, i32]
    L.nl_forward.argtypes = [vp, i32, i32, i32, fp]
    L.nl_forward_argmax.argtypes = [vp, i32, i32, i32, ip]
    L.nl_decode_greedy.argtypes = [vp, i32, i32, i32, i32, ip, ip]
    L.nl_prefill.argtypes = [vp, i32, ip, i32, i32, fp]
    L.nl_forward_batch.argtypes = [vp, ip, ip, ip, i32, fp, ip]
    L.nl_get_config.argtypes = [vp, C.POINTER(NlConfig)]
    L.nl_synchronize.argtypes = [vp]
    L.nl_timer_start.argtypes = [vp]
    L.nl_timer_stop.argtypes = [vp, fp]
    L.nl_kernel_kind_name.restype = C.c_char_p
    L.nl_kernel_kind_name.argtypes = [i32]
    L.nl_profile_forward.argtypes = [vp, i32, i32, i32, i32, fp, ip]
    L.nl_memory_usage.argtypes = [vp] + [C.POINTER(C.c_uint64)] * 3
    L.nl_debug_read.restype = C.c_int64
    L.nl_debug_read.argtypes = [vp, C.c_char_p, i32, fp, C.c_int64]
    L.nl_op_matmul.argtypes = [i32, C.c_uint32, vp, C.c_uint64, fp, fp, i32, i32]
    L.nl_op_matmul_batch.argtypes = [i32, C.c_uint32, vp, C.c_uint64, fp, fp, i32, i32, i32]
    L.nl_op_rmsnorm.argtypes = [i32, fp, fp, C.c_float, fp, i32]
    L.nl_comm_get_unique_id.argtypes = [vp]
    L.nl_comm_init.argtypes = [vp, vp]
    L.nl_p2p_export.argtypes = [vp, vp]
    L.nl_p2p_import.argtypes = [vp, vp]
    L.nl_p2p_info.argtypes = [vp, ip, ip]
    L.nl_p2p_loopback.argtypes = [vp]
    L.nl_plan_info.argtypes = [vp, ip, ip, ip, ip]
    L.nl_create_group.argtypes = [C.POINTER(NlConfig), ip, i32, C.POINTER(vp)]
    L.nl_host_logits.argtypes = [vp]
    L.nl_host_logits.restype = C.POINTER(C.c_float)
    L.nl_persist_info.argtypes = [vp, ip, ip, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
    L.nl_op_exp.argtypes = [C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int]
    L.nl_op_gran16_soak.argtypes = [C.c_int, C.c_int, C.c_uint, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    L.nl_group_forward.argtypes = [C.POINTER(vp), i32, i32, i32, i32, fp]
    L.nl_sample_decode.argtypes = [vp, i32, i32, i32, C.POINTER(NlSampleParams), fp, ip, ip, ip, ip]
    L.nl_op_sample.argtypes = [i32, fp, i32, C.POINTER(NlSampleParams), C.c_float, ip, ip, ip]
    _lib = L
    return L


def check(handle, rc: int):
    if rc != 0:
        msg = lib().nl_last_error(handle)
        raise NlError(rc, msg.decode() if msg else "")
