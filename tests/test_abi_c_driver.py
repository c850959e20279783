"""The C ABI from plain C: include/nanollama_hip.h must compile as C99 (no C++-isms, no torch types), and a C
program linked against libnanollama_hip.so must reproduce the reference-Python golden logits -- the same call
sequence a cgo shim makes (integration/go/hip_backend.go)."""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

from nanollama_amd import _lib, gguf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
SRC = os.path.join(ROOT, "tests", "abi_driver.c")
CFLAGS = ["-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic-errors", "-I", os.path.join(ROOT, "include")]


def test_header_and_driver_compile_as_c99(tmp_path):
    subprocess.check_call(["gcc"] + CFLAGS + ["-c", SRC, "-o", str(tmp_path / "abi_driver.o")])
    # the header alone, as the first include of an otherwise empty translation unit
    (tmp_path / "h.c").write_text('#include "nanollama_hip.h"\nint main(void) { return sizeof(nl_config) == 68 ? 0 : 1; }\n')
    subprocess.check_call(["gcc"] + CFLAGS + [str(tmp_path / "h.c"), "-o", str(tmp_path / "h")])
    assert subprocess.call([str(tmp_path / "h")]) == 0
    assert len(bytes(_lib.NlConfig())) == 68          # the ctypes mirror has the same layout


def write_driver_file(path, tag):
    g = gguf.load_gguf(os.path.join(GOLDEN, tag + ".gguf"))
    v = np.load(os.path.join(GOLDEN, tag + ".npz"))
    m = g.meta
    head_dim = m.head_dim or m.embed_dim // m.num_heads
    with open(path, "wb") as f:
        f.write(b"NLDRV1\0\0")
        f.write(struct.pack("<8i2f7i", m.num_layers, m.embed_dim, m.num_heads, m.num_kv_heads, head_dim, m.interm_size,
                            m.vocab_size, min(m.seq_len, 2048), m.rms_norm_eps, m.rope_theta, int(m.qk_norm),
                            int(m.rope_conjugate), 1, 0, 0, 1, 0))
        f.write(struct.pack("<i", len(g.tensor_order)))
        for name in g.tensor_order:
            data, info = g.get_tensor(name)
            data = np.ascontiguousarray(data)
            rows, cols = (1, info.dims[0]) if info.ndims == 1 else (info.dims[1], info.dims[0])
            nb = name.encode()
            f.write(struct.pack("<i", len(nb)) + nb + struct.pack("<IQQQ", info.type, rows, cols, data.nbytes))
            f.write(data.tobytes())
        toks = [int(t) for t in v["prompt"]]
        logits = np.ascontiguousarray(v["logits_full"][:len(toks)], dtype=np.float32)
        f.write(struct.pack("<i", len(toks)) + struct.pack(f"<{len(toks)}i", *toks))
        f.write(logits.tobytes())


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["tiny_q8_0", "tiny_mha_q4_0"])
def test_c_program_reproduces_the_golden_logits(tmp_path, tag):
    exe = str(tmp_path / "abi_driver")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["gcc"] + CFLAGS + [SRC, "-o", exe, "-L", libdir, "-lnanollama_hip", "-lm", f"-Wl,-rpath,{libdir}"])
    data = str(tmp_path / "model.bin")
    write_driver_file(data, tag)
    out = subprocess.run([exe, data, "1e-4"], capture_output=True, text=True, timeout=120)
    sys.stdout.write(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "max |logit - golden|" in out.stdout
