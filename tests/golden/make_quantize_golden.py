#!/usr/bin/env python3
"""Runs the REFERENCE's scripts/quantize_gguf.py (build container only) on tests/golden/tiny_f16.gguf and records
the SHA-256 of its output; tests/test_quantize_cli.py requires nanollama_amd.quantize to reproduce it."""
import hashlib, json, os, runpy, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "q8.gguf")
    sys.argv = ["quantize_gguf.py", os.path.join(HERE, "tiny_f16.gguf"), out]
    so = sys.stdout; sys.stdout = open(os.devnull, "w")
    try:
        runpy.run_path("/root/reference/scripts/quantize_gguf.py", run_name="__main__")
    finally:
        sys.stdout = so
    blob = open(out, "rb").read()
json.dump({"input": "tiny_f16.gguf", "tool": "scripts/quantize_gguf.py", "size": len(blob),
           "sha256": hashlib.sha256(blob).hexdigest()}, open(os.path.join(HERE, "quantize_golden.json"), "w"))
print(len(blob), hashlib.sha256(blob).hexdigest())
