"""profiles/<round>_traffic.json from the per-kernel PMC summaries written by tools/pmc_run.sh.

    python tools/make_traffic_json.py r02

HBM bytes per launch = FETCH_SIZE [KiB] x 1024 x 2 (gfx950: FETCH_SIZE reports half of a wide coalesced read
stream, MI355X_MICROARCH.md section HBM) + WRITE_SIZE [KiB] x 1024.  Kernel template names are mapped to the
bench.py kind names through the (prologue, epilogue) template arguments of gemv_kernel.  The file records the hash of
nanollama_amd/csrc/nl_kernels.h it was measured on: bench.py reports `traffic` only while that still matches.
"""
import csv, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nanollama_amd import _lib  # noqa: E402

ROUND = sys.argv[1] if len(sys.argv) > 1 else "r02"
KIND = {("1", "3"): "qkv_rope", ("2", "1"): "wo_resid", ("1", "2"): "gate_up_swiglu", ("3", "2"): "gate_up_swiglu",
        ("0", "1"): "down_resid", ("1", "0"): "lm_head"}


def kind_of(name):
    m = re.search(r"gemv_kernel<\d+, (\d+), (\d+)(, \d+)?>", name)
    if m:
        return KIND.get((m.group(1), m.group(2)))
    for key, kind in (("ffn_block_kernel", "ffn_block"), ("wide_ffn_kernel", "ffn_block"), ("tp_ffn_kernel", "ffn_block"), ("attn_block_kernel", "attn_block"),
                      ("tp_attn_kernel", "attn_block"), ("qkv_attn_kernel", "attn_block"), ("attn_kernel", "attention"),
                      ("argmax_kernel", "argmax"), ("embed_kernel", "embed")):
        if key in name and "bembed" not in name and "bargmax" not in name:
            return kind
    return None


def load(tag, ctr):
    out = {}
    path = os.path.join(ROOT, "profiles", f"{ROUND}_{tag}_pmc_{ctr}.csv")
    with open(path) as fh:
        for row in csv.DictReader(fh):
            k = kind_of(row["kernel"])
            if k and k not in out:          # rows are sorted by total traffic: the decode launch of a kind comes first
                out[k] = float(row["mean_" + ctr])
    return out


def persistent(tag, tokens_per_launch):
    """the persistent decode launch (nl_persist.h): the per-dispatch rows of tools/collect_r05.sh; the steady-state launch is the median
    dispatch (twelve of the fourteen are the bench's 20-token segments; the 128-token and the warm-up launch differ)"""
    import collections
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        path = os.path.join(ROOT, "profiles", f"{ROUND}_{tag}_pmc_{ctr}.csv")
        if not os.path.exists(path):
            return None
        with open(path) as fh:
            v = [float(r["mean_" + ctr]) for r in csv.DictReader(fh) if r["kernel"].startswith("pd_decode_kernel dispatch")]
        if not v:
            return None
        vals[ctr] = sorted(v)[len(v) // 2]          # the median dispatch: one of the bench's equal segments
    return {"bytes_per_launch": int(vals["FETCH_SIZE"] * 1024 * 2 + vals["WRITE_SIZE"] * 1024), "tokens_per_launch": tokens_per_launch,
            "fetch_bytes": int(vals["FETCH_SIZE"] * 1024 * 2), "write_bytes": int(vals["WRITE_SIZE"] * 1024)}


res = {"_doc": "HBM bytes per launch from rocprofv3 PMC (FETCH_SIZE KiB x1024 x2 [gfx950 correction] + WRITE_SIZE KiB x1024); "
               f"tools/make_traffic_json.py from profiles/{ROUND}_*_pmc_*.csv",
       "nl_kernels_sha16": _lib.source_sha([os.path.join(ROOT, "nanollama_amd", "csrc", "nl_kernels.h")])}
for tag in ("nano_q8_0", "big_q4_0"):
    f, w = load(tag, "FETCH_SIZE"), load(tag, "WRITE_SIZE")
    res[tag] = {k: int(round(f[k] * 1024 * 2 + w.get(k, 0.0) * 1024)) for k in f}
pd = persistent("nano_q8_0", int(sys.argv[2]) if len(sys.argv) > 2 else 20)
if pd:
    res["nano_q8_0"]["persistent_decode"] = pd
    res["nl_persist_sha16"] = _lib.source_sha([os.path.join(ROOT, "nanollama_amd", "csrc", "nl_persist.h")])
json.dump(res, open(os.path.join(ROOT, "profiles", f"{ROUND}_traffic.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
