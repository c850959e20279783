#!/usr/bin/env python3
"""bench.py -- decode tokens/sec + achieved-HBM-bandwidth fraction on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one decoded token: one pass of the hot path (Forward, go/model.go:490-620
restated as HIP kernels) plus the greedy argmax, chained on the device.

  N = 1 : headline = BASELINE.json configs[1], nano (89M) Q8_0 single-stream greedy decode;
          "secondary" = big (7.9B) Q4_0 on the one GPU (the 1-GPU point of configs[4]);
          "other_configs" = mini prefill, goldie prefill, goldie x 64 streams.
  N > 1 : headline = the SAME workload as N = 1 -- nano Q8_0 single-stream greedy decode -- as N independent
          replicas, one process per GPU, no data-path collective (the smallest tier does not shard: "scaling": "weak",
          value = tokens of all ranks / the slowest rank's time), so that the per-N values of one scaling run are
          values of one metric.  "secondary" = BASELINE.json configs[4], big (7.9B) Q4_0 tensor-parallel over the N
          GPUs (strong scaling: the two per-layer all-reduces are the push all-reduce over xGMI (nl_p2p_*), RCCL when
          that cannot be set up), with "reference_1gpu" = the same big model on rank 0's GPU alone, measured in the
          same run, and the first greedy ids compared against the tensor-parallel run.  A tensor-parallel transport
          that cannot be set up is reported inside "secondary"; it does not take the headline with it.

Launch: under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` the ranks come from the
environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  A plain `python bench.py --gpus N` starts the N rank
processes itself (fresh children, before this process touches a GPU) and exits with their status.  In both cases a
rank that finds WORLD_SIZE != N exits non-zero: a line can never carry n_gpus = N from fewer ranks, and the line
reports "ranks" = the number of distinct ranks that met in the rendezvous.

Weights are a deterministic random-weight GGUF written on the box in the reference exporter's layout
(nanollama_amd.synth); inputs are resident in HBM before the timed region.  The timed region -- exactly K steps
between barrier + synchronize -- is run REPEATS times and the median is the value (min / max beside it).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PROMPT_LEN = 8
SEGMENT = 128  # greedy 128-token decode (BASELINE.json configs)
REPEATS = 5


# ------------------------------------------------------------------ launcher (no GPU, no library) ---

def launch_ranks(n, argv, timeout_s=None):
    """`python bench.py --gpus N` outside a launcher: start N rank processes (this process has not loaded the HIP
    library and never will), relay rank 0's stdout, exit non-zero unless every rank exits zero."""
    timeout_s = timeout_s or int(os.environ.get("NL_BENCH_LAUNCH_TIMEOUT", "1500"))
    port = int(os.environ.get("MASTER_PORT", str(29500 + os.getpid() % 400)))
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    deadline = time.time() + timeout_s
    rcs = [None] * n
    out0 = b""
    try:
        out0, _ = procs[0].communicate(timeout=max(1.0, deadline - time.time()))
        rcs[0] = procs[0].returncode
        for r in range(1, n):
            rcs[r] = procs[r].wait(timeout=max(1.0, deadline - time.time()))
    except subprocess.TimeoutExpired:
        for p in procs:
            if p.poll() is None:
                p.kill()
        print(f"[bench] rank processes did not finish within {timeout_s}s", file=sys.stderr)
        return 3
    sys.stdout.write(out0.decode(errors="replace"))
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        print(f"[bench] ranks failed: {bad}", file=sys.stderr)
        return 1
    return 0


# ------------------------------------------------------------------ measurement ---

def kernel_bytes(shape, wtype, pos, tp=1, wo_in_block=None):
    """Algorithmic HBM bytes per LAUNCH of each kernel kind:
    every weight byte once + the vectors the kernel must read/write."""
    from nanollama_amd import gguf, synth
    t = synth.WTYPES[wtype]
    bpe = gguf.ggml_block_size(t) / gguf.ggml_block_elements(t)
    d, i, v, kv, hd = shape.dim, shape.ffn // tp, shape.vocab // tp, shape.kv_dim // tp, shape.head_dim
    hq = shape.n_head // tp * hd
    return {
        "embed": d * bpe + d * 4,
        "qkv_rope": (hq + 2 * kv) * d * bpe + 2 * d * 4 + (hq + 2 * kv) * 4,
        "attention": (pos + 1) * kv * 2 * 4 + hq * 4 * 2,
        # fused launches: small tiers run norm + QKV + attention + WO in one (nl_block.h); wide tiers QKV + attention (nl_group.h)
        # (wo_in_block: the plan has no separate WO launch -- small tiers, and a tensor-parallel rank's two-launch layers, nl_tp.h)
        "attn_block": (hq + 2 * kv) * d * bpe + (d * hq * bpe if (wo_in_block if wo_in_block is not None else (shape.n_head <= 12 and d <= 1024)) else 0)
                      + 2 * d * 4 + (pos + 1) * kv * 2 * 4,
        "ffn_block": 3 * i * d * bpe + 2 * d * 4,       # gate + up + down in one launch (nl_block.h), small tiers
        "wo_resid": d * hq * bpe + hq * 4 + 2 * d * 4,
        "gate_up_swiglu": 2 * i * d * bpe + 2 * d * 4 + i * 4,
        "down_resid": d * i * bpe + i * 4 + 2 * d * 4,
        "lm_head": v * d * bpe + 2 * d * 4 + v * 4,
        "argmax": shape.vocab * 4,
        "allreduce": tp * d * 8 + 2 * d * 4,     # tp tagged 8-byte granules per element in, the residual in and out
    }


def ensure_gguf(shape, wtype, mode, rank=0):
    from nanollama_amd import synth
    path = os.path.join(os.environ.get("NL_BENCH_DIR", "/tmp"), f"nl_bench_{shape.name}_{wtype}_{mode}.gguf")
    if rank == 0 and not os.path.exists(path):
        t0 = time.time()
        tmp = path + ".tmp"
        synth.generate_gguf(tmp, shape, wtype, mode=mode)
        os.replace(tmp, path)
        print(f"[bench] wrote {path} ({os.path.getsize(path) / 1e6:.1f} MB) in {time.time() - t0:.1f}s", file=sys.stderr)
    return path


def cpu_baseline(path, prompt, budget_s=20.0, min_tokens=1, max_tokens=SEGMENT):
    """The Go engine's algorithm (C restatement, oracle/) timed on this box's
    host cores on a bounded sample of the same workload (timer after the prompt, go/main.go:171,222-226).
    Reported, not optimised against."""
    from nanollama_amd import gguf
    from oracle import oracle
    ncpu = os.cpu_count() or 1
    m = oracle.OracleModel(gguf.load_gguf(path))
    pos = 0
    oracle.set_threads(min(ncpu, 16))
    for t in prompt:
        m.forward(t, pos)
        pos += 1
    nxt = oracle.argmax(m.logits())
    # The Go engine uses runtime.NumCPU() workers; on a many-core host that mostly degenerates to the
    # serial path (rows < 4*NumCPU, go/quant.go:49).  Pick the best of a few worker counts, state it.
    best, cores = None, 1
    for c in sorted({1, min(ncpu, 8), min(ncpu, 16), min(ncpu, 32), min(ncpu, 64)}):
        oracle.set_threads(c)
        t0 = time.perf_counter()
        m.forward(nxt, pos)
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, cores = dt, c
    oracle.set_threads(cores)
    import numpy as np
    first = int(nxt)
    ids, margins = [], []
    n, t0 = 0, time.perf_counter()
    while n < max_tokens and (n < min_tokens or (time.perf_counter() - t0) < budget_s):   # timer after prefill, go/main.go:171
        m.forward(nxt, pos)
        lg = m.logits()
        nxt = oracle.argmax(lg)
        top2 = np.partition(lg, -2)[-2:]
        ids.append(int(nxt))
        margins.append(float(top2[1] - top2[0]))
        pos += 1
        n += 1
    dt = time.perf_counter() - t0
    m.close()
    return {"value": round(n / dt, 3), "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": f"{n} greedy decode tokens after an {len(prompt)}-token prefill, same GGUF, "
                      f"C restatement of go/quant.go+go/model.go with the Go row partition on {cores} threads (BEST OF 1..64 threads on a {ncpu}-cpu host; "
                      f"the Go engine itself would use runtime.NumCPU() workers)",
            "_first": first, "_ids": ids, "_margins": margins}


ID_MARGIN = 1e-3   # a greedy id may differ from the CPU engine's only where the CPU engine's own top-1 / top-2 logit gap is below this


def ids_check(cpu, dev_first, dev_ids):
    """the device's greedy ids against the ids the CPU baseline generated on the same file (the same
    prompt, the same first token): equal one for one, up to a step where the CPU engine's own top-2 margin is < ID_MARGIN
    (after such a step the two runs are different texts and the comparison stops)."""
    ids, margins = cpu.pop("_ids"), cpu.pop("_margins")
    first = cpu.pop("_first")
    n = min(len(ids), len(dev_ids))
    if first != dev_first:
        return {"ids_match_cpu": False, "n": 0, "note": f"first token after the prompt differs: cpu {first}, device {dev_first}"}
    for k in range(n):
        if ids[k] != dev_ids[k]:
            ok = margins[k] < ID_MARGIN
            return {"ids_match_cpu": bool(ok), "n": k, "of": n,
                    "note": f"step {k}: cpu {ids[k]} vs device {dev_ids[k]}, cpu top-2 margin {margins[k]:.2e} "
                            f"({'inside' if ok else 'OUTSIDE'} the {ID_MARGIN} margin; comparison stops here)"}
    return {"ids_match_cpu": True, "n": n, "of": n}


_DROPIN = None


def dropin_lib():
    """integration/c/dropin_loop.c built with gcc next to the library (C99, links libnanollama_hip.so): the per-token loops an
    unpatched Go host runs over the cgo shim, in compiled code."""
    global _DROPIN
    if _DROPIN is not None:
        return _DROPIN or None
    import ctypes as C
    from nanollama_amd import _lib
    out = os.path.join(os.environ.get("NL_BENCH_DIR", "/tmp"), "libnl_dropin.so")
    libdir = os.path.dirname(_lib.LIB_PATH)
    try:
        subprocess.check_call(["gcc", "-std=c99", "-D_POSIX_C_SOURCE=199309L", "-O2", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"),
                               os.path.join(ROOT, "integration", "c", "dropin_loop.c"), "-o", out, "-L", libdir,
                               "-l:" + os.path.basename(_lib.LIB_PATH), f"-Wl,-rpath,{libdir}"], stderr=subprocess.DEVNULL)
        L = C.CDLL(out)
        ip, fp = C.POINTER(C.c_int), C.POINTER(C.c_float)
        L.nl_dropin_forward_loop.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, fp, C.c_int, ip, C.POINTER(C.c_double)]
        L.nl_dropin_forward_argmax_loop.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, ip, C.POINTER(C.c_double)]
        L.nl_dropin_forward_inplace_loop.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, ip, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        _DROPIN = L
    except (OSError, subprocess.CalledProcessError) as exc:
        print(f"[bench] drop-in loop library not built ({exc!r})", file=sys.stderr)
        _DROPIN = False
    return _DROPIN or None


def dropin_rates(dev, first, pos0, n, chained_ids):
    """tok/s of the per-token loops -- nl_forward + host argmax (what integration/go/model_hip.patch gives an
    UNMODIFIED Engine.Generate, go/main.go:173-219) and nl_forward_argmax -- next to the chained nl_decode_greedy headline."""
    import ctypes as C
    import numpy as np
    L = dropin_lib()
    if L is None:
        return {"error": "gcc or the drop-in loop source is missing"}
    v = dev.config.vocab_size
    logits = np.zeros(v, dtype=np.float32)
    ids = (C.c_int * n)()
    sec, csec = C.c_double(0.0), C.c_double(0.0)
    out = {"loop": "integration/c/dropin_loop.c (C99, gcc -O2), one call per token", "tokens": n}
    for key, call in (("nl_forward_plus_host_argmax_tokens_per_s",
                       lambda: L.nl_dropin_forward_loop(dev._h, 0, first, pos0, n, logits.ctypes.data_as(C.POINTER(C.c_float)), v, ids, C.byref(sec))),
                      ("nl_forward_inplace_plus_host_argmax_tokens_per_s",      # State.Logits = nl_host_logits (integration/go/hip_backend.go)
                       lambda: L.nl_dropin_forward_inplace_loop(dev._h, 0, first, pos0, n, v, ids, C.byref(sec), C.byref(csec))),
                      ("nl_forward_argmax_tokens_per_s",
                       lambda: L.nl_dropin_forward_argmax_loop(dev._h, 0, first, pos0, n, ids, C.byref(sec)))):
        best = None
        for _ in range(3):
            rc = call()
            if rc != 0:
                return {"error": f"{key}: status {rc}"}
            best = sec.value if best is None else min(best, sec.value)
        out[key] = round(n / best, 1)
        if "inplace" in key:
            out["nl_forward_inplace_us_per_call"] = round(csec.value / n * 1e6, 1)      # (of the last repeat)
        out[key.replace("_tokens_per_s", "_ids_equal_chained")] = [int(ids[i]) for i in range(n)] == list(chained_ids[:n])
    return out


def measured_traffic(tier, wtype):
    """HBM bytes per launch from the rocprofv3 PMC passes (profiles/README.md), only if they were collected on the
    decode kernels the loaded library was built from (the file records the hash of nl_kernels.h)."""
    from nanollama_amd import _lib
    for name in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic.json")), reverse=True):
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", name)))
        except (OSError, ValueError):
            continue
        want = _lib.source_sha([os.path.join(ROOT, "nanollama_amd", "csrc", "nl_kernels.h")])
        if tj.get("nl_kernels_sha16") != want:
            continue
        return tj.get(f"{tier}_{wtype}"), name
    return None, None


def persist_traffic_current():
    """the persistent launch's PMC traffic is reported only while nl_persist.h is the source it was measured on"""
    from nanollama_amd import _lib
    for name in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic.json")), reverse=True):
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", name)))
        except (OSError, ValueError):
            continue
        if "nl_persist_sha16" in tj:
            return tj["nl_persist_sha16"] == _lib.source_sha([os.path.join(ROOT, "nanollama_amd", "csrc", "nl_persist.h")])
    return False


def run_workload(tier, wtype, rdv, steps, warmup, profile_pos, model, tp=True, comm=None, keep=None, shard_of=None, dropin=False):
    """Load the tier's random-weight GGUF, run `warmup` + REPEATS x `steps` chained greedy decode steps, profile the
    launches.  tp=True: the ranks of rdv form one tensor-parallel engine (comm = "p2p" push all-reduce or "rccl");
    tp=False: every rank is an independent replica (no data-path collective).  Timing = max over ranks."""
    import numpy as np
    from nanollama_amd import gguf, synth
    rank, local_rank = rdv.rank, rdv.local_rank
    replicas = 1 if tp else rdv.world
    world = rdv.world if tp else 1
    if shard_of:          # rank 0's shard of a shard_of-rank tensor-parallel group, alone on this GPU (nl_p2p_loopback)
        world = shard_of
    shape = synth.TIERS[tier]
    mode = "qrand" if tier in ("big", "goldie") else "float"
    path = ensure_gguf(shape, wtype, mode, rank)
    rdv.barrier()
    g = gguf.load_gguf(path)
    kw = {}
    if shard_of:
        kw["p2p_loopback"] = True
    elif world > 1 and comm == "p2p":
        kw["p2p_allgather"] = rdv.allgather_bytes
    elif world > 1 or os.environ.get("NL_FORCE_TP_PLAN"):
        kw["comm_id"] = rdv.broadcast_bytes(model.comm_unique_id)
    dev = model.load_llama_model(g, device=local_rank, tp_rank=0 if shard_of else (rank if world > 1 else 0), tp_size=world, **kw)
    try:
        prompt = synth.prompt_ids(PROMPT_LEN, shape.vocab)
        rdv.barrier()
        dev.prefill(prompt)
        prefill_logits = dev.state.logits.copy()
        first, pos0 = int(np.argmax(prefill_logits)), len(prompt)

        def run_steps(k):
            """k chained decode steps in segments of SEGMENT tokens; every segment restarts at pos0 on
            the still-valid prompt prefix of the KV cache, so no prefill is inside the loop."""
            done, ids = 0, []
            while done < k:
                seg = min(SEGMENT, k - done)
                ids = dev.decode_greedy(first, pos0, seg)
                done += seg
            return ids

        head_ids = dev.decode_greedy(first, pos0, min(SEGMENT, shape.seq_len - pos0))
        run_steps(warmup)
        walls, evs, ids = [], [], []
        for _ in range(REPEATS):
            dev.synchronize()
            rdv.barrier()
            dev.timer_start()
            t0 = time.perf_counter()
            ids = run_steps(steps)
            dev.synchronize()
            wall_ms = (time.perf_counter() - t0) * 1e3
            evs.append(dev.timer_stop())
            rdv.barrier()
            walls.append(rdv.max_over_ranks(wall_ms))
        order = sorted(range(REPEATS), key=lambda i: walls[i])
        med = order[REPEATS // 2]
        wall_ms = walls[med]
        ms_per_step = wall_ms / steps

        # per-launch device time: every launch of the plan replayed 20x back to back between HIP events on the
        # engine's stream (nl_profile_forward), at a mid-run position
        ppos = min(profile_pos if profile_pos is not None else pos0 + (min(SEGMENT, steps) - 1) // 2, shape.seq_len - 1)
        prof = dev.profile_forward(first, ppos, iters=20)
        wo_in_block = bool(prof.get("attn_block", (0, 0))[1]) and not prof.get("wo_resid", (0, 0))[1]
        kb = kernel_bytes(shape, wtype, ppos, tp=shard_of or world, wo_in_block=wo_in_block)
        kernels = {}
        for kind, (ms, calls) in prof.items():
            if calls:
                per = ms / calls
                kernels[kind] = {"launches": calls, "us_per_launch": round(per * 1e3, 3),
                                 "GBps": round(kb[kind] / (per * 1e-3) / 1e9, 1)}
        traffic, traffic_file = measured_traffic(tier, wtype) if world == 1 else (None, None)
        if shard_of:
            world = 1        # (one GPU did the work: fractions below are against ONE device's peak, bytes are the shard's)
        # dominant kernel = the kind with the largest TIME share of the step (launches x time per launch; ties broken by
        # name); the kind that moves the most algorithmic bytes per step is reported beside it as roofline_by_bytes
        cand = sorted(k for k in kernels if k not in ("argmax", "allreduce"))
        dom = max(cand, key=lambda k: (prof[k][0], k))          # prof[k][0] = sum over the kind's launches of ms per launch
        dom_bytes = max(cand, key=lambda k: (kb[k] * prof[k][1], k))
        step_ms_profiled = sum(prof[k][0] for k in kernels)

        def roof(kind):
            return {"bound": "hbm", "kernel": kind, "achieved": kernels[kind]["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(kernels[kind]["GBps"] / HBM_PEAK_GBS, 4), "traffic": (traffic or {}).get(kind),
                    "traffic_source": traffic_file, "bytes_per_launch": int(kb[kind]),
                    "us_per_launch": kernels[kind]["us_per_launch"], "launches_per_step": kernels[kind]["launches"],
                    "time_share_of_step": round(prof[kind][0] / step_ms_profiled, 3), "profiled_at_pos": ppos}
        mean_pos = pos0 + (min(SEGMENT, steps) - 1) / 2.0
        step_bytes = (synth.weight_bytes_per_token(shape, wtype) + synth.kv_bytes_per_token(shape, int(mean_pos))) / (shard_of or 1)
        step_gbs = step_bytes / (ms_per_step * 1e-3) / 1e9
        # The smallest tier's greedy chain is ONE persistent launch per segment (nl_persist.h): that launch IS the timed region's
        # dominant kernel.  Its duration is measured live with HIP events on the engine's stream around single launches of the
        # timed segment; achieved = algorithmic bytes of the segment's tokens (weights + f32 KV read per token, each
        # byte counted once per token although the weights stay on chip for the whole launch) / that duration.
        pinfo = dev.persist_info() if world == 1 and not shard_of else {"ready": False}
        persist_roof = None
        if pinfo["ready"] and pos0 + min(SEGMENT, steps) <= pinfo["max_pos"]:
            seg = min(SEGMENT, steps)
            n0 = pinfo["launches"]
            ts = []
            for _ in range(7):
                dev.synchronize()
                dev.timer_start()
                dev.decode_greedy(first, pos0, seg)
                ts.append(dev.timer_stop())
            ts.sort()
            launch_ms = ts[len(ts) // 2]
            assert dev.persist_info()["launches"] == n0 + 7, "the persistent launch was retired during the run: " + dev.last_error()
            gbs = step_bytes * seg / (launch_ms * 1e-3) / 1e9
            kernels["persistent_decode"] = {"launches": round(1.0 / seg, 5), "tokens_per_launch": seg, "us_per_launch": round(launch_ms * 1e3, 2),
                                            "us_per_token": round(launch_ms * 1e3 / seg, 3), "GBps": round(gbs, 1)}
            ptr = (traffic or {}).get("persistent_decode") or {}
            ptraffic = ptr.get("bytes_per_launch") if ptr.get("tokens_per_launch") == seg and persist_traffic_current() else None
            persist_roof = {"bound": "hbm", "kernel": "persistent_decode (pd_decode_kernel, nl_persist.h)", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": ptraffic, "traffic_source": traffic_file if ptraffic else None,
                            "bytes_per_launch": int(step_bytes * seg), "us_per_launch": round(launch_ms * 1e3, 2), "tokens_per_launch": seg,
                            "launches_per_step": round(1.0 / seg, 5), "time_share_of_step": round(min(1.0, launch_ms / seg / ms_per_step), 3),
                            "note": "one launch decodes the whole segment with the weights resident on chip; bytes are the algorithmic bytes of its tokens"}
            # What actually binds this launch is not HBM (its measured traffic is a fraction of the algorithmic bytes) but the chain of
            # in-launch hand-offs a token walks through: per layer q|k|v -> heads (one in-XCD hop), then o, x', h and the layer's
            # output each gathered by all 32 units of the XCD; per token seven cross-XCD hops and the chip-wide argmax exchange.
            # Priced with the primitives tools/xcd_exchange_probe.hip measured (profiles/r05_xcd_exchange_probe.log): 0.25 us per
            # in-XCD hop, 1.27 us per in-XCD all-gather of 576 values, 0.58 us per cross-XCD hop, 2.8 us chip-wide -- no compute.
            n_layers = int(dev.config.num_layers)
            chain_floor = n_layers * (4 * 1.27 + 0.25) + 7 * 0.58 + 2.8
            persist_roof.update({
                "binding": "latency of the in-launch hand-off chain (the weights never leave the chip), not HBM bandwidth",
                "hbm_traffic_GBps_measured": (round(ptraffic / (launch_ms * 1e-3) / 1e9, 1) if ptraffic else None),
                "chain_floor_us": round(chain_floor, 1),
                "chain_floor_model": f"{n_layers} layers x (4 all-gathers x 1.27 + 1 hop x 0.25) + 7 cross-XCD hops x 0.58 + 2.8 (argmax exchange), us per token",
                "us_per_token": round(launch_ms * 1e3 / seg, 2),
                "chain_floor_frac": round(chain_floor / (launch_ms * 1e3 / seg), 3)})
        p2p = dev.p2p_info() if (world > 1 or shard_of) else None
        res = {
            "tier": tier, "wtype": wtype, "path": path, "prompt": prompt,
            "tok_s": replicas * steps / (wall_ms / 1e3), "ms_per_step": ms_per_step, "device_ms_per_step": evs[med] / steps,
            "ms_per_step_min": min(walls) / steps, "ms_per_step_max": max(walls) / steps,
            "replicas": replicas, "tp": world, "p2p": p2p, "plan": dev.plan_info(),
            "step_bytes": int(step_bytes), "hbm_frac_whole_step": step_gbs / (HBM_PEAK_GBS * max(world, 1)),
            "kernels": kernels, "last_ids": ids[-4:], "head_ids": head_ids, "first_id": first, "prefill_logits": prefill_logits,
            "dropin": (dropin_rates(dev, first, pos0, 64 if tier != "big" else 32, head_ids) if dropin and world == 1 and not shard_of else None),
            "roofline": persist_roof or roof(dom), "roofline_by_bytes": roof(dom_bytes), "persist": pinfo,
            "launch_plan_kernels_note": ("the per-kind entries beside persistent_decode are the launch plans this handle keeps for per-call Forward, "
                                         "sampling and contexts beyond the persistent launch's position limit" if persist_roof else None),
            "segment": min(SEGMENT, steps), "pos_first": pos0, "pos_last": pos0 + min(SEGMENT, steps) - 1,
        }
    except BaseException:
        dev.close()      # (a failed attempt must not keep a 4 GB model and its receive area alive while a fallback loads another)
        raise
    if keep is not None:
        keep.append(dev)
    else:
        dev.close()
    return res


def prefill_x1(dev, toks, dt_hilo, gemm_flop):
    """The same prompt in the single-product precision mode (NL_PREFILL_PRECISION=fp16x1): time, and its logits
    against the default hi + lo mode's on this device (the hi + lo mode is the one held to 1e-4 against the CPU engine)."""
    import numpy as np
    dev.reset()
    dev.prefill(toks)
    base = dev.state.logits.copy()
    os.environ["NL_PREFILL_PRECISION"] = "fp16x1"
    try:
        dev.reset()
        dev.prefill(toks)
        dev.synchronize()
        dt = 1e9
        for _ in range(3):
            dev.reset()
            t0 = time.perf_counter()
            dev.prefill(toks)
            dt = min(dt, time.perf_counter() - t0)
        lg = dev.state.logits.copy()
    finally:
        del os.environ["NL_PREFILL_PRECISION"]
    scale = max(1.0, float(base.std()))
    return {"prefill_ms": round(dt * 1e3, 2), "prefill_tokens_per_s": round(len(toks) / dt, 1), "speedup_vs_hi_lo": round(dt_hilo / dt, 3),
            "mfma_peak_frac_f16_dense": round(gemm_flop / dt / 2.5e15, 5),
            "max_abs_logit_diff_vs_hi_lo_over_std": round(float(np.abs(lg - base).max()) / scale, 6), "stated_tolerance_over_std": 1e-2,
            "same_argmax": bool(int(np.argmax(lg)) == int(np.argmax(base)))}


def side_configs(model):
    """BASELINE.json configs[2] and [3] as side results of the single-GPU run: mini Q4_0 2047-token prefill
    (MFMA multi-token path) + one decode step, and goldie Q4_0 with 64 concurrent decode streams."""
    import numpy as np
    from nanollama_amd import gguf, synth
    out = {}
    # -- mini: prefill + decode
    shape = synth.TIERS["mini"]
    g = gguf.load_gguf(ensure_gguf(shape, "q4_0", "qrand"))
    dev = model.load_llama_model(g)
    toks = synth.prompt_ids(2047, shape.vocab)
    dev.prefill(toks)              # warm-up: allocates and first-touches the 2048-token step buffers
    dev.synchronize()
    dt = 1e9
    for _ in range(3):             # steady state (a server's n-th prompt), host wall clock incl. logits read-back
        dev.reset()
        t0 = time.perf_counter()
        dev.prefill(toks)
        dt = min(dt, time.perf_counter() - t0)
    first = int(np.argmax(dev.state.logits))
    dev.decode_greedy(first, 2047, 1)   # first replay uploads the graph
    t1 = time.perf_counter()
    dev.decode_greedy(first, 2047, 1)
    dt1 = time.perf_counter() - t1
    gemm_flop = 2.0 * shape.matrix_params() * 2047 - 2.0 * shape.vocab * shape.dim * 2046  # LM head on the last token only
    out["mini_q4_0_prefill_2047"] = {"prefill_tokens_per_s": round(2047 / dt, 1), "prefill_ms": round(dt * 1e3, 2),
                                     "gemm_TFLOPs_algorithmic": round(gemm_flop / dt / 1e12, 2),
                                     "mfma_peak_frac_f16_dense": round(gemm_flop / dt / 2.5e15, 5),
                                     "decode_step_ms_at_pos_2047": round(dt1 * 1e3, 3),
                                     "precision": "x = fp16 hi + fp16 lo (two MFMAs per product; logits within 1e-4 of the CPU engine)"}
    out["mini_q4_0_prefill_2047"]["fp16x1_mode"] = prefill_x1(dev, toks, dt, gemm_flop)
    dev.close()
    # -- goldie: the same prompt step at 841M parameters (what the MFMA path does with longer rows), then 64 streams
    shape = synth.TIERS["goldie"]
    g = gguf.load_gguf(ensure_gguf(shape, "q4_0", "qrand"))
    dev = model.load_llama_model(g)
    toks = synth.prompt_ids(2047, shape.vocab)
    dev.prefill(toks)
    dev.synchronize()
    dt = 1e9
    for _ in range(3):
        dev.reset()
        t0 = time.perf_counter()
        dev.prefill(toks)
        dt = min(dt, time.perf_counter() - t0)
    gemm_flop = 2.0 * shape.matrix_params() * 2047 - 2.0 * shape.vocab * shape.dim * 2046
    out["goldie_q4_0_prefill_2047"] = {"prefill_tokens_per_s": round(2047 / dt, 1), "prefill_ms": round(dt * 1e3, 2),
                                       "gemm_TFLOPs_algorithmic": round(gemm_flop / dt / 1e12, 2),
                                       "mfma_peak_frac_f16_dense": round(gemm_flop / dt / 2.5e15, 5)}
    out["goldie_q4_0_prefill_2047"]["fp16x1_mode"] = prefill_x1(dev, toks, dt, gemm_flop)
    dev.close()
    ns, steps, pos0 = 64, 32, 8
    dev = model.load_llama_model(g, max_streams=ns)
    rng = np.random.Generator(np.random.PCG64(3))
    ids = [int(t) for t in rng.integers(3, shape.vocab, size=ns)]
    streams = list(range(ns))
    for p in range(pos0):
        ids, _ = dev.forward_batch(streams, ids, [p] * ns)
    dev.synchronize()
    dev.timer_start()              # HIP events on the engine's stream around the same loop the wall clock brackets
    t0 = time.perf_counter()
    for k in range(steps):
        ids, _ = dev.forward_batch(streams, ids, [pos0 + k] * ns)
    dt = time.perf_counter() - t0
    ev_ms = dev.timer_stop()
    step_bytes = synth.weight_bytes_per_token(shape, "q4_0") + ns * synth.kv_bytes_per_token(shape, pos0 + steps // 2)
    out["goldie_q4_0_64_streams"] = {"tokens_per_s_aggregate": round(ns * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 3),
                                     "device_ms_per_step": round(ev_ms / steps, 3), "steps": steps,
                                     "positions": f"{pos0}..{pos0 + steps - 1}", "launch": "one nl_forward_batch call per step (host loop; ids read back every step); the step is a cached hipGraph of 5 launches per layer (nl_dgemm.h)",
                                     "hbm_frac": round(step_bytes / (dt / steps) / 1e9 / HBM_PEAK_GBS, 4)}
    # ... and the same batch deep in its context (the K / V rows below are whatever the cache holds: timing only)
    lpos, lsteps = 1000, 16
    for p in range(lpos - 4, lpos):
        ids, _ = dev.forward_batch(streams, ids, [p] * ns)
    dev.synchronize()
    t0 = time.perf_counter()
    for k in range(lsteps):
        ids, _ = dev.forward_batch(streams, ids, [lpos + k] * ns)
    dt = time.perf_counter() - t0
    step_bytes = synth.weight_bytes_per_token(shape, "q4_0") + ns * synth.kv_bytes_per_token(shape, lpos + lsteps // 2)
    out["goldie_q4_0_64_streams"]["at_position_1000"] = {"tokens_per_s_aggregate": round(ns * lsteps / dt, 1), "ms_per_step": round(dt / lsteps * 1e3, 3),
                                                          "positions": f"{lpos}..{lpos + lsteps - 1}",
                                                          "hbm_frac": round(step_bytes / (dt / lsteps) / 1e9 / HBM_PEAK_GBS, 4)}
    dev.close()
    # -- nano at the reference's DEFAULT generation settings (go/main.go:29-34: temp 0.8, top-p 0.9, repetition
    #    penalty 1.15 over 64 tokens): sampling loop on the device vs logits read back and sampled on the host
    from nanollama_amd.engine import Engine, GenParams
    shape = synth.TIERS["nano"]
    g = gguf.load_gguf(ensure_gguf(shape, "q8_0", "float"))
    dev = model.load_llama_model(g)
    prompt = synth.prompt_ids(PROMPT_LEN, shape.vocab)
    res = {}
    for key, on_device, n in (("device_loop_tokens_per_s", True, 256), ("host_loop_tokens_per_s", False, 48)):
        eng = Engine(dev, eos_id=-1, rep_penalty=1.15, rep_window=64, seed=1, device_sampling=on_device)
        eng.generate_ids(prompt, GenParams(max_tokens=16, temperature=0.8, top_p=0.9))
        t0 = time.perf_counter()
        ids = eng.generate_ids(prompt, GenParams(max_tokens=n, temperature=0.8, top_p=0.9))
        res[key] = round(len(ids) / (time.perf_counter() - t0), 1)
    out["nano_q8_0_default_sampling"] = res
    # -- nano greedy decode against the context length (the headline is timed at positions 8 .. 27): 32 chained tokens from each
    #    position, the persistent launch below its position limit (a head's 128-position passes shared by three units), the launch plans beyond
    try:
        long_prompt = synth.prompt_ids(1100, shape.vocab)
        dev.reset(); dev.prefill(long_prompt)
        bypos = {}
        # (the second witness of every timed chain: the launch plans of the same file, NL_PERSIST=0)
        os.environ["NL_PERSIST"] = "0"
        try:
            plain = model.load_llama_model(g)
        finally:
            del os.environ["NL_PERSIST"]
        plain.prefill(long_prompt)
        ids_equal = True
        for p0 in (64, 300, 470, 700, 980, 1060):
            ids_equal = ids_equal and dev.decode_greedy(5, p0, 32) == plain.decode_greedy(5, p0, 32)
            best = None
            for _ in range(3):
                t0 = time.perf_counter(); dev.decode_greedy(5, p0, 32); dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            bypos[str(p0)] = round(32 / best, 1)
        plain.close()
        out["nano_q8_0_greedy_tokens_per_s_by_position"] = dict(bypos, persistent_decode_below=dev.persist_info()["max_pos"],
                                                               ids_equal_launch_plans=bool(ids_equal))
    except Exception as exc:  # a side measurement: the line must survive it
        out["nano_q8_0_greedy_tokens_per_s_by_position"] = {"error": repr(exc)}
    dev.close()
    # -- the same settings on the 7.9B tier (96000-entry vocabulary: the selection streams its candidates out of L2)
    try:
        shape = synth.TIERS["big"]
        g = gguf.load_gguf(ensure_gguf(shape, "q4_0", "qrand"))
        dev = model.load_llama_model(g)
        prompt = synth.prompt_ids(PROMPT_LEN, shape.vocab)
        eng = Engine(dev, eos_id=-1, rep_penalty=1.15, rep_window=64, seed=1, device_sampling=True)
        eng.generate_ids(prompt, GenParams(max_tokens=16, temperature=0.8, top_p=0.9))
        t0 = time.perf_counter()
        ids = eng.generate_ids(prompt, GenParams(max_tokens=96, temperature=0.8, top_p=0.9))
        out["big_q4_0_default_sampling"] = {"device_loop_tokens_per_s": round(len(ids) / (time.perf_counter() - t0), 1)}
        # greedy decode against the context length (the secondary line is timed at short positions): 16 chained tokens from each
        # position; the two-launch layers at every position, a head's 256-position passes shared by four blocks from the second pass on
        long_prompt = synth.prompt_ids(1940, shape.vocab)
        dev.reset(); dev.prefill(long_prompt)
        bypos = {}
        for p0 in (64, 300, 700, 980, 1500, 1900):
            dev.decode_greedy(5, p0, 16)
            best = None
            for _ in range(3):
                t0 = time.perf_counter(); dev.decode_greedy(5, p0, 16); dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            bypos[str(p0)] = round(16 / best, 1)
        out["big_q4_0_greedy_tokens_per_s_by_position"] = dict(bypos, two_launch_layers_below=dev.plan_info()["fused_max_pos"])
        dev.close()
    except Exception as exc:  # a side measurement: the line must survive it
        out["big_q4_0_default_sampling"] = {"error": repr(exc)}
    return out


def shard_probe(model, rdv, ns=(2, 4, 8), seam_us=2.0, steps=96, warmup=16, base=None):
    """per-rank step time of big Q4_0 at tp_size N with NO contention -- rank 0's shard alone on
    this GPU with the real tensor-parallel plan (grids, granule stores into its own receive slots, polls;
    nl_p2p_loopback), only the xGMI hop missing.  predicted = per-rank time + 2 L seams x an ASSUMED wire latency
    (seam_us per seam: no xGMI figure has been measured by this repo) -- a prediction, labelled as such."""
    from nanollama_amd import synth
    shape = synth.TIERS["big"]
    out = {"workload": "big (7.9B) Q4_0: rank 0's shard of a tensor-parallel group of N, alone on one idle GPU "
                       "(nl_p2p_loopback: real plan, grids, granule stores and polls; no xGMI hop; logits are not a model's)",
           "assumed_seam_us": seam_us, "seams_per_step": 2 * shape.n_layer, "per_rank": {}, "predicted_scaling": {}}
    for n in ns:
        try:
            r = run_workload("big", "q4_0", rdv, steps, warmup, None, model, tp=False, shard_of=n)
        except Exception as exc:
            out["per_rank"][str(n)] = {"error": repr(exc)}
            continue
        kern = {k: {"launches": v["launches"], "us_per_launch": v["us_per_launch"]} for k, v in r["kernels"].items()}
        out["per_rank"][str(n)] = {"ms_per_step": round(r["ms_per_step"], 5), "launches_per_step": sum(v["launches"] for v in kern.values()),
                                   "plan": r["plan"],
                                   "shard_bytes_per_step": r["step_bytes"], "hbm_frac_of_one_gpu": round(r["hbm_frac_whole_step"], 4),
                                   "kernels": kern}
        pred_ms = r["ms_per_step"] + out["seams_per_step"] * seam_us * 1e-3
        out["predicted_scaling"][str(n)] = {"predicted_ms_per_step": round(pred_ms, 4), "predicted_tokens_per_s": round(1e3 / pred_ms, 1)}
        if base:
            out["predicted_scaling"][str(n)]["predicted_speedup_vs_1gpu"] = round(base / pred_ms, 2)
    # the same shard on the four-launches-per-layer plan (projection + attention fused, nl_group.h), for the comparison
    if 8 in ns and not os.environ.get("NL_TP_FUSED"):
        os.environ["NL_TP_FUSED"] = "0"
        try:
            r = run_workload("big", "q4_0", rdv, steps, warmup, None, model, tp=False, shard_of=8)
            out["per_rank_four_launch_plan"] = {"8": {"ms_per_step": round(r["ms_per_step"], 5), "plan": r["plan"],
                                                      "launches_per_step": sum(v["launches"] for v in r["kernels"].values())}}
        except Exception as exc:
            out["per_rank_four_launch_plan"] = {"error": repr(exc)}
        finally:
            del os.environ["NL_TP_FUSED"]
    out["label"] = "PREDICTION from one GPU: no store has crossed xGMI in this measurement"
    return out


def workload_text(shape, tier, wtype, r, steps):
    """What the timed region ran: `steps` chained greedy decode steps as segments that each restart behind the prompt."""
    seg = r["segment"]
    nseg = (steps + seg - 1) // seg
    return (f"{tier} ({shape.matrix_params() / 1e6:.0f}M matrix params) {wtype.upper()} GGUF, {PROMPT_LEN}-token prompt, then "
            f"{steps} timed greedy decode steps = {nseg} segment{'s' if nseg > 1 else ''} of {seg} tokens at positions "
            f"{r['pos_first']}..{r['pos_last']}, 1 stream")


def summary(r, keys=("kernels",)):
    out = {"value": round(r["tok_s"], 2), "unit": "tokens/s", "ms_per_step": round(r["ms_per_step"], 5),
           "ms_per_step_min": round(r["ms_per_step_min"], 5), "ms_per_step_max": round(r["ms_per_step_max"], 5),
           "hbm_frac_whole_step": round(r["hbm_frac_whole_step"], 4), "roofline": r["roofline"],
           "roofline_by_bytes": r["roofline_by_bytes"]}
    for k in keys:
        out[k] = r[k]
    if r.get("persist", {}).get("ready"):
        out["persistent_decode"] = r["persist"]      # nl_persist_info: the one-launch-per-chunk decode served the timed steps
        out["launch_plan_kernels_note"] = r.get("launch_plan_kernels_note")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--workload", default=None, help="tier:wtype override, e.g. big:q4_0 (default by --gpus)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the side measurements")
    ap.add_argument("--comm", default=None, choices=("p2p", "rccl"), help="N > 1: force the all-reduce transport")
    ap.add_argument("--shard-of", type=int, default=None, choices=(2, 4, 8),
                    help="one GPU: time rank 0's shard of big Q4_0 at tp_size N alone (loopback slots) and print that line only")
    ap.add_argument("--seam-us", type=float, default=2.0, help="assumed xGMI latency per all-reduce seam for the predicted-scaling table")
    ap.add_argument("--no-shard-probe", action="store_true", help="N = 1: skip the tensor-parallel shard-only probe of big")
    ap.add_argument("--rccl-worker", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--profile-pos", type=int, default=None, help="position of the per-launch profile (default: the mean timed position)")
    args = ap.parse_args()
    n = args.gpus
    if n < 1:
        raise SystemExit("--gpus must be >= 1")
    if n > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: this process becomes one, before anything of it touches the GPU
        raise SystemExit(launch_ranks(n, sys.argv[1:]))

    # the product library is loaded before anything else that could pull in a second HIP runtime
    import numpy as np
    from nanollama_amd import _lib, model, synth
    from nanollama_amd.dist import Rendezvous
    dry = os.environ.get("NL_BENCH_DRYRUN")   # tests/test_bench_launcher.py: rank bookkeeping only, no GPU work
    if not dry:
        _lib.lib()
    try:
        rdv = Rendezvous(timeout_s=float(os.environ.get("NL_RDV_TIMEOUT", "300")))
    except (TimeoutError, OSError) as exc:
        print(f"[bench] rendezvous failed: {exc}", file=sys.stderr)
        raise SystemExit(2)
    rank, world = rdv.rank, rdv.world
    if world != n:
        print(f"[bench] --gpus {n} but this rank sees WORLD_SIZE={world}: refusing to report n_gpus={n}", file=sys.stderr)
        raise SystemExit(2)
    ranks_seen = len(set(rdv.allgather_bytes(str(rank).encode())))
    if ranks_seen != n:
        print(f"[bench] only {ranks_seen} of {n} ranks met in the rendezvous", file=sys.stderr)
        raise SystemExit(2)
    if dry:
        if rank == 0:
            print(json.dumps({"n_gpus": n, "ranks": ranks_seen, "dryrun": True}))
        rdv.close()
        return
    common = {"unit": "tokens/s", "n_gpus": n, "ranks": ranks_seen, "steps": args.steps, "warmup": args.warmup,
              "repeats": REPEATS, "higher_is_better": True, "vs_baseline": None, "dtype": "f32",
              "data": "synthetic (random-weight GGUF, reference exporter layout, nanollama_amd.synth)",
              "build": _lib.build_info()}

    if args.rccl_worker is not None:   # guarded child of an N > 1 run: big Q4_0 over RCCL
        r = run_workload("big", "q4_0", rdv, args.steps, args.warmup, args.profile_pos, model, tp=True, comm="rccl")
        if rank == 0:
            r.pop("prefill_logits")
            print("TPJSON " + json.dumps(r))
        rdv.close()
        return

    if args.shard_of:
        if n != 1:
            raise SystemExit("--shard-of runs on one GPU")
        out = dict(common, metric=f"per-rank decode step, big Q4_0 shard 0 of {args.shard_of} (loopback)", scaling="strong",
                   **shard_probe(model, rdv, ns=(args.shard_of,), seam_us=args.seam_us, steps=args.steps, warmup=args.warmup))
        print(json.dumps(out))
        rdv.close()
        return

    if args.workload:
        tier, wtype = args.workload.split(":")
        shape = synth.TIERS[tier]
        r = run_workload(tier, wtype, rdv, args.steps, args.warmup, args.profile_pos, model, tp=world > 1,
                         comm=args.comm or "p2p")
        out = dict(common, metric=f"decode tokens/sec, {tier} {wtype.upper()} single-stream greedy"
                   + (f", tensor-parallel over {world} GPUs" if world > 1 else ""),
                   scaling="strong" if world > 1 else "weak",
                   config={"workload": workload_text(shape, tier, wtype, r, args.steps), "parallelism": f"tp{world}" if world > 1 else "single-gpu",
                           "allreduce": (args.comm or "p2p") if world > 1 else None},
                   device_ms_per_step=round(r["device_ms_per_step"], 5), algorithmic_bytes_per_step=r["step_bytes"],
                   last_ids=r["last_ids"], **summary(r))
        if rank == 0:
            print(json.dumps(out))
        rdv.close()
        return

    if n == 1:
        tier, wtype = "nano", "q8_0"
        shape = synth.TIERS[tier]
        r = run_workload(tier, wtype, rdv, args.steps, args.warmup, args.profile_pos, model, tp=False, dropin=not args.no_secondary)
        out = dict(common, metric=f"decode tokens/sec, {tier} {wtype.upper()} single-stream greedy", scaling="weak",
                   config={"workload": workload_text(shape, tier, wtype, r, args.steps) + " per GPU", "parallelism": "single-gpu",
                           "weights": f"{wtype} blocks dequantised in-register, f32 activations and KV cache"},
                   device_ms_per_step=round(r["device_ms_per_step"], 5), algorithmic_bytes_per_step=r["step_bytes"],
                   last_ids=r["last_ids"], **summary(r))
        bad_ids = []
        if r.get("dropin"):
            out["dropin_forward_loop"] = r["dropin"]
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(r["path"], r["prompt"])
            out.update({"greedy_ids_vs_cpu": ids_check(out["cpu_baseline"], r["first_id"], r["head_ids"])})
            if not out["greedy_ids_vs_cpu"]["ids_match_cpu"]:
                bad_ids.append("nano")
            if not args.no_secondary:
                # BASELINE.json configs[0]: nano f16, greedy 128-token decode on the CPU engine (go/quant.go:527-563), no GPU involved
                try:
                    f16 = cpu_baseline(ensure_gguf(shape, "f16", "float"), r["prompt"], budget_s=15.0)
                    for k in ("_ids", "_margins", "_first"):
                        f16.pop(k, None)
                    out["config0_nano_f16_cpu"] = f16
                except Exception as exc:
                    out["config0_nano_f16_cpu"] = {"error": repr(exc)}
        if not args.no_secondary:
            # BASELINE.json's metric also names big Q4_0 @ 1 GPU: the 1-GPU point of the tensor-parallel curve
            try:
                b = run_workload("big", "q4_0", rdv, 96, 16, args.profile_pos, model, dropin=True)
                out["secondary"] = dict(workload=workload_text(synth.TIERS["big"], "big", "q4_0", b, 96) + ", 1 GPU", **summary(b))
                if b.get("dropin"):
                    out["secondary"]["dropin_forward_loop"] = b["dropin"]
                if not args.no_cpu_baseline:
                    # north_star tabulates both tiers next to the CPU engine; SURVEY 8(d) allows a shortened run for big
                    out["secondary"]["cpu_baseline"] = cpu_baseline(b["path"], b["prompt"], budget_s=12.0, min_tokens=4, max_tokens=16)
                    out["secondary"]["greedy_ids_vs_cpu"] = ids_check(out["secondary"]["cpu_baseline"], b["first_id"], b["head_ids"])
                    if not out["secondary"]["greedy_ids_vs_cpu"]["ids_match_cpu"]:
                        bad_ids.append("big")
            except Exception as exc:  # the headline result must survive a failure of the side measurement
                out["secondary"] = {"error": repr(exc)}
            try:
                out["other_configs"] = side_configs(model)
            except Exception as exc:
                out["other_configs"] = {"error": repr(exc)}
            if not args.no_shard_probe:
                try:
                    base = out["secondary"].get("ms_per_step") if isinstance(out.get("secondary"), dict) else None
                    out["tensor_parallel_shard_probe"] = shard_probe(model, rdv, seam_us=args.seam_us, base=base)
                except Exception as exc:
                    out["tensor_parallel_shard_probe"] = {"error": repr(exc)}
        for cb in (out.get("cpu_baseline"), (out.get("secondary") or {}).get("cpu_baseline") if isinstance(out.get("secondary"), dict) else None):
            if isinstance(cb, dict):
                for k in ("_ids", "_margins", "_first"):
                    cb.pop(k, None)
        print(json.dumps(out))
        rdv.close()
        if bad_ids:
            print(f"[bench] greedy ids differ from the CPU engine's outside the {ID_MARGIN} margin: {bad_ids}", file=sys.stderr)
            raise SystemExit(4)
        return

    # ---- N > 1: the headline workload as N independent replicas (weak scaling, no collective); the 7.9B tier tensor-parallel
    #      over the N GPUs (BASELINE.json configs[4]) as "secondary" ----
    tier, wtype = "nano", "q8_0"
    r = run_workload(tier, wtype, rdv, args.steps, args.warmup, args.profile_pos, model, tp=False)
    out = dict(common, metric=f"decode tokens/sec, {tier} {wtype.upper()} single-stream greedy", scaling="weak",
               config={"workload": workload_text(synth.TIERS[tier], tier, wtype, r, args.steps) + f" per GPU, {n} independent replicas",
                       "parallelism": f"replicas x{n} (one process per GPU, no data-path collective)",
                       "weights": f"{wtype} blocks dequantised in-register, f32 activations and KV cache"},
               device_ms_per_step=round(r["device_ms_per_step"], 5), algorithmic_bytes_per_step=r["step_bytes"],
               last_ids=r["last_ids"], **summary(r))
    shape = synth.TIERS["big"]
    ref1 = None
    tp_res, transport, notes = None, None, []
    watchdog = None
    if not args.no_secondary:
        # The tensor-parallel wire has never run across devices on this repository's test pool: whatever it
        # does on the real node -- every poll in it is bounded, RCCL runs in guarded children -- it must not take the measured
        # headline with it.  Past the deadline every rank leaves; rank 0 prints the line it has.
        import threading

        def give_up_secondary():
            if rank == 0:
                late = dict(out, secondary={"metric": f"decode tokens/sec, big Q4_0 single-stream greedy, tensor-parallel over {n} GPUs",
                                            "value": None, "error": "the tensor-parallel measurement did not finish within its deadline",
                                            "attempts": notes})
                print(json.dumps(late))
                sys.stdout.flush()
            os._exit(0)
        watchdog = threading.Timer(float(os.environ.get("NL_TP_DEADLINE", "600")), give_up_secondary)
        watchdog.daemon = True
        watchdog.start()
        # the same model on rank 0's GPU alone, in the same run: the 1-GPU point of the curve and the parity anchor
        if rank == 0:
            try:
                from nanollama_amd.dist import Rendezvous as _R
                solo = _R.solo()
                ref1 = run_workload("big", "q4_0", solo, 96, 16, args.profile_pos, model, tp=False)
            except Exception as exc:
                ref1 = {"error": repr(exc)}
        rdv.barrier()
        for comm in ([args.comm] if args.comm else ["p2p", "rccl"]):
            if comm == "p2p":
                err = None
                try:
                    cand = run_workload("big", "q4_0", rdv, min(args.steps, 128), min(args.warmup, 16), args.profile_pos, model, tp=True, comm="p2p")
                except Exception as exc:
                    cand, err = None, repr(exc)
                # every rank must have succeeded, and the tensor-parallel logits must be the 1-GPU logits (summation order only)
                if cand is not None and rank == 0 and ref1 and "prefill_logits" in ref1:
                    d = float(np.abs(cand["prefill_logits"] - ref1["prefill_logits"]).max())
                    tol = 2e-3 * max(1.0, float(ref1["prefill_logits"].std()))
                    cand["max_abs_logit_diff_vs_1gpu"] = d
                    if not d <= tol:
                        err = f"tensor-parallel logits differ from the 1-GPU logits by {d:.3g} (tolerance {tol:.3g})"
                    # ... and the first greedy ids (16-step graphs, argmax exchange, logits gather) must be the 1-GPU ids: the
                    # cross-device wire has never been validated by a test on this repo's 1-GPU pool
                    elif list(cand.get("head_ids", []))[:8] != list(ref1.get("head_ids", []))[:8]:
                        err = f"tensor-parallel greedy ids {cand.get('head_ids', [])[:8]} differ from the 1-GPU ids {ref1.get('head_ids', [])[:8]}"
                failed = rdv.max_over_ranks(1.0 if (cand is None or err) else 0.0, tag="vote:p2p") > 0
                if failed:
                    notes.append({"transport": "p2p", "error": err or "another rank failed"})
                    continue
                tp_res, transport = cand, "push all-reduce over xGMI (hipIpc-mapped receive slots, 8-byte tagged granules)"
                break
            else:
                # RCCL can hang where the push path merely times out: run it in guarded child processes
                child = spawn_rccl_child(args)
                res = collect_child(child)
                rdv.barrier()
                if rank == 0 and res and "tok_s" in res:
                    tp_res, transport = res, "RCCL all-reduce / all-gather"
                ok = rdv.max_over_ranks(0.0 if (rank != 0 or tp_res) else 1.0, tag="vote:rccl") == 0
                if not ok:
                    notes.append({"transport": "rccl", "error": (res or {}).get("error", "child failed")})
                    tp_res = None
    if watchdog is not None:
        watchdog.cancel()
    if rank == 0:
        if not args.no_secondary:
            sec = {"metric": f"decode tokens/sec, big Q4_0 single-stream greedy, tensor-parallel over {n} GPUs", "scaling": "strong"}
            if tp_res is None:
                sec.update(value=None, error="no tensor-parallel transport worked", attempts=notes)
            else:
                sec.update(config={"workload": workload_text(shape, "big", "q4_0", tp_res, min(args.steps, 128)), "parallelism": f"tp{n}", "allreduce": transport,
                                   "weights": "q4_0 blocks dequantised in-register, f32 activations and KV cache, rows/columns sharded per rank"},
                           algorithmic_bytes_per_step=tp_res["step_bytes"], last_ids=tp_res["last_ids"], **summary(tp_res))
                if "device_ms_per_step" in tp_res:
                    sec["device_ms_per_step"] = round(tp_res["device_ms_per_step"], 5)
                if tp_res.get("p2p"):
                    sec["config"]["p2p_area"] = tp_res["p2p"]
                if "max_abs_logit_diff_vs_1gpu" in tp_res:
                    sec["max_abs_logit_diff_vs_1gpu"] = tp_res["max_abs_logit_diff_vs_1gpu"]
                if notes:
                    sec["attempts"] = notes
            if ref1 and "tok_s" in ref1:
                same = 0
                for a, b in zip(ref1["head_ids"], (tp_res or {}).get("head_ids", [])):
                    if a != b:
                        break
                    same += 1
                sec["reference_1gpu"] = dict(workload="the same big Q4_0 model on rank 0's GPU alone, same run",
                                             greedy_ids_equal_prefix=f"{same}/{len(ref1['head_ids'])}", **summary(ref1, keys=()))
                if tp_res is not None:
                    sec["speedup_vs_1gpu"] = round(tp_res["tok_s"] / ref1["tok_s"], 3)
            elif ref1:
                sec["reference_1gpu"] = ref1
            out["secondary"] = sec
        print(json.dumps(out))
    rdv.close()


def spawn_rccl_child(args):
    env = dict(os.environ)
    env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 17)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.pop("TORCHELASTIC_RUN_ID", None)
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(args.gpus), "--rccl-worker", "1", "--steps", str(args.steps),
           "--warmup", str(args.warmup), "--no-cpu-baseline"]
    return subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def collect_child(proc, timeout_s=300):
    timeout_s = int(os.environ.get("NL_TP_CHILD_TIMEOUT", timeout_s))
    try:
        so, se = proc.communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        proc.kill()
        proc.communicate()
        return {"error": f"tensor-parallel child did not finish within {timeout_s}s"}
    for line in so.splitlines():
        if line.startswith("TPJSON "):
            return json.loads(line[7:])
    return {"error": "tensor-parallel child failed", "rc": proc.returncode, "stderr_tail": se[-600:]} if proc.returncode else {}


if __name__ == "__main__":
    main()
