#!/usr/bin/env python3
"""bench.py -- decode tokens/sec + achieved-HBM-bandwidth fraction on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one decoded token: one pass of the hot path (Forward, go/model.go:490-620
restated as HIP kernels) plus the greedy argmax, chained on the device.

  N = 1 : BASELINE.json configs[1] -- nano (89M) Q8_0, single-stream greedy decode.
  N > 1 : BASELINE.json configs[4] -- big (7.9B) Q4_0, tensor-parallel over N GPUs
          (one process per GPU, RCCL all-reduce after WO and down), strong scaling.

Weights are a deterministic random-weight GGUF written on the box in the
reference exporter's layout (nanollama_amd.synth); inputs are resident in HBM
before the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from nanollama_amd import gguf, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PROMPT_LEN = 8
SEGMENT = 128  # greedy 128-token decode (BASELINE.json configs)


def kernel_bytes(shape: synth.ModelShape, wtype: str, pos: int, tp: int = 1):
    """Algorithmic HBM bytes per LAUNCH of each kernel kind:
    every weight byte once + the vectors the kernel must read/write."""
    t = synth.WTYPES[wtype]
    bpe = gguf.ggml_block_size(t) / gguf.ggml_block_elements(t)
    d, i, v, kv, hd = shape.dim, shape.ffn // tp, shape.vocab // tp, shape.kv_dim // tp, shape.head_dim
    hq = shape.n_head // tp * hd
    return {
        "embed": d * bpe + d * 4,
        "qkv_rope": (hq + 2 * kv) * d * bpe + 2 * d * 4 + (hq + 2 * kv) * 4,
        "attention": (pos + 1) * kv * 2 * 4 + hq * 4 * 2,
        "wo_resid": d * hq * bpe + hq * 4 + 2 * d * 4,
        "gate_up_swiglu": 2 * i * d * bpe + 2 * d * 4 + i * 4,
        "down_resid": d * i * bpe + i * 4 + 2 * d * 4,
        "lm_head": v * d * bpe + 2 * d * 4 + v * 4,
        "argmax": shape.vocab * 4,
    }


def ensure_gguf(shape, wtype, mode, rank=0):
    path = os.path.join(os.environ.get("NL_BENCH_DIR", "/tmp"), f"nl_bench_{shape.name}_{wtype}_{mode}.gguf")
    if rank == 0 and not os.path.exists(path):
        t0 = time.time()
        tmp = path + ".tmp"
        synth.generate_gguf(tmp, shape, wtype, mode=mode)
        os.replace(tmp, path)
        print(f"[bench] wrote {path} ({os.path.getsize(path) / 1e6:.1f} MB) in {time.time() - t0:.1f}s", file=sys.stderr)
    return path


def cpu_baseline(path, prompt, budget_s=20.0):
    """The Go engine's algorithm (C restatement, oracle/) timed on this box's
    host cores on a bounded sample of the same workload.  Reported, not optimised against."""
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
    n, t0 = 0, time.perf_counter()
    while n < SEGMENT and (time.perf_counter() - t0) < budget_s:   # timer after prefill, go/main.go:171
        m.forward(nxt, pos)
        nxt = oracle.argmax(m.logits())
        pos += 1
        n += 1
    dt = time.perf_counter() - t0
    m.close()
    return {"value": round(n / dt, 3), "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": f"{n} greedy decode tokens after an {len(prompt)}-token prefill, same GGUF, "
                      f"C restatement of go/quant.go+go/model.go with the Go row partition on {cores} threads (best of 1..64 on a {ncpu}-cpu host)"}


def run_workload(tier, wtype, rdv, steps, warmup, profile_pos, model, tp=True):
    """Load the tier's random-weight GGUF, run `warmup` + `steps` chained greedy decode steps, profile the
    launches.  tp=True: the ranks of rdv form one tensor-parallel engine; tp=False: every rank is an
    independent replica (no data-path collective).  Returns the result dict (timing = max over ranks)."""
    rank, local_rank = rdv.rank, rdv.local_rank
    replicas = 1 if tp else rdv.world
    world = rdv.world if tp else 1
    shape = synth.TIERS[tier]
    mode = "qrand" if tier in ("big", "goldie") else "float"
    comm_id = rdv.broadcast_bytes(model.comm_unique_id) if (world > 1 or os.environ.get("NL_FORCE_TP_PLAN")) else None
    path = ensure_gguf(shape, wtype, mode, rank)
    rdv.barrier()
    g = gguf.load_gguf(path)
    dev = model.load_llama_model(g, device=local_rank, tp_rank=rank if world > 1 else 0, tp_size=world,
                                 comm_id=comm_id)
    prompt = synth.prompt_ids(PROMPT_LEN, shape.vocab)
    dev.prefill(prompt)
    first, pos0 = int(np.argmax(dev.state.logits)), len(prompt)

    def run_steps(k):
        """k chained decode steps in segments of SEGMENT tokens; every segment restarts at pos0 on
        the still-valid prompt prefix of the KV cache, so no prefill is inside the loop."""
        done, ids = 0, []
        while done < k:
            seg = min(SEGMENT, k - done)
            ids = dev.decode_greedy(first, pos0, seg)
            done += seg
        return ids

    run_steps(warmup)
    dev.synchronize()
    rdv.barrier()
    dev.timer_start()
    t0 = time.perf_counter()
    ids = run_steps(steps)
    dev.synchronize()
    wall_ms = (time.perf_counter() - t0) * 1e3
    ev_ms = dev.timer_stop()
    rdv.barrier()
    wall_ms = rdv.max_over_ranks(wall_ms)
    ms_per_step = wall_ms / steps

    # per-launch device time: every launch of the plan replayed 20x back to back between HIP events on the
    # engine's stream (nl_profile_forward), at a mid-run position
    ppos = min(profile_pos, shape.seq_len - 1)
    prof = dev.profile_forward(first, ppos, iters=20)
    kb = kernel_bytes(shape, wtype, ppos, tp=world)
    kernels = {}
    for kind, (ms, calls) in prof.items():
        if calls:
            per = ms / calls
            kernels[kind] = {"launches": calls, "us_per_launch": round(per * 1e3, 3),
                             "GBps": round(kb[kind] / (per * 1e-3) / 1e9, 1)}
    traffic = None
    try:  # HBM bytes per launch measured with rocprofv3 PMC counters in a separate pass (profiles/README.md)
        tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
        traffic = tj.get(f"{tier}_{wtype}") if world == 1 else None
    except (OSError, ValueError):
        pass
    # dominant kernel = the kind that moves the most algorithmic bytes per step (stable from run to run; for the
    # launch-bound nano tier the five per-layer kinds have near-equal time shares and a time ranking flips)
    dom = max((k for k in kernels if k != "argmax"), key=lambda k: kb[k] * prof[k][1])
    dom_gbs = kernels[dom]["GBps"]
    mean_pos = pos0 + (min(SEGMENT, steps) - 1) / 2.0
    step_bytes = synth.weight_bytes_per_token(shape, wtype) + synth.kv_bytes_per_token(shape, int(mean_pos))
    step_gbs = step_bytes / (ms_per_step * 1e-3) / 1e9
    dev.close()
    return {
        "tier": tier, "wtype": wtype, "path": path, "prompt": prompt,
        "tok_s": replicas * steps / (wall_ms / 1e3), "ms_per_step": ms_per_step, "device_ms_per_step": ev_ms / steps,
        "replicas": replicas, "tp": world,
        "step_bytes": int(step_bytes), "hbm_frac_whole_step": step_gbs / (HBM_PEAK_GBS * max(world, 1)),
        "kernels": kernels, "last_ids": ids[-4:],
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": dom_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(dom_gbs / HBM_PEAK_GBS, 4), "traffic": (traffic or {}).get(dom),
                     "bytes_per_launch": int(kb[dom]), "us_per_launch": kernels[dom]["us_per_launch"]},
    }


def side_configs(model):
    """BASELINE.json configs[2] and [3] as side results of the single-GPU run: mini Q4_0 2047-token prefill
    (MFMA multi-token path) + one decode step, and goldie Q4_0 with 64 concurrent decode streams."""
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
                                     "decode_step_ms_at_pos_2047": round(dt1 * 1e3, 3)}
    dev.close()
    # -- goldie: 64 streams
    shape = synth.TIERS["goldie"]
    g = gguf.load_gguf(ensure_gguf(shape, "q4_0", "qrand"))
    ns, steps, pos0 = 64, 32, 8
    dev = model.load_llama_model(g, max_streams=ns)
    rng = np.random.Generator(np.random.PCG64(3))
    ids = [int(t) for t in rng.integers(3, shape.vocab, size=ns)]
    streams = list(range(ns))
    for p in range(pos0):
        ids, _ = dev.forward_batch(streams, ids, [p] * ns)
    t0 = time.perf_counter()
    for k in range(steps):
        ids, _ = dev.forward_batch(streams, ids, [pos0 + k] * ns)
    dt = time.perf_counter() - t0
    step_bytes = synth.weight_bytes_per_token(shape, "q4_0") + ns * synth.kv_bytes_per_token(shape, pos0 + steps // 2)
    out["goldie_q4_0_64_streams"] = {"tokens_per_s_aggregate": round(ns * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 3),
                                     "hbm_frac": round(step_bytes / (dt / steps) / 1e9 / HBM_PEAK_GBS, 4)}
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
    dev.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--workload", default=None, help="tier:wtype override, e.g. big:q4_0 (default by --gpus)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the big Q4_0 side measurement")
    ap.add_argument("--tp-worker", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--profile-pos", type=int, default=PROMPT_LEN + SEGMENT // 2)
    args = ap.parse_args()

    # the product library is loaded before torch so it binds /opt/rocm's HIP runtime
    from nanollama_amd import _lib, model
    from nanollama_amd.dist import Rendezvous
    _lib.lib()
    n = args.gpus
    tp_child = None
    if args.tp_worker is None and n > 1 and not args.workload and not args.no_secondary:
        # big Q4_0 tensor-parallel over the N GPUs runs in a child process per rank (own rendezvous port, hard
        # timeout): a collective that hangs must not take the headline line down with it
        tp_child = spawn_tp_child(args)
    rdv = Rendezvous()
    rank, world = rdv.rank, rdv.world
    if world != n and world != 1:
        raise SystemExit(f"--gpus {n} but WORLD_SIZE={world}")
    tp_res = None
    if tp_child is not None:
        # the child owns the GPU until it exits: parent and child never hold the device at the same time
        tp_res = collect_tp_child(tp_child)
        rdv.barrier()

    if args.tp_worker is not None:   # child: big Q4_0, tensor-parallel over all ranks
        r = run_workload("big", "q4_0", rdv, args.steps, args.warmup, args.profile_pos, model, tp=True)
        if rank == 0:
            print("TPJSON " + json.dumps({k: r[k] for k in ("tok_s", "ms_per_step", "hbm_frac_whole_step", "roofline",
                                                              "kernels", "tp")}))
        rdv.close()
        return

    if args.workload:
        tier, wtype = args.workload.split(":")
        tp = world > 1
    else:
        # N = 1: BASELINE configs[1], nano Q8_0.  N > 1: nano does not shard (SURVEY 8e "replicas only"), so the
        # same workload runs as N independent replicas (weak scaling, no collective); big Q4_0 over the N GPUs
        # (tensor parallel, RCCL all-reduce, strong scaling) is reported next to it under "secondary".
        tier, wtype, tp = "nano", "q8_0", False
    shape = synth.TIERS[tier]

    r = run_workload(tier, wtype, rdv, args.steps, args.warmup, args.profile_pos, model, tp=tp)
    par = f"tp{world}" if (tp and world > 1) else (f"{world} independent replicas" if world > 1 else "single-gpu")
    out = {
        "metric": f"decode tokens/sec, {tier} {wtype.upper()} single-stream greedy"
                  + (f", tensor-parallel over {world} GPUs" if tp and world > 1 else "")
                  + (f", {world} replicas (one per GPU)" if (not tp) and world > 1 else ""),
        "value": round(r["tok_s"], 2), "unit": "tokens/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(r["ms_per_step"], 5), "higher_is_better": True,
        "scaling": "strong" if (tp and world > 1) else "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic (random-weight GGUF, reference exporter layout, nanollama_amd.synth)",
        "config": {"workload": f"{tier} ({shape.matrix_params() / 1e6:.0f}M matrix params) {wtype.upper()} GGUF, "
                               f"{PROMPT_LEN}-token prompt + {SEGMENT}-token greedy decode segments, 1 stream per GPU",
                   "parallelism": par,
                   "weights": f"{wtype} blocks dequantised in-register, f32 activations and KV cache"},
        "device_ms_per_step": round(r["device_ms_per_step"], 5),
        "hbm_frac_whole_step": round(r["hbm_frac_whole_step"], 4),
        "algorithmic_bytes_per_step": r["step_bytes"],
        "roofline": r["roofline"], "kernels": r["kernels"], "last_ids": r["last_ids"],
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(r["path"], r["prompt"])
    if world == 1 and not args.workload and not args.no_secondary:
        # BASELINE.json's metric also names big Q4_0 @ 1 GPU: measured here as a side result (bandwidth-bound
        # regime; the headline nano config is launch-latency-bound)
        try:
            b = run_workload("big", "q4_0", rdv, 96, 16, args.profile_pos, model)
            out["secondary"] = {"workload": "big (7.9B) Q4_0 single-stream greedy decode, 1 GPU",
                                "value": round(b["tok_s"], 2), "unit": "tokens/s",
                                "ms_per_step": round(b["ms_per_step"], 5),
                                "hbm_frac_whole_step": round(b["hbm_frac_whole_step"], 4),
                                "roofline": b["roofline"], "kernels": b["kernels"]}
        except Exception as exc:  # the headline result must survive a failure of the side measurement
            out["secondary"] = {"error": repr(exc)}
        try:
            out["other_configs"] = side_configs(model)
        except Exception as exc:
            out["other_configs"] = {"error": repr(exc)}
    if tp_res is not None:
        res = tp_res
        if rank == 0:
            if "tok_s" in res:
                out["secondary"] = {"workload": f"big (7.9B) Q4_0 single-stream greedy decode, tensor-parallel over {n} GPUs "
                                                "(RCCL all-reduce after WO and down, logits all-gather)",
                                    "scaling": "strong", "value": round(res["tok_s"], 2), "unit": "tokens/s",
                                    "ms_per_step": round(res["ms_per_step"], 5),
                                    "hbm_frac_whole_step": round(res["hbm_frac_whole_step"], 4),
                                    "roofline": res["roofline"], "kernels": res["kernels"]}
            else:
                out["secondary"] = res
    if rank == 0:
        print(json.dumps(out))
    rdv.close()


def spawn_tp_child(args):
    import subprocess
    env = dict(os.environ)
    env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 17)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.pop("TORCHELASTIC_RUN_ID", None)
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(args.gpus), "--tp-worker", "1", "--steps", "96",
           "--warmup", "16", "--no-cpu-baseline"]
    return subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def collect_tp_child(proc, timeout_s=300):
    import subprocess
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
