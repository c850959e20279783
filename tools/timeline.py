"""Developer tool: decode a short run (for rocprofv3 --kernel-trace), or analyse the trace it left.
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o t -- python3 tools/timeline.py run nano q8_0 [shard_of]
  python3 tools/timeline.py show gpurun_out/tl/*kernel_trace.csv
`show` prints, for the last 70 launches, start and end relative to the predecessor's end: a negative start = overlap."""
import os, sys, csv
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def run(tier, wtype, shard_of=0):
    from nanollama_amd import gguf, model, synth
    path = os.path.join(os.environ.get("NL_BENCH_DIR", "/tmp"), f"nl_bench_{tier}_{wtype}_qrand.gguf")
    if not os.path.exists(path):
        synth.generate_gguf(path, synth.TIERS[tier], wtype, mode="qrand")
    # shard_of N: rank 0's shard of a tensor-parallel group of N, alone (nl_p2p_loopback): the two-launch layers of nl_tp.h
    kw = dict(tp_rank=0, tp_size=shard_of, p2p_loopback=True) if shard_of else {}
    dev = model.load_llama_model(gguf.load_gguf(path), **kw)
    ids = dev.decode_greedy(5, 0, 64)
    dev.synchronize()
    print(ids[:8])
    dev.close()

def show(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-70:]
    prev_end = None
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].split("<")[0].split("(")[0][-28:]
        if prev_end is not None:
            print(f"{name:28s} start {0.001*(s-prev_end):+7.2f} us  dur {0.001*(e-s):6.2f} us  q={r.get('Queue_Id','?')}")
        prev_end = e if prev_end is None else max(prev_end, e)

if __name__ == "__main__":
    if sys.argv[1] == "run": run(sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 0)
    else: show(sys.argv[2])
