"""GPU tests of the one-process tensor-parallel group (nl_create_group, include/nanollama_hip.h): the drop-in host's way to shard
a model over the GPUs of a node (go/main.go:63 LoadLlamaModel is one process; `--gpus N` of the CLI / server, NANOLLAMA_GPUS of
the cgo shim).  The 1-GPU test pool runs it with every rank on device 0: rank engines, worker threads, peer-pointer receive
areas, granule protocol, owner-side reduction, argmax exchange and logits gather are what runs with one rank per GPU; only
the wire (local HBM instead of xGMI) differs.  Held BITWISE to the in-process shard group (same shards, same summation order)
and to the oracle within the engine's tolerance."""
import os
import subprocess
import sys

import numpy as np
import pytest

from nanollama_amd import gguf, synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOGIT_TOL = 1e-4


@pytest.fixture(scope="module")
def hip():
    from nanollama_amd import _lib, model
    if _lib.lib().nl_device_count() < 1:
        pytest.fail("no HIP device visible: the GPU tests must run on the MI355X box")
    return model


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, {root!r})
from nanollama_amd import gguf, model, synth
from oracle import oracle as orc
n, plan, path = int(sys.argv[1]), sys.argv[2], sys.argv[3]
LOGIT_TOL = 1e-4
g = gguf.load_gguf(path)
vocab = g.meta.vocab_size
grp = model.load_llama_model(g, devices=[0] * n)
assert grp.plan_info()["fused_mode"] == (0 if plan == "five_launch" else 3), grp.plan_info()
assert grp.p2p_info()["push_allreduce"]
local = model.LocalTPGroup(g, n, fused=(plan == "two_launch"))
ref = orc.OracleModel(g)
toks = synth.prompt_ids(12, vocab, seed=9)
worst = 0.0
for pos, t in enumerate(toks):
    grp.forward(t, pos)
    want_bits = local.forward(t, pos)
    assert grp.state.logits.tobytes() == want_bits.tobytes(), f"pos {{pos}}: one-process group != in-process shard group"
    want = ref.forward(t, pos)
    worst = max(worst, float(np.abs(grp.state.logits - want).max()) / max(1.0, float(want.std())))
assert worst <= LOGIT_TOL, worst
# chained greedy decode through the 16-step graphs and the argmax exchange: the oracle's ids, and a replay repeats them
tok, want_ids = int(np.argmax(grp.state.logits)), []
first = tok
for k in range(24):
    tok = int(orc.argmax(ref.forward(tok, len(toks) + k)))
    want_ids.append(tok)
ids = grp.decode_greedy(first, len(toks), 24)
assert ids == want_ids, (ids, want_ids)
assert grp.decode_greedy(first, len(toks), 24) == ids
# per-call argmax, prefill, reset, sampled decode from the gathered logits, memory accounting: the handle is an ordinary one
assert grp.forward_argmax(first, len(toks)) == want_ids[0]
grp.reset()
grp.prefill(toks[:5])
ref.reset()
for pos, t in enumerate(toks[:5]):
    want = ref.forward(t, pos)
assert np.abs(grp.state.logits - want).max() <= LOGIT_TOL * max(1.0, float(want.std()))
us = np.random.default_rng(5).random(12, dtype=np.float32)
sampled, recent = grp.sample_decode(5, 12, 0.8, 0.9, 50, 1.15, 16, us, [])
assert len(sampled) == 12 and all(0 <= t < vocab for t in sampled)
mem = grp.memory_usage()
assert mem["weights"] > 0 and mem["kv_cache"] > 0
assert grp.last_error() == "", grp.last_error()
grp.close(); local.close(); ref.close()
print(f"group ok: {{n}} ranks on device 0, {{plan}} plan: bitwise = in-process shard group; max|gpu-oracle| = {{worst:.2e}}")
"""


@pytest.mark.parametrize("n,plan", [(2, "five_launch"), (4, "five_launch"), (2, "two_launch"), (4, "two_launch")])
def test_one_process_group_equals_the_shard_group_bitwise(tmp_path, n, plan):
    # (a fresh process with GPU_MAX_HW_QUEUES >= ranks: the ranks share ONE device here, and a rank's launch spins until its
    #  peers' launches run -- two rank streams on one hardware queue would wait for each other.  One rank per GPU has its own
    #  device's queues and needs no such setting.)
    shape = synth.ModelShape("grp_probe", 3, 1024 if n == 2 else 512, 16 if n == 2 else 8, 4, 4096, seq_len=64, interm=2048 if n == 2 else 1024)
    p = str(tmp_path / "m.gguf")
    synth.generate_gguf(p, shape, "q4_0", 61, mode="qrand")
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8", NL_P2P_TIMEOUT_MS="5000")
    if plan == "five_launch":
        env["NL_FUSED_ATTN"] = "0"       # (the plain shard group steps the five-launch plan: hold the ranks to it)
    r = subprocess.run([sys.executable, "-c", WORKER.format(root=ROOT), str(n), plan, p], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "group ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
    print("\n" + r.stdout.strip().splitlines()[-1])


def test_group_of_eight_ranks_in_a_fresh_process(tmp_path):
    # 8 ranks on ONE device need 8 hardware queues (a rank's launch spins until its peers' launches run): GPU_MAX_HW_QUEUES must
    # be set before the HIP runtime starts, hence the child process.  One rank per GPU has its own device's queues.
    shape = synth.ModelShape("grp8", 2, 512, 8, 8, 2048, seq_len=64, interm=1024)
    p = str(tmp_path / "m.gguf")
    synth.generate_gguf(p, shape, "q4_0", 67, mode="qrand")
    code = f"""
import sys, numpy as np
sys.path.insert(0, {ROOT!r})
from nanollama_amd import gguf, model, synth
g = gguf.load_gguf({p!r})
grp = model.load_llama_model(g, devices=[0] * 8)
loc = model.LocalTPGroup(g, 8)
for pos, t in enumerate(synth.prompt_ids(8, 2048, seed=3)):
    grp.forward(t, pos)
    assert grp.state.logits.tobytes() == loc.forward(t, pos).tobytes(), pos
ids = grp.decode_greedy(int(np.argmax(grp.state.logits)), 8, 20)
assert len(ids) == 20 and grp.last_error() == "", grp.last_error()
grp.close(); loc.close()
print("group of 8 ok")
"""
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8", NL_FUSED_ATTN="0", NL_P2P_TIMEOUT_MS="5000")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and "group of 8 ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_group_takes_gamma_and_decode_batches_rank_by_rank(tmp_path):
    # nl_set_gamma on a group handle swaps the embedding table of one rank after the other (allocations, frees and graph
    # captures must not overlap another rank thread's capture); nl_forward_batch on a group steps every stream through the rank
    # engines.  Both against the oracle (go/model.go:503-505 gamma, :510-612 per stream), gamma on and off again.
    shape = synth.ModelShape("grp_gamma", 2, 512, 8, 4, 1024, seq_len=64, interm=1024)
    p = str(tmp_path / "m.gguf")
    synth.generate_gguf(p, shape, "q4_0", 73, mode="qrand")
    code = f"""
import sys, numpy as np
sys.path.insert(0, {ROOT!r})
from nanollama_amd import gguf, model, synth
from oracle import oracle as orc
g = gguf.load_gguf({p!r})
grp = model.load_llama_model(g, devices=[0, 0], max_streams=4)
rng = np.random.Generator(np.random.PCG64(3))
idx = np.array([5, 9, 700], dtype=np.int32)
vals = (rng.standard_normal((3, 512)) * 0.3).astype(np.float32)
toks = [5, 11, 9, 700, 3]
def run(with_gamma):
    refs = [orc.OracleModel(g) for _ in range(3)]
    if with_gamma:
        for r in refs: r.set_gamma(idx, vals)
    worst = 0.0
    grp.reset()
    for s in range(1, 3): grp.reset(stream=s)
    for pos, t in enumerate(toks):
        grp.forward(t, pos)
        want = refs[0].forward(t, pos)
        worst = max(worst, float(np.abs(grp.state.logits - want).max()) / max(1.0, float(want.std())))
    # three streams stepped together (stream 0 continues, 1 and 2 start): every stream's logits
    for k in range(3):
        st, tk, ps = [0, 1, 2], [toks[k], toks[k + 1], 700 if k == 0 else 9], [len(toks) + k, k, k]
        ids, lg = grp.forward_batch(st, tk, ps, want_logits=True)
        for j in range(3):
            want = refs[j].forward(tk[j], ps[j])
            worst = max(worst, float(np.abs(lg[j] - want).max()) / max(1.0, float(want.std())))
            assert ids[j] == int(orc.argmax(want)), (k, j)
    for r in refs: r.close()
    return worst
w0 = run(False)
grp.set_gamma(idx, vals)
w1 = run(True)
grp.set_gamma([], np.zeros((0, 512), np.float32))
w2 = run(False)
assert max(w0, w1, w2) <= 1e-4, (w0, w1, w2)
assert grp.last_error() == "", grp.last_error()
grp.close()
print(f"group gamma + batches ok: {{w0:.2e}} {{w1:.2e}} {{w2:.2e}}")
"""
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8", NL_P2P_TIMEOUT_MS="5000")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "group gamma + batches ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
    print("\n" + r.stdout.strip().splitlines()[-1])


def test_group_rejects_what_it_cannot_shard(hip, tmp_path):
    from nanollama_amd import _lib
    shape = synth.ModelShape("grp_bad", 2, 256, 4, 2, 512, seq_len=32, interm=512)      # 2 kv heads: no 4-way shard
    p = tmp_path / "m.gguf"
    synth.generate_gguf(str(p), shape, "q8_0", 69)
    g = gguf.load_gguf(str(p))
    with pytest.raises(_lib.NlError):
        hip.load_llama_model(g, devices=[0, 0, 0, 0])
    with pytest.raises(_lib.NlError):
        hip.load_llama_model(g, devices=[0, 0, 0])         # 3 ranks: not a supported group size
    with pytest.raises(_lib.NlError):
        hip.load_llama_model(g, devices=[0, 99])           # no such device


def test_cli_shards_over_gpus(tmp_path):
    # `python -m nanollama_amd --model m.gguf --gpus 2` on the one-GPU box: --device 0 + 2 ranks needs devices 0 and 1 -> refused
    # loudly; the same two ranks pinned to device 0 (NL_GROUP_ONE_DEVICE, the test pool's stand-in) generate text end to end
    shape = synth.ModelShape("grp_cli", 2, 256, 4, 4, 512, seq_len=64, interm=512)
    p = str(tmp_path / "m.gguf")
    synth.generate_gguf(p, shape, "q8_0", 71)
    env = dict(os.environ, NL_GROUP_ONE_DEVICE="1", NL_QUIET="1", GPU_MAX_HW_QUEUES="8")
    r = subprocess.run([sys.executable, "-m", "nanollama_amd", "--model", p, "--gpus", "2", "--prompt", "ab", "--max-tokens", "8", "--temp", "0",
                        "--rep-penalty", "1.0"], env=env, capture_output=True, text=True, timeout=240, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert "tensor-parallel over devices [0, 0]" in r.stdout
    one = subprocess.run([sys.executable, "-m", "nanollama_amd", "--model", p, "--prompt", "ab", "--max-tokens", "8", "--temp", "0",
                          "--rep-penalty", "1.0"], env=env, capture_output=True, text=True, timeout=240, cwd=ROOT)
    assert one.returncode == 0
    assert r.stdout.strip().splitlines()[-1] == one.stdout.strip().splitlines()[-1]      # the same generated text
