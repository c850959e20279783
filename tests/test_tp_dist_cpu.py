"""CPU coverage of the N>1 path: the tensor-parallel slicing arithmetic (checked with the oracle's
GEMVs on sliced raw bytes) and the world_size-2 rendezvous bench.py uses under torch.distributed.run
(a torch-free TCP star: importing torch next to the HIP library loads a second HIP runtime)."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from nanollama_amd import gguf, quant, synth
import tp_plan as tp
from oracle import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("wtype", ["q4_0", "q8_0", "f16"])
@pytest.mark.parametrize("n", [2, 4, 8])
def test_column_and_row_slices_reproduce_the_full_gemv(wtype, n):
    t = synth.WTYPES[wtype]
    rng = np.random.Generator(np.random.PCG64(n * 7 + t))
    dim, heads, kv, hd, interm, vocab = 512, 8, 8, 64, 1024, 2048
    sl = tp.shard_slices(dim, heads, kv, hd, interm, vocab, n, 0)
    assert sl["attn_output.weight"].ncols == heads // n * hd and sl["ffn_down.weight"].ncols == interm // n
    # column-parallel (attn_output / ffn_down): the all-reduce of per-rank partials equals the full product
    for name, rows, cols in (("attn_output.weight", dim, heads * hd), ("ffn_down.weight", dim, interm)):
        w = (rng.random((rows, cols), dtype=np.float32) - 0.5) * 0.1
        raw = quant.encode(w, t)
        x = rng.standard_normal(cols, dtype=np.float32)
        full = oracle.matmul(raw, t, x, rows, cols)
        acc = np.zeros(rows, dtype=np.float32)
        for r in range(n):
            s = tp.shard_slices(dim, heads, kv, hd, interm, vocab, n, r)[name]
            part = tp.slice_raw(raw, t, rows, cols, s)
            acc += oracle.matmul(part, t, x[s.col0:s.col0 + s.ncols], s.nrows, s.ncols)
        assert np.abs(acc - full).max() <= 2e-5 * (1 + np.abs(full).max())
    # row-parallel (q/k/v, gate/up, LM head): concatenating the per-rank outputs IS the full output, bitwise
    for name, rows, cols in (("attn_q.weight", heads * hd, dim), ("ffn_gate.weight", interm, dim), ("output.weight", vocab, dim)):
        w = (rng.random((rows, cols), dtype=np.float32) - 0.5) * 0.1
        raw = quant.encode(w, t)
        x = rng.standard_normal(cols, dtype=np.float32)
        full = oracle.matmul(raw, t, x, rows, cols)
        parts = []
        for r in range(n):
            s = tp.shard_slices(dim, heads, kv, hd, interm, vocab, n, r)[name]
            parts.append(oracle.matmul(tp.slice_raw(raw, t, rows, cols, s), t, x, s.nrows, s.ncols))
        assert np.concatenate(parts).tobytes() == full.tobytes()


def _free_port_pair(start):
    """A port p such that p (the launcher's store) and p + 23 (nanollama_amd.dist's TCP star) can both be bound now."""
    import socket
    for p in range(start, start + 400):
        ok = True
        for q in (p, p + 23):
            s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            try:
                s.bind(("127.0.0.1", q))
            except OSError:
                ok = False
            finally:
                s.close()
        if ok:
            return p
    return start


def test_big_tier_divides_for_1_2_4_8_gpus():
    b = synth.TIERS["big"]
    for n in (1, 2, 4, 8):
        sl = tp.shard_slices(b.dim, b.n_head, b.n_kv_head, b.head_dim, b.ffn, b.vocab, n, n - 1)
        assert sl["ffn_down.weight"].ncols % 32 == 0 and sl["attn_q.weight"].nrows % 64 == 0
    assert tp.collectives_per_token(b.n_layer, b.dim, b.vocab, 8) == {"all_reduce_f32": (80, 16384), "all_gather_f32": (1, 48000)}
    with pytest.raises(ValueError):
        tp.shard_slices(b.dim, b.n_head, b.n_kv_head, b.head_dim, b.ffn, b.vocab, 3, 0)
    with pytest.raises(ValueError):
        tp.slice_raw(np.zeros(36, np.uint8), gguf.GGML_Q4_0, 1, 64, tp.Slice(0, 1, 16, 32))


def test_world_size_2_rendezvous(tmp_path):
    # bench.py multi-process scaffolding (id broadcast, barrier, max over ranks) under torch.distributed.run, CPU only
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        from nanollama_amd.dist import Rendezvous
        r = Rendezvous()
        ident = r.broadcast_bytes(lambda: bytes(range(128)))
        assert ident == bytes(range(128)) and r.world == 2
        r.barrier()
        m = r.max_over_ranks(10.0 + r.rank)
        assert m == 11.0, m
        parts = r.allgather_bytes(bytes([r.rank]) * (64 + r.rank))     # the hipIpc handle exchange of the push all-reduce
        assert parts == [bytes([0]) * 64, bytes([1]) * 65], parts
        assert "torch" not in sys.modules, "the rendezvous must not import torch"
        r.close()
        sys.stdout.write("rank %d ok\\n" % r.rank); sys.stdout.flush()      # one write: two ranks share the pipe
    """))
    out = None
    for attempt in range(3):  # a busy rendezvous port is the only expected flake
        port = _free_port_pair(29000 + (os.getpid() * 7 + attempt * 131) % 2000)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(port), str(script)]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
        if out.returncode == 0:
            break
    assert out.returncode == 0, out.stderr[-2000:]
    assert "rank 0 ok" in out.stdout and "rank 1 ok" in out.stdout


def test_a_rank_that_leaves_the_common_flow_is_detected(tmp_path):
    # rank 1 "fails" inside a phase and goes straight to the vote the ranks take afterwards, while rank 0 is still at
    # the phase's barrier: the messages carry (operation, sequence, tag), so both sides raise RendezvousDesync at once
    # instead of pairing the vote with the barrier (or waiting for the socket timeout)
    script = tmp_path / "d.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys
        sys.path.insert(0, {ROOT!r})
        from nanollama_amd.dist import Rendezvous, RendezvousDesync
        r = Rendezvous(timeout_s=30.0)
        r.barrier("start")
        try:
            if r.rank == 0:
                r.barrier("phase:after-load")
            else:
                r.max_over_ranks(1.0, tag="vote:p2p")
            print("rank", r.rank, "NOT DETECTED")
        except (RendezvousDesync, ConnectionError) as exc:
            print("rank", r.rank, "detected", type(exc).__name__)
        r.close()
    """))
    port = 29000 + (os.getpid() * 3 + 977) % 2000
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]
    assert "rank 0 detected RendezvousDesync" in outs[0][0], outs[0]
    assert "detected" in outs[1][0], outs[1]          # (rank 1 sees the desync reply, or rank 0 closing the socket first)
