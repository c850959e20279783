"""One rank of a tensor-parallel run over the push all-reduce (nl_p2p_*), started by tests/test_gpu_p2p.py.

Several ranks share GPU 0 of the test box (NL_BENCH_ONE_DEVICE): the hipIpc mapping, the granule protocol, the
reduce kernel, the argmax exchange and the logits gather are exactly what runs with one rank per GPU; only the
wire (local HBM instead of xGMI) differs.  Rank 0 stores what it computed for the parent to compare.

    RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the environment; argv: gguf path, out .npz, n_tokens, n_greedy
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from nanollama_amd import gguf, model, synth  # noqa: E402
from nanollama_amd.dist import Rendezvous  # noqa: E402


def main():
    path, out, n_tok, n_greedy = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    rdv = Rendezvous(timeout_s=60.0)
    g = gguf.load_gguf(path)
    dev = model.load_llama_model(g, device=rdv.local_rank, tp_rank=rdv.rank, tp_size=rdv.world,
                                 p2p_allgather=rdv.allgather_bytes)
    toks = synth.prompt_ids(n_tok, g.meta.vocab_size, seed=9)
    want_mode = os.environ.get("NL_EXPECT_FUSED_MODE")
    if want_mode is not None:
        assert dev.plan_info()["fused_mode"] == int(want_mode), dev.plan_info()
    rdv.barrier()
    logits = []
    for pos, t in enumerate(toks):
        dev.forward(t, pos)
        logits.append(dev.state.logits.copy())
    first = int(np.argmax(logits[-1]))
    ids = dev.decode_greedy(first, n_tok, n_greedy)            # 16-step graphs + single steps, argmax exchange
    again = dev.decode_greedy(first, n_tok, n_greedy)          # replay from the same prefix: identical
    dev.prefill(toks[:5])                                       # token-at-a-time plan under tensor parallelism
    pre = dev.state.logits.copy()
    sampled = []
    n_samp = int(os.environ.get("NL_P2P_SAMPLED", "0"))
    if n_samp:
        # the sampled loop reads the GATHERED logits on every rank (the LM-head slices the peers pushed): same uniforms,
        # so every rank must pick the same ids -- a stale slice on one rank would show up here
        us = np.random.default_rng(5).random(n_samp, dtype=np.float32)
        sampled, _ = dev.sample_decode(5, n_samp, 0.8, 0.9, 50, 1.15, 16, us, [])
    rdv.barrier()
    if os.environ.get("NL_EXPECT_RETIRED"):      # a fused-launch give-up on ANY rank retires the plan on EVERY rank
        assert dev.plan_info()["fused_mode"] == 0, (rdv.rank, dev.plan_info())
    elif want_mode is not None:                  # ... and otherwise the plan the test asked for is still the one in use
        assert dev.plan_info()["fused_mode"] == int(want_mode), (rdv.rank, dev.plan_info(), dev.last_error())
    if rdv.rank == 0:
        np.savez(out, logits=np.stack(logits), ids=np.array(ids), again=np.array(again), pre=pre, toks=np.array(toks),
                 sampled=np.array(sampled, np.int32))
    # every rank must hold the same gathered logits and ids: compare through the star
    mine = np.stack(logits).tobytes() + np.array(ids, np.int32).tobytes() + np.array(sampled, np.int32).tobytes()
    parts = rdv.allgather_bytes(mine)
    assert all(p == parts[0] for p in parts), "ranks disagree on the gathered logits / greedy ids"
    dev.close()
    rdv.close()
    print(f"rank {rdv.rank} ok")


if __name__ == "__main__":
    main()
