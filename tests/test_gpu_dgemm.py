"""GPU tests of the decode-batch GEMM path (nanollama_amd/csrc/nl_dgemm.h): short multi-token steps (batches of decode streams,
short prompts) of Q4_0 models whose K dimensions are whole 256-column groups run five launches per layer -- Q|K|V + RoPE + KV
store, attention, WO + residual + folded norm, gate || up + SwiGLU, down + residual + folded norm -- instead of eight.
Reference: go/model.go:510-612 per stream (the layer), go/quant.go:45-94 (MatMulQ4_0), go/quant.go:597-607 (RMSNorm).
Every shape class of the kernel's instantiations (blocks per wavefront 1, 2, 3, 4, 6, 8; 8 and 16 wavefronts), biases,
conjugate RoPE, ragged batches (token tiles that are not full), against the oracle; the split-K launches of the same handle
(NL_DGEMM=0) are the second witness."""
import os
import sys

import numpy as np
import pytest

from nanollama_amd import gguf, synth

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

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


def _rel(a, b):
    return float(np.abs(a - b).max()) / max(1.0, float(b.std()))


SHAPES = {
    # D 256 -> 8 blocks (1 per wavefront), I 768 -> 24 blocks (3 per wavefront); GQA 2, attention biases
    "k256_i768_bias": synth.ModelShape("dg_a", 2, 256, 4, 2, 640, seq_len=160, interm=768, attn_bias=True),
    # D 512 -> 16 blocks (2 per wavefront), I 1024 -> 32 (4); conjugate RoPE, MHA, head_dim 64
    "k512_i1024_conj": synth.ModelShape("dg_b", 2, 512, 8, 8, 512, seq_len=96, interm=1024, rope_conjugate=True),
    # D 1536 -> 48 (6 per wavefront), I 4096 -> 128 (16 wavefronts x 8): goldie's K dimensions, one layer, GQA 4
    "goldie_k": synth.ModelShape("dg_c", 1, 1536, 24, 6, 512, seq_len=64, interm=4096),
    # D 1024 -> 32 (4 per wavefront), I 3072 -> 96 (16 wavefronts x 6), head_dim 32, GQA 8
    "k1024_i3072_hd32": synth.ModelShape("dg_d", 1, 1024, 32, 4, 512, seq_len=64, interm=3072),
    # I 2048 -> 64 (8 wavefronts x 8)
    "k1024_i2048": synth.ModelShape("dg_e", 1, 1024, 16, 8, 512, seq_len=64, interm=2048),
}


@pytest.mark.parametrize("name", list(SHAPES))
def test_decode_batches_on_dgemm_match_oracle(hip, orc, tmp_path, monkeypatch, name):
    shape = SHAPES[name]
    p = tmp_path / "m.gguf"
    synth.generate_gguf(str(p), shape, "q4_0", 211)
    g = gguf.load_gguf(str(p))
    ns, nsteps = 37, 4                                   # 37 streams: two full token tiles and a ragged third
    rng = np.random.Generator(np.random.PCG64(5))
    start = [int(v) for v in rng.integers(0, 20, size=ns)]
    start[0], start[36] = 0, 19
    seqs = [[int(t) for t in rng.integers(3, shape.vocab, size=start[s] + nsteps)] for s in range(ns)]
    check = [0, 15, 16, 31, 32, 36]                      # both edges of every token tile
    orc.set_threads(min(16, os.cpu_count() or 1))
    refs = {}
    for s in check:
        ref = orc.OracleModel(g)
        lg = [ref.forward(t, pos).copy() for pos, t in enumerate(seqs[s])]
        refs[s] = lg[start[s]:]
        ref.close()
    orc.set_threads(1)
    outs = {}
    for knob in ("1", "0"):
        monkeypatch.setenv("NL_DGEMM", knob)
        dev = hip.load_llama_model(g, max_streams=ns)
        w0 = dev.memory_usage()["weights"]
        for s in range(ns):
            for pos in range(start[s]):
                dev.forward(seqs[s][pos], pos, stream=s)
        got, worst = [], 0.0
        for k in range(nsteps):
            ids, lg = dev.forward_batch(list(range(ns)), [seqs[s][start[s] + k] for s in range(ns)],
                                        [start[s] + k for s in range(ns)], want_logits=True)
            got.append(lg.copy())
            for s in check:
                worst = max(worst, _rel(lg[s], refs[s][k]))
                top2 = np.partition(refs[s][k], -2)[-2:]
                if float(top2[1] - top2[0]) > 10 * LOGIT_TOL * max(1.0, float(refs[s][k].std())):
                    assert ids[s] == int(np.argmax(refs[s][k])), (knob, s, k)
        print(f"\n{name}, NL_DGEMM={knob}: max|gpu-oracle| = {worst:.2e}")
        assert worst <= LOGIT_TOL, (knob, worst)
        # the path taken: the first multi-token step builds dgemm's block-major weight copies
        grew = dev.memory_usage()["weights"] - w0
        assert (grew > 0) == (knob == "1"), (knob, grew)
        outs[knob] = np.stack(got)
        if knob == "1":
            # a replay from a reset is bit-identical (fixed summation orders; the step graph is the cached one)
            dev.reset()
            for s in range(ns):
                for pos in range(start[s]):
                    dev.forward(seqs[s][pos], pos, stream=s)
            for k in range(nsteps):
                _, lg = dev.forward_batch(list(range(ns)), [seqs[s][start[s] + k] for s in range(ns)],
                                          [start[s] + k for s in range(ns)], want_logits=True)
                assert lg.tobytes() == outs["1"][k].tobytes(), k
        dev.close()
    # the two GEMM families agree far inside the tolerance (different summation orders over the blocks)
    scale = max(1.0, float(outs["0"].std()))
    assert float(np.abs(outs["1"] - outs["0"]).max()) <= 2e-5 * scale


def test_short_prompt_and_split_attention_on_dgemm(hip, orc, tmp_path):
    # a 40-token prompt is one multi-token step on dgemm (causal tile attention inside), then decode batches of 5 streams that
    # cross the 128-position split of the attention launch (merge launch behind it): logits against the oracle throughout
    shape = synth.ModelShape("dg_p", 2, 256, 4, 2, 512, seq_len=192, interm=512)
    p = tmp_path / "m.gguf"
    synth.generate_gguf(str(p), shape, "q4_0", 223)
    g = gguf.load_gguf(str(p))
    rng = np.random.Generator(np.random.PCG64(9))
    ns = 5
    start = [40, 126, 127, 128, 150]
    seqs = [[int(t) for t in rng.integers(3, shape.vocab, size=start[s] + 4)] for s in range(ns)]
    dev = hip.load_llama_model(g, max_streams=ns)
    orc.set_threads(min(16, os.cpu_count() or 1))
    refs = []
    for s in range(ns):
        ref = orc.OracleModel(g)
        lg = [ref.forward(t, pos).copy() for pos, t in enumerate(seqs[s])]
        refs.append(lg)
        ref.close()
    orc.set_threads(1)
    worst = 0.0
    for s in range(ns):
        dev.prefill(seqs[s][:start[s]], stream=s)          # (stream 0: 40 tokens = one dgemm step; the others: 64-token steps)
        worst = max(worst, _rel(dev.state.logits, refs[s][start[s] - 1]))
    assert dev.memory_usage()["weights"] > 0
    for k in range(4):
        ids, lg = dev.forward_batch(list(range(ns)), [seqs[s][start[s] + k] for s in range(ns)],
                                    [start[s] + k for s in range(ns)], want_logits=True)
        for s in range(ns):
            worst = max(worst, _rel(lg[s], refs[s][start[s] + k]))
            assert ids[s] == int(np.argmax(refs[s][start[s] + k])), (s, k)
    print(f"\nshort prompt + split attention on dgemm: max|gpu-oracle| = {worst:.2e}")
    assert worst <= LOGIT_TOL
    dev.close()


def test_lm_head_of_a_decode_batch_on_dgemm(hip, orc, tmp_path, monkeypatch):
    # a Q4_0 LM head of whole 16-row tiles: final norm folded into the last down launch, logits + argmax candidates from the GEMM's
    # epilogue (go/model.go:616-619, go/main.go:400-408); NL_DGEMM_HEAD=0 and a vocabulary that does not qualify (520 rows) keep the
    # bnorm + split-K + argmax launches.  Logits against the oracle, ids = the argmax of the returned logits (ties: lowest index)
    # (9280 rows x 64 streams: 145 row groups over 64 resident workgroups per token tile -- two or three iterations each, the
    #  double-buffered path; 640 rows x 21 streams: one iteration, a ragged token tile)
    for vocab, ns in ((640, 21), (520, 21), (9280, 64)):
        shape = synth.ModelShape(f"dg_h{vocab}", 2, 256, 4, 2, vocab, seq_len=64, interm=512)
        p = tmp_path / f"h{vocab}.gguf"
        synth.generate_gguf(str(p), shape, "q4_0", 229)
        g = gguf.load_gguf(str(p))
        rng = np.random.Generator(np.random.PCG64(13))
        toks = [[int(t) for t in rng.integers(3, vocab, size=3)] for _ in range(ns)]
        orc.set_threads(min(16, os.cpu_count() or 1))
        refs = []
        for s in (0, 15, 16, ns - 1):
            ref = orc.OracleModel(g)
            refs.append((s, [ref.forward(t, pos).copy() for pos, t in enumerate(toks[s])]))
            ref.close()
        orc.set_threads(1)
        outs = {}
        for knob in ("1", "0"):
            monkeypatch.setenv("NL_DGEMM_HEAD", knob)
            dev = hip.load_llama_model(g, max_streams=ns)
            got = []
            for k in range(3):
                ids, lg = dev.forward_batch(list(range(ns)), [toks[s][k] for s in range(ns)], [k] * ns, want_logits=True)
                assert [int(i) for i in ids] == [int(np.argmax(lg[s])) for s in range(ns)], (vocab, knob, k)
                for s, ref in refs:
                    assert _rel(lg[s], ref[k]) <= LOGIT_TOL, (vocab, knob, k, s)
                got.append(lg.copy())
            outs[knob] = np.stack(got)
            dev.close()
        same = outs["1"].tobytes() == outs["0"].tobytes()
        assert same == (vocab == 520), vocab            # (the qualifying head sums its blocks in another order: close, not identical)
        assert float(np.abs(outs["1"] - outs["0"]).max()) <= 2e-5 * max(1.0, float(outs["0"].std()))
    monkeypatch.delenv("NL_DGEMM_HEAD")


def test_lm_head_on_dgemm_at_every_token_tile_count(hip, tmp_path, monkeypatch):
    # 3 .. 64 streams (one to four token tiles, full and ragged; the grid's row-group stride and the iterations per workgroup
    # change with the tile count): logits of the resident-workgroup LM head against the split-K launches of the same handle
    # state, ids = argmax of the returned logits
    shape = synth.ModelShape("dg_hn", 1, 256, 4, 2, 5136, seq_len=32, interm=512)       # 321 row tiles: ragged last row group
    p = tmp_path / "hn.gguf"
    synth.generate_gguf(str(p), shape, "q4_0", 233)
    g = gguf.load_gguf(str(p))
    rng = np.random.Generator(np.random.PCG64(17))
    for ns in (3, 15, 16, 17, 32, 33, 47, 48, 49, 64):
        toks = [int(t) for t in rng.integers(3, shape.vocab, size=ns)]
        outs = {}
        for knob in ("1", "0"):
            monkeypatch.setenv("NL_DGEMM_HEAD", knob)
            dev = hip.load_llama_model(g, max_streams=ns)
            ids, lg = dev.forward_batch(list(range(ns)), toks, [0] * ns, want_logits=True)
            assert [int(i) for i in ids] == [int(np.argmax(lg[s])) for s in range(ns)], (ns, knob)
            outs[knob] = lg.copy()
            dev.close()
        assert np.isfinite(outs["1"]).all(), ns
        assert float(np.abs(outs["1"] - outs["0"]).max()) <= 2e-5 * max(1.0, float(outs["0"].std())), ns
    monkeypatch.delenv("NL_DGEMM_HEAD")


def test_models_outside_dgemm_keep_the_split_k_launches(hip, tmp_path):
    # K not a whole number of 256-column groups, Q8_0 weights, QK-norm: the multi-token step keeps its split-K launches and
    # no second weight copy is built
    # (and a gate || up whose eight tiles x K / 32 blocks of nibbles do not fit the LDS: D 2048)
    for shape, wt in ((synth.ModelShape("dg_k192", 2, 192, 3, 3, 512, seq_len=64, interm=512), "q4_0"),
                      (synth.ModelShape("dg_k2048", 1, 2048, 32, 8, 512, seq_len=64, interm=1024), "q4_0"),
                      (synth.ModelShape("dg_q8", 2, 256, 4, 2, 512, seq_len=64, interm=512), "q8_0"),
                      (synth.ModelShape("dg_qkn", 2, 256, 4, 2, 512, seq_len=64, interm=512, qk_norm=True), "q4_0")):
        p = tmp_path / f"{shape.name}.gguf"
        synth.generate_gguf(str(p), shape, wt, 227)
        dev = hip.load_llama_model(gguf.load_gguf(str(p)), max_streams=8)
        w0 = dev.memory_usage()["weights"]
        ids, _ = dev.forward_batch(list(range(8)), [5] * 8, [0] * 8)
        assert len(ids) == 8 and dev.memory_usage()["weights"] == w0, shape.name
        dev.close()
