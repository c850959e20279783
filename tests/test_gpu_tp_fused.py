"""GPU tests of the two-launches-per-layer tensor-parallel plan (nanollama_amd/csrc/nl_tp.h: tp_attn_kernel,
tp_ffn_kernel).  The kernels are exercised three ways:

  * as the shards of an IN-PROCESS group (LocalTPGroup(fused=True)): the same kernels, the group adds the partial
    vectors itself -- against the CPU oracle (stated tolerance) and the five-launch group plan, at test shapes and at
    the full 7.9B shape (tests/test_gpu_parity.py::test_big_full_shape_matches_oracle);
  * as rank PROCESSES sharing the box's one GPU over the push all-reduce (tests/test_gpu_p2p.py) -- bitwise against the
    in-process group (same kernels, same summation order);
  * alone in loopback (bench.py --shard-of N) for timing.
Reference lines: go/model.go:517-594 (attention half), :597-612 (feed-forward half).
"""
import os
import sys

import numpy as np
import pytest

from nanollama_amd import gguf, synth

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

LOGIT_TOL = 1e-4

SHAPES = {
    # name: (shape, tp sizes)
    "gqa4": (synth.ModelShape("tpf_gqa4", 3, 1024, 16, 4, 4096, seq_len=320, interm=2048), (2, 4)),   # 300 positions: two 256-position passes
    "mha8": (synth.ModelShape("tpf_mha8", 3, 512, 8, 8, 4096, seq_len=160, interm=1024), (2, 8)),
    # interm = 43 blocks of 32 per rank at tp 2 (the 7.9B tier's 1376 rows per rank at tp 8): a ragged last pair and group
    "ragged": (synth.ModelShape("tpf_ragged", 2, 1024, 16, 4, 4096, seq_len=160, interm=2752), (2,)),
    "qknorm_conj_bias": (synth.ModelShape("tpf_var", 2, 1024, 16, 4, 4096, seq_len=160, interm=2048, qk_norm=True,
                                          rope_conjugate=True, attn_bias=True), (4,)),
    # wide enough that a producer wavefront holds two column groups in pair mode (D = 4096: 16 groups)
    "wide": (synth.ModelShape("tpf_wide", 2, 4096, 64, 16, 2048, seq_len=160, interm=4096), (4,)),
    # 16 kv groups of 2 heads per rank at tp 2: a projection workgroup holds two tiles, a wavefront two column groups
    "two_tiles": (synth.ModelShape("tpf_two_tiles", 2, 4096, 64, 32, 2048, seq_len=160, interm=2048), (2,)),
    # a rank's shard of the feed-forward half too large for one-tile producers (256 gate tiles per rank at tp 2 > the compute
    # units' budget: the 7.9B tier at tp 2): its second launch is wide_ffn_kernel with the all-reduce seam in its tail -- four
    # rounds of gate / up tiles per workgroup (NL_WIDE_FFN=2 lets the 64-workgroup grid of this small width through)
    "wide_ffn": (synth.ModelShape("tpf_wide_ffn", 2, 1024, 16, 4, 4096, seq_len=160, interm=8192), (2,)),
}


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


@pytest.mark.parametrize("pair", ["auto", "pair", "split"])
@pytest.mark.parametrize("name,wtype", [("gqa4", "q4_0"), ("gqa4", "q8_0"), ("mha8", "q4_0"), ("ragged", "q4_0"),
                                        ("qknorm_conj_bias", "q8_0"), ("wide", "q4_0"), ("two_tiles", "q4_0"), ("wide_ffn", "q4_0"),
                                        ("wide_ffn", "q8_0")])
def test_two_launch_layer_in_process_group_matches_oracle(hip, orc, tmp_path, monkeypatch, name, wtype, pair):
    shape, sizes = SHAPES[name]
    if pair != "auto" and name not in ("gqa4", "ragged", "wide"):
        pytest.skip("producer geometry variants are covered on three shapes")
    p = tmp_path / "m.gguf"
    synth.generate_gguf(str(p), shape, wtype, 71, mode="qrand")
    g = gguf.load_gguf(str(p))
    if pair != "auto":
        monkeypatch.setenv("NL_TP_PAIR", "1" if pair == "pair" else "0")
    if name == "wide_ffn":
        monkeypatch.setenv("NL_WIDE_FFN", "2")
    ref = orc.OracleModel(g)
    orc.set_threads(min(16, os.cpu_count() or 1))
    toks = synth.prompt_ids(shape.seq_len - 20, shape.vocab, seed=29)      # (gqa4: crosses the 256-position pass boundary inside the launch)
    wants = [ref.forward(t, pos).copy() for pos, t in enumerate(toks)]
    orc.set_threads(1)
    ref.close()
    for n in sizes:
        fused = hip.LocalTPGroup(g, n, fused=True)
        info = fused.shards[0].plan_info()
        assert info["fused_mode"] == 3, info
        assert info["launches_fused"] == 2 * shape.n_layer + 3, info       # embed + 2 per layer + LM head + argmax
        plain = hip.LocalTPGroup(g, n)
        worst = gap = 0.0
        for pos, t in enumerate(toks):
            a = fused.forward(t, pos).copy()
            b = plain.forward(t, pos)
            worst = max(worst, float(np.abs(a - wants[pos]).max()) / max(1.0, float(wants[pos].std())))
            gap = max(gap, float(np.abs(a - b).max()))
            assert int(np.argmax(a)) == int(orc.argmax(wants[pos])) or float(np.sort(wants[pos])[-1] - np.sort(wants[pos])[-2]) < 1e-3
        print(f"\ntwo-launch layer {name}/{wtype}/{pair} tp{n}: max|gpu-oracle|={worst:.2e}, max|fused-general|={gap:.2e}")
        assert worst <= LOGIT_TOL
        # replay from a reset: bit-identical (fixed reduction orders, no atomics on values)
        again = hip.LocalTPGroup(g, n, fused=True)
        for pos, t in enumerate(toks[:20]):
            again.forward(t, pos)
        for s in fused.shards:
            s.reset()
        for pos, t in enumerate(toks[:20]):
            lg = fused.forward(t, pos)
        assert lg.tobytes() == again.logits.tobytes()
        assert not fused.shards[0].last_error()         # (the fused plan was never retired)
        fused.close(); plain.close(); again.close()


def test_group_exchange_timeout_falls_back_to_the_general_plan(hip, orc, tmp_path, monkeypatch):
    # every cluster exchange of the fused launches gives up at once (NL_FUSED_SPIN_LIMIT=0): the group retires the plan
    # in every shard, redoes the step on the five-launch plan and returns the general plan's logits, bit for bit
    shape, _ = SHAPES["gqa4"]
    p = tmp_path / "m.gguf"
    synth.generate_gguf(str(p), shape, "q4_0", 73, mode="qrand")
    g = gguf.load_gguf(str(p))
    plain = hip.LocalTPGroup(g, 4)
    monkeypatch.setenv("NL_FUSED_SPIN_LIMIT", "0")
    monkeypatch.setenv("NL_QUIET", "1")
    fused = hip.LocalTPGroup(g, 4, fused=True)
    assert fused.shards[0].plan_info()["fused_mode"] == 3
    toks = synth.prompt_ids(6, shape.vocab, seed=31)
    for pos, t in enumerate(toks):
        a = fused.forward(t, pos).copy()
        b = plain.forward(t, pos)
        assert a.tobytes() == b.tobytes(), pos
    assert fused.shards[0].plan_info()["fused_mode"] == 0        # retired
    assert "timed out" in fused.shards[0].last_error()
    fused.close(); plain.close()


def test_big_attention_geometry_across_the_in_launch_passes(hip, orc, tmp_path):
    # The 7.9B tier's TRUE attention geometry (D 4096 / 64 heads / 16 kv heads: two projection tiles per workgroup, a wavefront
    # holding two column groups) with a context long enough for every in-launch attention pass: fused mode 4 (one GPU) serves
    # EVERY position of the context (2048 is the reference's own cap, go/model.go:145-148) with 256-position passes -- from the second
    # pass on shared by the head's runner and three helper blocks, one or two passes each --, fused mode 3 (tensor-parallel shards,
    # tp 2 / 4 / 8) positions < 512.  Round 4 only checked passes 2 and 3 at D 1024 / 16 heads.  Teacher-forced against the oracle
    # at the pass edges (go/model.go:557-587), then a chained greedy run to the end of the context.
    shape = synth.ModelShape("big_geo", 2, 4096, 64, 16, 1024, seq_len=2048, interm=2048)
    p = tmp_path / "m.gguf"
    synth.generate_gguf(str(p), shape, "q4_0", 97, mode="qrand")
    g = gguf.load_gguf(str(p))
    check = (255, 256, 300, 511, 512, 700, 767, 768, 900, 1023, 1024, 1279, 1280, 1400, 1535, 1536, 1791, 1792, 1900, 2046, 2047)
    prompt = synth.prompt_ids(1998, shape.vocab, seed=41)
    ref = orc.OracleModel(g)
    orc.set_threads(min(32, os.cpu_count() or 1))
    seq, wants = list(prompt), {}
    for pos in range(2048):                     # the prompt teacher-forced, then the oracle's own greedy ids
        lg = ref.forward(seq[pos], pos)
        if pos in check:
            wants[pos] = lg.copy()
        if pos + 1 >= len(seq):
            seq.append(int(orc.argmax(lg)))
    orc.set_threads(1)
    ref.close()
    greedy = seq[len(prompt):]                  # the ids the steps at positions 1997 .. 2047 produce

    def held(lg, pos, what):
        d = float(np.abs(lg - wants[pos]).max()) / max(1.0, float(wants[pos].std()))
        assert d <= LOGIT_TOL, (what, pos, d)
        return d

    # one GPU: mode 4 (projection + attention + WO in one launch), passes 1 .. 3 and the first position of the general plan
    dev = hip.load_llama_model(g)
    info = dev.plan_info()
    assert info["fused_mode"] == 4 and info["fused_max_pos"] == 2048, info
    worst = 0.0
    for pos in range(2048):
        dev.forward(seq[pos], pos)
        if pos in check:
            worst = max(worst, held(dev.state.logits, pos, "mode 4"))
    assert dev.last_error() == ""
    # chained greedy decode from the prompt to the end of the context: 16-step graphs of the fused plan, single steps at the end
    dev.reset()
    dev.prefill(prompt)
    first = int(np.argmax(dev.state.logits))
    got = [first] + dev.decode_greedy(first, len(prompt), len(greedy) - 1)
    assert got == greedy
    assert dev.last_error() == ""
    dev.close()
    print(f"\nbig geometry, mode 4: max|gpu-oracle| = {worst:.2e} over positions {check}; {len(greedy)} greedy ids to the end of the context equal")

    # tensor-parallel shards: mode 3 below 512, the four-launch rank plan from there on
    for n in (2, 4, 8):
        grp = hip.LocalTPGroup(g, n, fused=True)
        info = grp.shards[0].plan_info()
        assert info["fused_mode"] == 3 and info["fused_max_pos"] == 512, info
        worst = 0.0
        for pos in range(769):
            lg = grp.forward(seq[pos], pos)
            if pos in check:
                worst = max(worst, held(lg, pos, f"mode 3 tp {n}"))
        assert not grp.shards[0].last_error()
        grp.close()
        print(f"big geometry, mode 3 tp {n}: max|gpu-oracle| = {worst:.2e}")


def test_attention_helpers_agree_with_the_general_plan_after_a_prompt(hip, tmp_path, monkeypatch):
    # the long-context plan of the wide tier on one GPU (from the second 256-position pass on, the passes below the last run on
    # three helper blocks, one or two passes each: nl_tp.h) against the five-launch plan of the same handle type, on a cache a
    # 2040-token prompt wrote (the batched path): positions in every pass count 1 .. 8 and at the pass edges, and the same with the
    # helpers switched off (a lone runner: positions below 1024)
    shape = synth.ModelShape("big_geo_h", 2, 4096, 64, 16, 1024, seq_len=2048, interm=2048)
    p = tmp_path / "m.gguf"
    synth.generate_gguf(str(p), shape, "q4_0", 97, mode="qrand")
    g = gguf.load_gguf(str(p))
    toks = synth.prompt_ids(2048, shape.vocab, seed=41)
    positions = (100, 255, 256, 300, 511, 512, 700, 767, 768, 769, 900, 1023, 1024, 1279, 1280, 1535, 1536, 1791, 1792, 2000, 2047)
    res = {}
    for name, env in (("general", {"NL_FUSED_MAX_POS": "0"}), ("helpers", {}), ("alone", {"NL_ATTN_HELPERS": "0"})):
        for k in ("NL_FUSED_MAX_POS", "NL_ATTN_HELPERS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        dev = hip.load_llama_model(g)
        dev.prefill(toks[:2040])
        out = {}
        for pos in positions:
            dev.forward(toks[pos], pos)
            out[pos] = dev.state.logits.copy()
        assert dev.last_error() == "", (name, dev.last_error())
        res[name] = out
        dev.close()
    for pos in positions:
        ref = res["general"][pos]
        for name in ("helpers", "alone"):      # (beyond 1024 the lone runner's handle is on the five-launch plan itself)
            d = float(np.abs(res[name][pos] - ref).max()) / max(1.0, float(ref.std()))
            assert d <= 2e-5, (name, pos, d)


def test_fused_modes_keep_out_of_a_launch_whose_staying_blocks_exceed_the_compute_units(hip, orc, tmp_path):
    # Modes 3 / 4 keep every LIVE block (projection tiles of an existing kv group) and every WO-owning block resident until its
    # rows are done; tile-less blocks without WO rows leave at once.  D 2560 / 40 heads / 5 kv heads: 200 live blocks in a grid of
    # 320 (block b is live when b % 8 < 5) + 160 WO owners (b < 160), 100 of them both = 260 blocks that stay > 256 compute
    # units: the last live blocks could never start and the first step would spin into the timeout fallback.  The engine must
    # pick mode 2 (whose tile-less blocks all leave) up front, silently and correctly; 3 kv groups of 8 heads at D 1536 -- a grid
    # of 320 too, but only 150 blocks stay -- keeps mode 4.
    for name, shape, mode in (("stay260", synth.ModelShape("fg_stay", 2, 2560, 40, 5, 512, seq_len=160, interm=1024), 2),
                              ("stay150", synth.ModelShape("fg_grid", 2, 1536, 24, 3, 512, seq_len=160, interm=1024), 4)):
        p = tmp_path / f"{name}.gguf"
        synth.generate_gguf(str(p), shape, "q4_0", 99, mode="qrand")
        g = gguf.load_gguf(str(p))
        dev = hip.load_llama_model(g)
        assert dev.plan_info()["fused_mode"] == mode, (name, dev.plan_info())
        ref = orc.OracleModel(g)
        orc.set_threads(min(16, os.cpu_count() or 1))
        for pos, t in enumerate(synth.prompt_ids(24, shape.vocab, seed=43)):
            dev.forward(t, pos)
            want = ref.forward(t, pos)
            assert np.abs(dev.state.logits - want).max() <= LOGIT_TOL * max(1.0, float(want.std())), (name, pos)
        orc.set_threads(1)
        assert dev.last_error() == "" and dev.plan_info()["fused_mode"] == mode      # no stall, no retirement
        dev.close(); ref.close()


def test_matrix_pipe_dot_products_match_oracle_and_vector_pipe(hip, orc, tmp_path, monkeypatch):
    # nl_tp.h mf_*: the feed-forward launch of the wide tiers multiplies Q4_0 weights on the matrix pipe (v_mfma_i32_4x4x4_16b_i8
    # over base-256 digits of the inputs, a permuted second copy of the matrices); NL_MFMA_DOT=0 keeps the vector pipe, 2 puts the
    # attention launch on the matrix pipe as well.  go/quant.go:45-94 (Q4_0 rows), go/model.go:597-612.  Four gate / up rounds per
    # workgroup (I 4096 on a 64-block grid).  All three against the oracle; the matrix-pipe results are NOT the vector pipe's bits
    # (which proves the path ran) but agree with them far inside the tolerance.
    shape = synth.ModelShape("mf_dot", 2, 1024, 16, 4, 2048, seq_len=96, interm=4096)
    p = tmp_path / "m.gguf"
    synth.generate_gguf(str(p), shape, "q4_0", 211, mode="qrand")
    g = gguf.load_gguf(str(p))
    tokens = synth.prompt_ids(40, shape.vocab, seed=43)
    ref = orc.OracleModel(g)
    orc.set_threads(min(16, os.cpu_count() or 1))
    want = [ref.forward(t, pos).copy() for pos, t in enumerate(tokens)]
    orc.set_threads(1)
    ref.close()
    got = {}
    monkeypatch.setenv("NL_WIDE_FFN", "2")      # (the one-launch feed-forward half on a 64-block grid too: by default it waits for grids that fill the chip)
    for knob in ("0", "1", "2"):
        monkeypatch.setenv("NL_MFMA_DOT", knob)
        dev = hip.load_llama_model(g)
        assert dev.plan_info()["fused_mode"] == 4, dev.plan_info()
        out, worst = [], 0.0
        for pos, t in enumerate(tokens):
            dev.forward(t, pos)
            out.append(dev.state.logits.copy())
            worst = max(worst, float(np.abs(out[-1] - want[pos]).max()) / max(1.0, float(want[pos].std())))
        assert dev.last_error() == ""
        dev.close()
        print(f"\nNL_MFMA_DOT={knob}: max|gpu-oracle| = {worst:.2e} over 40 positions")
        assert worst <= LOGIT_TOL, knob
        got[knob] = out
    monkeypatch.delenv("NL_MFMA_DOT")
    for knob in ("1", "2"):
        assert any(a.tobytes() != b.tobytes() for a, b in zip(got["0"], got[knob])), f"NL_MFMA_DOT={knob} ran the vector pipe"
        d = max(float(np.abs(a - b).max()) / max(1.0, float(a.std())) for a, b in zip(got["0"], got[knob]))
        assert d <= 2e-5, (knob, d)
    assert any(a.tobytes() != b.tobytes() for a, b in zip(got["1"], got["2"]))


def test_sixteen_byte_granules_are_never_seen_torn(hip):
    # the in-launch exchanges of nl_tp.h publish {tag, v0, v1, v2} with ONE dwordx4 store and read it with ONE dwordx4 load; that
    # the pair is single-copy atomic is an assumed hardware property (the tag check alone cannot see a torn copy).  128 writer
    # blocks x 256 lanes x 20000 generations against readers on other XCDs: every copy read must be of one generation.
    import ctypes as C
    from nanollama_amd import _lib
    torn, seen = C.c_ulonglong(0), C.c_ulonglong(0)
    rc = _lib.lib().nl_op_gran16_soak(0, 128, 20000, C.byref(torn), C.byref(seen))
    assert rc == 0
    print(f"\n16-byte granule soak: {seen.value:.3e} cross-XCD reads, {torn.value} torn")
    assert seen.value > 128 * 256 * 100 and torn.value == 0
