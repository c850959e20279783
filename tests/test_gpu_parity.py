"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the
committed reference-Python goldens.  Run with `pytest -m gpu` on the MI355X box.

Tolerances (stated once, used everywhere below):
  * single GEMV vs oracle: |d| <= 2e-5 * (1 + |out|)   -- only the summation order differs
    (per-lane block dots + tree reduction vs the Go loop's left-to-right float32 sum)
  * whole-forward logits vs oracle: <= LOGIT_TOL * max(1, std(logits)); measured values are printed
  * greedy token ids: identical, on inputs whose top-1/top-2 margin is >> LOGIT_TOL
"""
import os
import sys
from dataclasses import replace

import numpy as np
import pytest

from nanollama_amd import gguf, synth

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MODELS = ["tiny_f16", "tiny_q8_0", "tiny_q4_0", "tiny_qknorm_q8_0", "tiny_conj_q4_0", "tiny_tied_q8_0", "tiny_mha_q4_0"]
LOGIT_TOL = 1e-4
FP16X1_TOL = 1e-2    # the fp16x1 prompt precision mode's own tolerance (activations rounded to 11 bits): measured 5.8e-3 on mini at 1920 tokens


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


def _rand_matrix(rng, rows, cols, wtype):
    from nanollama_amd import quant
    w = (rng.random((rows, cols), dtype=np.float32) * 2 - 1) * np.float32(0.05)
    return quant.encode(w, synth.WTYPES[wtype])


@pytest.mark.parametrize("wtype", ["q8_0", "q4_0", "f16", "f32"])
@pytest.mark.parametrize("rows,cols", [(576, 576), (1536, 576), (576, 1536), (192, 768), (16, 32), (1, 64),
                                        (100, 96), (32000, 576), (4096, 4096), (1376, 4096), (512, 11008)])
def test_gemv_matches_oracle(hip, orc, wtype, rows, cols):
    if wtype == "f32" and rows * cols > 4_000_000:
        pytest.skip("f32 weights are plumbing only")
    rng = np.random.Generator(np.random.PCG64(rows * 131 + cols))
    raw = _rand_matrix(rng, rows, cols, wtype)
    x = rng.standard_normal(cols, dtype=np.float32)
    t = synth.WTYPES[wtype]
    want = orc.matmul(raw, t, x, rows, cols)
    got = hip.op_matmul(raw, t, x, rows, cols)
    err = np.abs(got - want) / (1 + np.abs(want))
    assert err.max() <= 2e-5, (err.max(), int(err.argmax()))


def test_gemv_edge_values(hip, orc):
    # extreme quants, zero / subnormal / negative scales, exact-cancel inputs
    rows, cols = 48, 128
    rng = np.random.Generator(np.random.PCG64(5))
    for t, bsz in ((gguf.GGML_Q8_0, 34), (gguf.GGML_Q4_0, 18)):
        raw = rng.integers(0, 256, size=rows * cols // 32 * bsz, dtype=np.uint8).reshape(-1, bsz)
        scales = np.array([0x0000, 0x8000, 0x0001, 0x03FF, 0x3C00, 0xBC00, 0x2E66, 0x5640], dtype=np.uint16)
        sel = scales[np.arange(raw.shape[0]) % len(scales)]
        raw[:, 0] = (sel & 0xFF).astype(np.uint8)
        raw[:, 1] = (sel >> 8).astype(np.uint8)
        raw = raw.reshape(-1)
        x = rng.standard_normal(cols, dtype=np.float32)
        want = orc.matmul(raw, t, x, rows, cols)
        got = hip.op_matmul(raw, t, x, rows, cols)
        assert np.all(np.abs(got - want) <= 2e-5 * (1 + np.abs(want)))
        zero = hip.op_matmul(raw, t, np.zeros(cols, np.float32), rows, cols)
        assert np.all(zero == 0)


def test_gemv_inf_and_nan_scales_follow_the_reference(hip, orc):
    # go/gguf.go:603-636: the fp16 -> f32 table keeps inf and NaN, and MatMulQ8_0 / MatMulQ4_0 (go/quant.go:74-94,
    # :149-165) multiply the block dot by that scale: a row with a +inf block is +inf, with +inf and -inf blocks NaN,
    # with a NaN scale NaN, and 0 * inf (an all-zero x under an inf scale) NaN.  The special blocks hold only
    # positive quants and x is positive, so the sign of each block dot does not depend on the summation order.
    rows, cols = 32, 256
    rng = np.random.Generator(np.random.PCG64(17))
    x = (np.abs(rng.standard_normal(cols)) + 0.25).astype(np.float32)
    special = {0: [(2, 0x7C00)], 1: [(2, 0xFC00)], 2: [(1, 0x7C00), (5, 0xFC00)], 3: [(7, 0x7E00)], 4: [(0, 0xFE01)],
               5: [(3, 0x7C00), (4, 0x7C00)], 17: [(6, 0x7C00)], 31: [(0, 0xFC00), (7, 0x7E00)]}
    for t, bsz in ((gguf.GGML_Q8_0, 34), (gguf.GGML_Q4_0, 18)):
        raw = rng.integers(0, 256, size=(rows, cols // 32, bsz), dtype=np.uint8)
        raw[:, :, 0:2] = np.frombuffer(np.float16(rng.uniform(0.01, 0.2, size=(rows, cols // 32))).tobytes(), np.uint8).reshape(rows, cols // 32, 2)
        for r, blocks in special.items():
            for b, bits in blocks:
                raw[r, b, 0], raw[r, b, 1] = bits & 0xFF, bits >> 8
                raw[r, b, 2:] = rng.integers(1, 128, size=bsz - 2) if t == gguf.GGML_Q8_0 else (rng.integers(9, 16, size=bsz - 2) * 17)
        flat = raw.reshape(-1)
        for xv in (x, np.where(np.arange(cols) // 32 == 2, 0, x).astype(np.float32)):   # second: block 2 of x all zero -> 0 * inf
            want = orc.matmul(flat, t, xv, rows, cols)
            assert np.isinf(want[0]) or np.isnan(want[0])
            for got in (hip.op_matmul(flat, t, xv, rows, cols),
                        hip.op_matmul_batch(flat, t, np.stack([xv, xv * 2, xv]), rows, cols)[0]):
                assert np.array_equal(np.isnan(got), np.isnan(want)), (t, np.flatnonzero(np.isnan(got) != np.isnan(want)))
                inf = np.isinf(want)
                assert np.array_equal(got[inf], want[inf])                      # same infinities, same signs
                fin = np.isfinite(want)
                assert np.all(np.abs(got[fin] - want[fin]) <= 2e-5 * (1 + np.abs(want[fin])))


@pytest.mark.parametrize("kind", ["q4_k", "q6_k"])
def test_gemv_reproduces_the_hand_built_kquant_super_blocks(hip, kind):
    # first-principles KAT (tests/kquant_kat.py; go/quant.go:171-396): exact binary fractions everywhere, so the device's
    # dot products with one-hot and small-integer x are exact in any summation order
    from test_oracle_kquants import _kat_matrix
    t = gguf.GGML_Q4_K if kind == "q4_k" else gguf.GGML_Q6_K
    raw, want = _kat_matrix(kind)
    probes = [((np.arange(512) * 7) % 5 - 2).astype(np.float32), np.ones(512, np.float32)]
    for k in (0, 1, 31, 32, 63, 64, 100, 127, 128, 191, 255, 256, 300, 511):
        e = np.zeros(512, np.float32)
        e[k] = 1.0
        probes.append(e)
    for x in probes:
        exact = (want.astype(np.float64) @ x.astype(np.float64)).astype(np.float32)
        assert np.array_equal(hip.op_matmul(raw, t, x, 2, 512), exact)


def test_gemv_rejects_unsupported_type(hip):
    # Q4_1 (type 3) is parsed by go/gguf.go but has no MatMul in go/quant.go either
    from nanollama_amd._lib import NlError
    with pytest.raises(NlError):
        hip.op_matmul(np.zeros(20 * 8, np.uint8), gguf.GGML_Q4_1, np.zeros(256, np.float32), 1, 256)


def test_rmsnorm_matches_oracle(hip, orc):
    rng = np.random.Generator(np.random.PCG64(3))
    for n in (64, 576, 4096):
        x = rng.standard_normal(n, dtype=np.float32) * 3
        w = rng.uniform(0.5, 1.5, n).astype(np.float32)
        got = hip.op_rmsnorm(x, w, 1e-5)
        want = orc.rmsnorm_into(x, w, 1e-5)
        assert np.abs(got - want).max() <= 1e-6 * (1 + np.abs(want).max())


def test_fast_exp_matches_float64_exp(hip):
    # Softmax / SiLU use float32(math.Exp(float64(x))) (go/quant.go:619, :629-631); the kernels compute it with a
    # short-chain float64 exponential (exp_f64_as_f32): bit-identical float32 results over the ranges the forward
    # pass feeds it (scores minus their maximum, minus the gate), denormal / overflow / NaN / infinity edges included
    rng = np.random.Generator(np.random.PCG64(41))
    x = np.concatenate([
        (-np.abs(rng.standard_normal(1_500_000)) * rng.choice([0.1, 1.0, 8.0, 40.0], 1_500_000)).astype(np.float32),
        (rng.standard_normal(1_500_000) * rng.choice([0.5, 4.0, 30.0], 1_500_000)).astype(np.float32),
        rng.uniform(-110.0, 90.0, 1_000_000).astype(np.float32),
        np.array([0.0, -0.0, 1.0, -1.0, -87.3, -88.0, -103.9, -104.0, -745.0, -746.0, -1e4, 88.7, 88.8, 709.0, 710.0, 1e4,
                  np.inf, -np.inf, np.nan, 1e-30, -1e-30, 0.6931472, -0.6931472, 0.34657359, -0.34657359], np.float32)])
    got = hip.op_exp(x)
    with np.errstate(over="ignore", under="ignore"):
        want = np.exp(x.astype(np.float64)).astype(np.float32)
    same = (got == want) | (np.isnan(got) & np.isnan(want))
    assert same.all(), (int((~same).sum()), x[~same][:5], got[~same][:5], want[~same][:5])


def _run_teacher_forced(hip, orc, path, tokens):
    g = gguf.load_gguf(path)
    dev = hip.load_llama_model(g)
    ref = orc.OracleModel(g)
    worst, scale = 0.0, 1.0
    for pos, tok in enumerate(tokens):
        dev.forward(int(tok), pos)
        want = ref.forward(int(tok), pos)
        scale = max(scale, float(want.std()))
        worst = max(worst, float(np.abs(dev.state.logits - want).max()))
    dev.close()
    return worst, scale


@pytest.mark.parametrize("tag", MODELS)
def test_forward_logits_match_oracle_and_goldens(hip, orc, tag):
    path = os.path.join(GOLDEN, tag + ".gguf")
    v = np.load(os.path.join(GOLDEN, tag + ".npz"))
    g = gguf.load_gguf(path)
    dev = hip.load_llama_model(g)
    ref = orc.OracleModel(g)
    scale = max(1.0, float(v["logits_full"].std()))
    worst_o = worst_g = 0.0
    for pos, tok in enumerate(v["prompt"]):
        dev.forward(int(tok), pos)
        want = ref.forward(int(tok), pos)
        worst_o = max(worst_o, float(np.abs(dev.state.logits - want).max()))
        worst_g = max(worst_g, float(np.abs(dev.state.logits - v["logits_full"][pos]).max()))
        assert int(np.argmax(dev.state.logits)) == int(np.argmax(want))
    print(f"\n{tag}: max|gpu-oracle|={worst_o:.2e} max|gpu-golden|={worst_g:.2e} (logit std {scale:.2f})")
    assert worst_o <= LOGIT_TOL * scale
    assert worst_g <= LOGIT_TOL * scale
    dev.close()


@pytest.mark.parametrize("tag", MODELS)
def test_greedy_ids_bit_exact(hip, orc, tag):
    from nanollama_amd.engine import Engine, GenParams
    v = np.load(os.path.join(GOLDEN, tag + ".npz"))
    g = gguf.load_gguf(os.path.join(GOLDEN, tag + ".gguf"))
    dev = hip.load_llama_model(g)
    eng = Engine(dev, eos_id=g.meta.eos_id, rep_penalty=1.0, rep_window=128)
    prompt = [int(t) for t in v["prompt"]]
    n = len(v["greedy_ids"])
    ids = eng.generate_ids(prompt, GenParams(max_tokens=n, temperature=0.0))
    assert ids == [int(t) for t in v["greedy_ids"]]                      # reference-Python golden
    ref_ids, _ = orc.OracleModel(g).generate_greedy(prompt, n)           # Go-restatement oracle
    assert ids == ref_ids
    # the same ids through the per-step host loop (nl_forward + host argmax) and nl_forward_argmax
    dev.reset()
    pos = 0
    for t in prompt:
        dev.forward(t, pos)
        pos += 1
    step_ids = []
    nxt = int(np.argmax(dev.state.logits))
    for _ in range(8):
        step_ids.append(nxt)
        nxt = dev.forward_argmax(nxt, pos)
        pos += 1
    assert step_ids == ids[:8]
    dev.close()


def test_graph_and_eager_agree_bitwise(hip):
    from nanollama_amd import _lib
    g = gguf.load_gguf(os.path.join(GOLDEN, "tiny_q8_0.gguf"))
    a = hip.load_llama_model(g)
    b = hip.load_llama_model(g, flags=_lib.NL_FLAG_NO_GRAPH)
    for pos, tok in enumerate([1, 17, 400, 3, 99]):
        a.forward(tok, pos)
        b.forward(tok, pos)
        assert a.state.logits.tobytes() == b.state.logits.tobytes()
    a.close(); b.close()


def test_reset_and_replay_is_deterministic(hip):
    g = gguf.load_gguf(os.path.join(GOLDEN, "tiny_q4_0.gguf"))
    dev = hip.load_llama_model(g)
    seq = [1, 5, 9, 200, 31]
    outs = []
    for _ in range(2):
        dev.reset()
        for pos, t in enumerate(seq):
            dev.forward(t, pos)
        outs.append(dev.state.logits.copy())
    assert outs[0].tobytes() == outs[1].tobytes()
    dev.close()


def test_step_beyond_the_reset_mark_reads_zero_rows(hip, orc):
    # Reset zeroes both caches (go/model.go:623-631); the library drops a per-stream mark instead and clears the
    # rows a later step could see without having rewritten them.  Forward straight at pos 5 after a reset therefore
    # attends over zero rows 0-4, exactly as the reference does on its freshly zeroed cache.
    g = gguf.load_gguf(os.path.join(GOLDEN, "tiny_q8_0.gguf"))
    dev = hip.load_llama_model(g)
    for pos, t in enumerate([1, 17, 45, 301, 7, 9, 12, 400]):     # dirty rows 0-7
        dev.forward(t, pos)
    dev.reset()
    ref = orc.OracleModel(g)
    for tok, pos in ((33, 5), (34, 6), (8, 2), (35, 7)):         # pos 5 first; 2 rewrites a cleared row later
        dev.forward(tok, pos)
        want = ref.forward(tok, pos)
        assert np.abs(dev.state.logits - want).max() <= LOGIT_TOL * max(1.0, float(want.std()))
    # chained decode and prefill beyond the mark clear their gap too
    dev.reset(); ref.reset()
    ids = dev.decode_greedy(21, 3, 4)
    want_ids = []
    tok = 21
    for k in range(4):
        tok = int(orc.argmax(ref.forward(tok, 3 + k)))
        want_ids.append(tok)
    assert ids == want_ids
    dev.close(); ref.close()


def test_streams_are_independent(hip):
    g = gguf.load_gguf(os.path.join(GOLDEN, "tiny_q8_0.gguf"))
    dev = hip.load_llama_model(g, max_streams=2)
    a, b = [1, 7, 8, 9], [1, 300, 301, 302]
    for pos in range(4):               # interleave two sequences on two KV streams
        dev.forward(a[pos], pos, stream=0)
        la = dev.state.logits.copy()
        dev.forward(b[pos], pos, stream=1)
    solo = hip.load_llama_model(g)
    for pos in range(4):
        solo.forward(a[pos], pos)
    assert la.tobytes() == solo.state.logits.tobytes()
    dev.close(); solo.close()


def test_long_context_split_attention(hip, orc, tmp_path):
    # seq 640 -> five 128-position attention splits; checks the split merge and position 0..639 plumbing
    shape = replace(synth.TIERS["tiny"], name="tiny_long", seq_len=640)
    p = tmp_path / "long.gguf"
    synth.generate_gguf(str(p), shape, "q8_0", 21)
    g = gguf.load_gguf(str(p))
    dev = hip.load_llama_model(g)
    ref = orc.OracleModel(g)
    toks = synth.prompt_ids(300, shape.vocab, seed=3)
    worst = 0.0
    for pos, t in enumerate(toks):
        dev.forward(t, pos)
        want = ref.forward(t, pos)
        if pos % 37 == 0 or pos >= 296:
            worst = max(worst, float(np.abs(dev.state.logits - want).max()))
    print(f"\nlong context: max|gpu-oracle|={worst:.2e}")
    assert worst <= LOGIT_TOL * max(1.0, float(want.std()))
    dev.close()


def test_maximum_context_2048_positions(hip, orc, tmp_path):
    # the largest context the Go engine allows (SeqLen is capped to 2048, go/model.go:145-148): all 16 attention
    # splits in use at the last position; the decode loop stops when pos reaches SeqLen (go/main.go:216)
    shape = replace(synth.TIERS["tiny"], name="tiny_max", seq_len=4096)      # the file says 4096, the engine caps it
    p = tmp_path / "max.gguf"
    synth.generate_gguf(str(p), shape, "q4_0", 27)
    g = gguf.load_gguf(str(p))
    dev = hip.load_llama_model(g)
    assert dev.config.seq_len == 2048
    ref = orc.OracleModel(g)
    toks = synth.prompt_ids(2048, shape.vocab, seed=2)
    for pos, t in enumerate(toks):
        want = ref.forward(t, pos)
    dev.prefill(toks[:2040])                       # positions 0..2039 in one multi-token step
    for pos in range(2040, 2048):                  # the last eight through the single-token kernels
        dev.forward(toks[pos], pos)
    err = float(np.abs(dev.state.logits - want).max())
    print(f"\nposition 2047: max|gpu-oracle|={err:.2e}")
    assert err <= LOGIT_TOL * max(1.0, float(want.std()))
    from nanollama_amd._lib import NlError
    with pytest.raises(NlError):
        dev.forward(1, 2048)                       # past the cache
    assert dev.decode_greedy(toks[2046], 2046, 10) == dev.decode_greedy(toks[2046], 2046, 2)   # stops at SeqLen: 2 steps
    assert len(dev.decode_greedy(toks[2047], 2047, 5)) == 1
    dev.close()


def test_argument_and_state_errors(hip):
    from nanollama_amd._lib import NlError
    g = gguf.load_gguf(os.path.join(GOLDEN, "tiny_q8_0.gguf"))
    dev = hip.load_llama_model(g)
    with pytest.raises(NlError, match="token"):
        dev.forward(512, 0)
    with pytest.raises(NlError, match="pos"):
        dev.forward(1, 64)
    with pytest.raises(NlError, match="stream"):
        dev.forward(1, 0, stream=3)
    dev.close()


def test_missing_tensor_is_reported(hip, tmp_path):
    from nanollama_amd._lib import NlError
    src = gguf.load_gguf(os.path.join(GOLDEN, "tiny_q8_0.gguf"))
    src.tensor_order = [n for n in src.tensor_order if n != "blk.1.ffn_down.weight"]
    with pytest.raises(NlError, match="ffn_down"):
        hip.load_llama_model(src)


@pytest.mark.parametrize("tier,wtype", [("small_test", "q8_0"), ("small_test", "q4_0")])
def test_mid_size_model_matches_oracle(hip, orc, tmp_path, tier, wtype):
    # 3 layers, D=192 (6 blocks per row: odd pair counts, partial k-lane steps), V=1024, MHA, hd=64
    shape = synth.TIERS[tier]
    p = tmp_path / "m.gguf"
    synth.generate_gguf(str(p), shape, wtype, 77)
    worst, scale = _run_teacher_forced(hip, orc, str(p), synth.prompt_ids(40, shape.vocab, seed=11))
    print(f"\n{tier}/{wtype}: max|gpu-oracle|={worst:.2e} (logit std {scale:.2f})")
    assert worst <= LOGIT_TOL * scale


@pytest.mark.parametrize("tier,wtype,ntok", [("nano", "q8_0", 6), ("nano", "f16", 3), ("mini", "q4_0", 4),
                                              ("goldie", "q4_0", 3)])
def test_full_size_tiers_match_oracle(hip, orc, tmp_path, tier, wtype, ntok):
    # BASELINE.json's own shapes (89M / 173M / 841M), teacher-forced, against the CPU oracle
    shape = synth.TIERS[tier]
    p = tmp_path / "m.gguf"
    synth.generate_gguf(str(p), shape, wtype, mode="qrand" if tier == "goldie" else "float")
    orc.set_threads(min(32, os.cpu_count() or 1))
    worst, scale = _run_teacher_forced(hip, orc, str(p), synth.prompt_ids(ntok, shape.vocab, seed=5))
    orc.set_threads(1)
    print(f"\n{tier}/{wtype}: max|gpu-oracle|={worst:.2e} (logit std {scale:.2f})")
    assert worst <= LOGIT_TOL * scale


def test_big_full_shape_matches_oracle(hip, orc, tmp_path):
    # BASELINE.json configs[4] at its own shape (nanollama/llama.py:50: 40 layers, D 4096, 64 heads / 16 kv heads,
    # FFN 11008, vocabulary 96000 -- 7.9B parameters, Q4_0): two teacher-forced tokens against the CPU oracle, on one
    # GPU and as 2, 4 and 8 tensor-parallel shards (the slicing and the partial sums of the multi-GPU plans, stepped
    # in-process on the one GPU).
    shape = synth.TIERS["big"]
    p = os.path.join(os.environ.get("NL_BENCH_DIR", "/tmp"), "nl_bench_big_q4_0_qrand.gguf")   # bench.py's file when present
    if not os.path.exists(p):
        synth.generate_gguf(p + ".tmp", shape, "q4_0", mode="qrand")
        os.replace(p + ".tmp", p)
    g = gguf.load_gguf(p)
    assert (g.meta.num_layers, g.meta.embed_dim, g.meta.num_heads, g.meta.num_kv_heads, g.meta.interm_size,
            g.meta.vocab_size) == (40, 4096, 64, 16, 11008, 96000)
    toks = synth.prompt_ids(2, shape.vocab, seed=5)
    ref = orc.OracleModel(g)
    orc.set_threads(min(32, os.cpu_count() or 1))
    wants = [ref.forward(t, pos).copy() for pos, t in enumerate(toks)]
    orc.set_threads(1)
    ref.close()
    dev = hip.load_llama_model(g)
    for pos, t in enumerate(toks):
        dev.forward(t, pos)
        d, scale = float(np.abs(dev.state.logits - wants[pos]).max()), max(1.0, float(wants[pos].std()))
        print(f"\nbig/q4_0 pos {pos}: max|gpu-oracle|={d:.2e} (logit std {scale:.2f})")
        assert d <= LOGIT_TOL * scale
        assert int(np.argmax(dev.state.logits)) == int(orc.argmax(wants[pos]))
    dev.close()
    for n in (2, 4, 8):      # the shardings of the 1/2/4/8-GPU curve, full shape (8: 8 q heads + 2 kv heads, 1376 FFN rows per rank)
        grp = hip.LocalTPGroup(g, n)
        for pos, t in enumerate(toks):
            lg = grp.forward(t, pos)
            d = float(np.abs(lg - wants[pos]).max())
            print(f"big/q4_0 tp{n} pos {pos}: max|tp-oracle|={d:.2e}")
            assert d <= LOGIT_TOL * max(1.0, float(wants[pos].std()))
            assert int(np.argmax(lg)) == int(orc.argmax(wants[pos]))
        grp.close()
    for n in (2, 4, 8):      # ... and a rank's layer as TWO launches (nl_tp.h: the plan of a push group; at tp 2 its second launch is wide_ffn_kernel), full shape
        grp = hip.LocalTPGroup(g, n, fused=True)
        assert grp.shards[0].plan_info()["fused_mode"] == 3, grp.shards[0].plan_info()
        for pos, t in enumerate(toks):
            lg = grp.forward(t, pos)
            d = float(np.abs(lg - wants[pos]).max())
            print(f"big/q4_0 tp{n} two-launch layers pos {pos}: max|tp-oracle|={d:.2e}")
            assert d <= LOGIT_TOL * max(1.0, float(wants[pos].std()))
            assert int(np.argmax(lg)) == int(orc.argmax(wants[pos]))
        grp.close()


def test_nano_full_size_long_greedy_run_matches_oracle(hip, orc, tmp_path):
    # BASELINE.json configs[1] at its own shape and length and beyond: nano Q8_0, 8-token prompt, 320 greedy tokens
    # (positions cross two 128-position attention split boundaries).  The oracle generates; the device is fed the
    # oracle's tokens and must pick the same next id at every step (a step whose oracle top-2 margin is below 5e-5
    # would be allowed to differ -- summation order -- but none is expected on this model); then the chained on-device
    # loop must reproduce the whole sequence by itself.
    shape = synth.TIERS["nano"]
    p = tmp_path / "nano.gguf"
    synth.generate_gguf(str(p), shape, "q8_0", mode="float")
    g = gguf.load_gguf(str(p))
    ref = orc.OracleModel(g)
    orc.set_threads(min(32, os.cpu_count() or 1))
    prompt = synth.prompt_ids(8, shape.vocab, seed=7)
    n = 320
    ids, logits = ref.generate_greedy(prompt, n, want_logits=True)
    orc.set_threads(1)
    assert len(ids) == n
    top2 = np.partition(logits, -2, axis=1)[:, -2:]
    margin = top2[:, 1] - top2[:, 0]
    dev = hip.load_llama_model(g)
    dev.prefill(prompt)
    assert int(np.argmax(dev.state.logits)) == ids[0]
    worst = 0.0
    for k in range(n - 1):
        got = dev.forward_argmax(ids[k], len(prompt) + k)
        if got != ids[k + 1]:
            assert margin[k + 1] < 5e-5, (k, got, ids[k + 1], float(margin[k + 1]))
    dev.forward(ids[n - 2], len(prompt) + n - 2)
    worst = float(np.abs(dev.state.logits - logits[n - 1]).max())
    print(f"\nnano 320-token run: min top-2 margin {float(margin.min()):.2e}, final-step max|gpu-oracle| {worst:.2e}")
    assert worst <= LOGIT_TOL * max(1.0, float(logits[n - 1].std()))
    if float(margin.min()) > 5e-5:
        dev.reset()
        dev.prefill(prompt)
        assert [ids[0]] + dev.decode_greedy(ids[0], len(prompt), n - 1) == ids
    dev.close()


@pytest.mark.parametrize("tag,n", [("tiny_q8_0", 2), ("tiny_q4_0", 2), ("tiny_mha_q4_0", 2)])
def test_tensor_parallel_shards_match_single_gpu(hip, orc, tag, n):
    # the TP sharding arithmetic (row-split QKV/gate/up/LM head, column-split WO/down, all-reduce seams)
    # stepped in-process on one GPU; must agree with the unsharded engine and the oracle
    g = gguf.load_gguf(os.path.join(GOLDEN, tag + ".gguf"))
    grp = hip.LocalTPGroup(g, n)
    one = hip.load_llama_model(g)
    ref = orc.OracleModel(g)
    v = np.load(os.path.join(GOLDEN, tag + ".npz"))
    worst1 = worst_o = 0.0
    for pos, tok in enumerate(v["prompt"]):
        lg = grp.forward(int(tok), pos)
        one.forward(int(tok), pos)
        want = ref.forward(int(tok), pos)
        worst1 = max(worst1, float(np.abs(lg - one.state.logits).max()))
        worst_o = max(worst_o, float(np.abs(lg - want).max()))
        assert int(np.argmax(lg)) == int(np.argmax(want))
    print(f"\n{tag} tp{n}: max|tp-single|={worst1:.2e} max|tp-oracle|={worst_o:.2e}")
    assert worst_o <= LOGIT_TOL and worst1 <= LOGIT_TOL
    grp.close(); one.close()


@pytest.mark.parametrize("tag", ["tiny_q8_0", "tiny_q4_0", "tiny_tied_q8_0"])
def test_collective_plan_runs_through_rccl_with_one_rank(hip, tag, monkeypatch):
    # The tensor-parallel launch plan (all-reduce after WO and after down, all-gather of the logit slices, residual
    # added in the next kernel's prologue) on the ONE GPU of this box: a 1-rank RCCL communicator makes every
    # collective an identity, so logits and greedy ids must equal the ordinary plan's -- this is the code path the
    # 8-GPU run takes, minus the data exchange itself.  Captured in the hipGraph (eager fallback if capture fails).
    g = gguf.load_gguf(os.path.join(GOLDEN, tag + ".gguf"))
    v = np.load(os.path.join(GOLDEN, tag + ".npz"))
    toks = [int(t) for t in v["prompt"]]
    base = hip.load_llama_model(g)
    monkeypatch.setenv("NL_FORCE_TP_PLAN", "1")
    try:
        cid = hip.comm_unique_id()
    except Exception as exc:      # no RCCL library on the box
        pytest.skip(f"RCCL unavailable: {exc}")
    tp = hip.load_llama_model(g, comm_id=cid)
    monkeypatch.delenv("NL_FORCE_TP_PLAN")
    for pos, t in enumerate(toks):
        base.forward(t, pos)
        tp.forward(t, pos)
        assert np.abs(tp.state.logits - base.state.logits).max() <= 2e-6, pos
    assert np.abs(tp.state.logits - v["logits_full"][len(toks) - 1]).max() <= LOGIT_TOL
    nxt = int(np.argmax(base.state.logits))
    assert tp.decode_greedy(nxt, len(toks), 8) == base.decode_greedy(nxt, len(toks), 8)
    base.close(); tp.close()


def test_tensor_parallel_big_shapes(hip, orc, tmp_path):
    # 2 layers with big-like ratios (GQA 2:1 over 16 heads, FFN multiple of 32*8) on 4 and 8 shards,
    # context long enough for two attention splits
    shape = synth.ModelShape("tp_probe", 2, 1024, 16, 8, 2048, seq_len=160, interm=2816)
    p = tmp_path / "tp.gguf"
    synth.generate_gguf(str(p), shape, "q4_0", 9)
    g = gguf.load_gguf(str(p))
    ref = orc.OracleModel(g)
    toks = synth.prompt_ids(140, shape.vocab, seed=2)
    wants = []
    for pos, t in enumerate(toks):
        wants.append(ref.forward(t, pos).copy())
    for n in (4, 8):
        grp = hip.LocalTPGroup(g, n)
        worst = 0.0
        for pos, t in enumerate(toks):
            lg = grp.forward(t, pos)
            if pos % 20 == 0 or pos > 130:
                worst = max(worst, float(np.abs(lg - wants[pos]).max()))
        print(f"\ntp{n}: max|tp-oracle|={worst:.2e}")
        assert worst <= LOGIT_TOL * max(1.0, float(wants[-1].std()))
        grp.close()


def test_prefill_matches_token_at_a_time(hip, orc):
    # nl_prefill runs 64-token tiles through the MFMA path; token-at-a-time Forward is the VALU decode path.
    # Same arithmetic up to summation order / the fp16 hi+lo activation split: logits within LOGIT_TOL,
    # greedy continuation identical.
    g = gguf.load_gguf(os.path.join(GOLDEN, "tiny_q4_0.gguf"))
    v = np.load(os.path.join(GOLDEN, "tiny_q4_0.npz"))
    toks = [int(t) for t in v["prompt"]]
    a = hip.load_llama_model(g)
    b = hip.load_llama_model(g)
    a.prefill(toks)
    for pos, t in enumerate(toks):
        b.forward(t, pos)
    assert np.abs(a.state.logits - b.state.logits).max() <= LOGIT_TOL
    assert np.abs(a.state.logits - v["logits_full"][len(toks) - 1]).max() <= LOGIT_TOL
    # the KV cache written by prefill serves the decode kernels: same greedy continuation as the golden
    nxt = int(np.argmax(a.state.logits))
    assert [nxt] + a.decode_greedy(nxt, len(toks), 8) == [int(t) for t in v["greedy_ids"][:9]]
    c = hip.load_llama_model(g)
    c.prefill(toks[:5], want_logits=False)
    c.prefill(toks[5:], pos0=5)
    assert np.abs(c.state.logits - b.state.logits).max() <= LOGIT_TOL
    from nanollama_amd._lib import NlError
    with pytest.raises(NlError, match="exceeds seq_len"):
        c.prefill([1] * 10, pos0=60)
    c.prefill([])  # empty prompt is a no-op
    a.close(); b.close(); c.close()


_FUSED_SNIPPET = r"""
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from nanollama_amd import gguf, model as hip
G = sys.argv[2]
for tag in ("tiny_q8_0", "tiny_q4_0", "tiny_conj_q4_0", "tiny_tied_q8_0", "tiny_mha_q4_0", "tiny_qknorm_q8_0"):
    g = gguf.load_gguf(os.path.join(G, tag + ".gguf"))
    v = np.load(os.path.join(G, tag + ".npz"))
    toks = [int(t) for t in v["prompt"]]
    dev = hip.load_llama_model(g)
    dev.prefill(toks)
    scale = max(1.0, float(v["logits_full"].std()))
    err = float(np.abs(dev.state.logits - v["logits_full"][len(toks) - 1]).max())
    assert err <= 1e-4 * scale, (tag, err)
    nxt = int(np.argmax(dev.state.logits))
    assert [nxt] + dev.decode_greedy(nxt, len(toks), 6) == [int(t) for t in v["greedy_ids"][:7]], tag
    dev.close()
# Qwen-style attention biases through the fused Q|K|V epilogue, against the oracle
import tempfile
sys.path.insert(0, os.path.join(sys.argv[1], "oracle"))
import oracle as orc
from nanollama_amd import synth
for wtype in ("q8_0", "q4_0"):
    shape = synth.ModelShape("bias_probe", 2, 128, 4, 2, 512, seq_len=64, interm=512, attn_bias=True)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "b.gguf")
        synth.generate_gguf(path, shape, wtype, 41)
        g = gguf.load_gguf(path)
        ref = orc.OracleModel(g)
        toks = synth.prompt_ids(14, shape.vocab, seed=9)
        for pos, t in enumerate(toks):
            want = ref.forward(t, pos).copy()
        dev = hip.load_llama_model(g)
        dev.prefill(toks)
        err = float(np.abs(dev.state.logits - want).max())
        assert err <= 1e-4, ("bias", wtype, err)
        dev.close()
print("fused ok")
"""


def test_fused_gemm_epilogues_on_golden_models(hip):
    # the gate || up GEMM with the SwiGLU epilogue and the Q|K|V GEMM with the RoPE / KV-store epilogue normally need
    # >= 128 workgroups (1000+ token prompts, covered by the mini 1920-token test); the two knobs force them onto
    # the tiny golden models (standard and conjugate RoPE, tied head, MHA; the QK-norm model keeps the unfused
    # RoPE path by design).  The knobs are read once per process, hence the child interpreter.
    import subprocess, sys
    env = dict(os.environ, NL_FUSED_SWIGLU_MIN_WG="1", NL_FUSED_ROPE_MIN_WG="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _FUSED_SNIPPET, root, GOLDEN], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "fused ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("tag", ["tiny_q8_0", "tiny_mha_q4_0", "tiny_qknorm_q8_0", "tiny_f16"])
@pytest.mark.parametrize("n", [3, 5, 7])
def test_short_prefill_matches_golden(hip, tag, n):
    # 3-7 tokens: the multi-token step with the per-token attention kernel, which (all positions < 128) normalises
    # its single split and writes the WO fragments itself -- GQA 2 and 1, with and without QK-norm
    g = gguf.load_gguf(os.path.join(GOLDEN, tag + ".gguf"))
    v = np.load(os.path.join(GOLDEN, tag + ".npz"))
    toks = [int(t) for t in v["prompt"]][:n]
    dev = hip.load_llama_model(g)
    dev.prefill(toks)
    scale = max(1.0, float(v["logits_full"].std()))
    assert np.abs(dev.state.logits - v["logits_full"][n - 1]).max() <= LOGIT_TOL * scale
    dev.close()


@pytest.mark.parametrize("tag", ["tiny_q8_0", "tiny_qknorm_q8_0", "tiny_conj_q4_0", "tiny_tied_q8_0", "tiny_mha_q4_0"])
def test_prefill_variants_match_golden(hip, tag):
    g = gguf.load_gguf(os.path.join(GOLDEN, tag + ".gguf"))
    v = np.load(os.path.join(GOLDEN, tag + ".npz"))
    toks = [int(t) for t in v["prompt"]]
    dev = hip.load_llama_model(g)
    dev.prefill(toks)
    scale = max(1.0, float(v["logits_full"].std()))
    assert np.abs(dev.state.logits - v["logits_full"][len(toks) - 1]).max() <= LOGIT_TOL * scale
    nxt = int(np.argmax(dev.state.logits))
    assert [nxt] + dev.decode_greedy(nxt, len(toks), 6) == [int(t) for t in v["greedy_ids"][:7]]
    dev.close()


def test_long_prefill_crosses_tiles_and_attention_splits(hip, orc, tmp_path):
    # 300-token prompt: five 64-token MFMA tiles, three 128-position attention splits
    shape = replace(synth.TIERS["tiny"], name="tiny_long", seq_len=640)
    p = tmp_path / "long.gguf"
    synth.generate_gguf(str(p), shape, "q4_0", 23)
    g = gguf.load_gguf(str(p))
    dev = hip.load_llama_model(g)
    ref = orc.OracleModel(g)
    toks = synth.prompt_ids(300, shape.vocab, seed=4)
    for pos, t in enumerate(toks):
        want = ref.forward(t, pos)
    dev.prefill(toks)
    err = float(np.abs(dev.state.logits - want).max())
    print(f"\nlong prefill: max|gpu-oracle|={err:.2e}")
    assert err <= LOGIT_TOL * max(1.0, float(want.std()))
    # and decode continues from that cache exactly like the oracle
    nxt = int(np.argmax(want))
    ids = dev.decode_greedy(nxt, len(toks), 5)
    ref_ids = []
    cur = nxt
    for k in range(5):
        lg = ref.forward(cur, len(toks) + k)
        cur = int(np.argmax(lg))
        ref_ids.append(cur)
    assert ids == ref_ids
    dev.close()


@pytest.mark.parametrize("heads,kv,hd", [(4, 4, 64), (4, 2, 64), (6, 2, 32), (8, 2, 64), (8, 1, 32), (3, 1, 64)])
def test_prefill_attention_tiles_every_gqa_ratio(hip, orc, tmp_path, heads, kv, hd):
    # the MFMA prefill attention kernel is instantiated per (head_dim, query heads per kv head); a 333-token
    # prompt in two calls (pos0 > 0, ragged last tile, three 128-position splits) against the oracle.
    shape = replace(synth.TIERS["tiny"], name=f"t{heads}_{kv}_{hd}", dim=heads * hd, n_head=heads, n_kv_head=kv,
                    seq_len=400, interm=256, n_layer=2)
    p = tmp_path / "g.gguf"
    synth.generate_gguf(str(p), shape, "q8_0", 31)
    g = gguf.load_gguf(str(p))
    dev = hip.load_llama_model(g)
    ref = orc.OracleModel(g)
    toks = synth.prompt_ids(333, shape.vocab, seed=9)
    for pos, t in enumerate(toks):
        want = ref.forward(t, pos)
    dev.prefill(toks[:77], want_logits=False)
    dev.prefill(toks[77:], pos0=77)
    err = float(np.abs(dev.state.logits - want).max())
    print(f"\nheads={heads} kv={kv} hd={hd}: max|gpu-oracle|={err:.2e}")
    assert err <= LOGIT_TOL * max(1.0, float(want.std()))
    nxt = int(np.argmax(want))
    assert dev.decode_greedy(nxt, len(toks), 3)[0] == int(np.argmax(ref.forward(nxt, len(toks))))
    dev.close()


@pytest.mark.parametrize("kv16_min", [16, 1 << 30])
@pytest.mark.parametrize("heads,kv,hd", [(4, 1, 64), (6, 2, 32), (8, 1, 32), (2, 2, 64)])
def test_prefill_attention_chunk_runs_with_and_without_prebuilt_images(hip, orc, tmp_path, monkeypatch, heads, kv, hd, kv16_min):
    # Prompts: a workgroup of attn_tile16_kernel folds a RUN of 128-key chunks with an online softmax.  From 256 tokens on
    # the chunks arrive as fp16 hi/lo LDS images built once per layer (kv16_build_kernel) and fetched by LDS-DMA into a
    # double buffer, 256 rows per workgroup; below, every workgroup converts its own chunks (128 rows).  Both variants,
    # forced by NL_KV16_MIN_TOKENS, on a 700-token prompt in three calls (ragged tiles, pos0 > 0, runs of several chunks,
    # rows that see no key of a run's first chunk) against the oracle.
    monkeypatch.setenv("NL_KV16_MIN_TOKENS", str(kv16_min))
    shape = replace(synth.TIERS["tiny"], name=f"r{heads}_{kv}_{hd}", dim=heads * hd, n_head=heads, n_kv_head=kv,
                    seq_len=768, interm=128, n_layer=2)
    p = tmp_path / "g.gguf"
    synth.generate_gguf(str(p), shape, "q8_0", 57)
    g = gguf.load_gguf(str(p))
    dev = hip.load_llama_model(g)
    ref = orc.OracleModel(g)
    toks = synth.prompt_ids(700, shape.vocab, seed=13)
    for pos, t in enumerate(toks):
        want = ref.forward(t, pos)
    dev.prefill(toks[:45], want_logits=False)
    dev.prefill(toks[45:391], pos0=45, want_logits=False)
    dev.prefill(toks[391:], pos0=391)
    err = float(np.abs(dev.state.logits - want).max())
    print(f"\nheads={heads} kv={kv} hd={hd} images from {kv16_min} tokens: max|gpu-oracle|={err:.2e}")
    assert err <= LOGIT_TOL * max(1.0, float(want.std()))
    nxt = int(np.argmax(want))
    assert dev.decode_greedy(nxt, len(toks), 3)[0] == int(np.argmax(ref.forward(nxt, len(toks))))
    dev.close()


def test_prefill_attention_more_workgroups_than_the_chip_holds(hip, orc, tmp_path, monkeypatch):
    # 32 kv heads x 3 tiles x one-chunk runs (NL_ATT_RUN=1) = 384 listed workgroups at one per CU: a second round of
    # dispatches, partial slots up to 5, G = 1 (256 tokens per tile).
    monkeypatch.setenv("NL_ATT_RUN", "1")
    shape = replace(synth.TIERS["tiny"], name="kv32", dim=1024, n_head=32, n_kv_head=32, seq_len=768, interm=64, n_layer=2)
    p = tmp_path / "g.gguf"
    synth.generate_gguf(str(p), shape, "q8_0", 91)
    g = gguf.load_gguf(str(p))
    dev = hip.load_llama_model(g)
    ref = orc.OracleModel(g)
    toks = synth.prompt_ids(700, shape.vocab, seed=21)
    for pos, t in enumerate(toks):
        want = ref.forward(t, pos)
    dev.prefill(toks)
    err = float(np.abs(dev.state.logits - want).max())
    print(f"\n32 kv heads, 384 attention workgroups: max|gpu-oracle|={err:.2e}")
    assert err <= LOGIT_TOL * max(1.0, float(want.std()))
    dev.close()


@pytest.mark.parametrize("dim,heads,kv,hd,interm,n", [(160, 5, 5, 32, 224, 200), (192, 3, 1, 64, 352, 131), (256, 4, 1, 64, 96, 300)])
def test_q4_prompt_gemm_through_lds_matches_the_per_wavefront_kernel(hip, orc, tmp_path, monkeypatch, dim, heads, kv, hd, interm, n):
    # Q4_0 prompts of >= 128 tokens take qgemm2_kernel (weights expanded once per workgroup into LDS, nl_qgemm2.h) with
    # its own plain / SwiGLU / RoPE epilogues.  Shapes whose K is not a multiple of the 128-column chunk, whose row count
    # is not a multiple of the 64-row workgroup and whose token count leaves a ragged last tile: bit-identical logits,
    # KV rows and ids to qgemm_kernel (same arithmetic per output, NL_QG2_MIN_TOKENS disables the new kernel), and
    # within tolerance of the oracle.
    shape = replace(synth.TIERS["tiny"], name=f"q4_{dim}_{interm}", dim=dim, n_head=heads, n_kv_head=kv, seq_len=320,
                    interm=interm, n_layer=2)
    p = tmp_path / "g.gguf"
    synth.generate_gguf(str(p), shape, "q4_0", 17)
    g = gguf.load_gguf(str(p))
    toks = synth.prompt_ids(n, shape.vocab, seed=4)
    ref = orc.OracleModel(g)
    for pos, t in enumerate(toks):
        want = ref.forward(t, pos)
    outs = []
    monkeypatch.setenv("NL_FOLD_NORM", "0")     # (the folded RMSNorm scales the consumer's output instead of its input: own test below)
    for knob in (None, "1000000"):
        if knob is None:
            monkeypatch.delenv("NL_QG2_MIN_TOKENS", raising=False)
        else:
            monkeypatch.setenv("NL_QG2_MIN_TOKENS", knob)
        dev = hip.load_llama_model(g)
        dev.prefill(toks)
        logits = dev.state.logits.copy()
        ids = dev.decode_greedy(int(np.argmax(logits)), n, 4)
        outs.append((logits, ids))
        dev.close()
    assert np.array_equal(outs[0][0], outs[1][0])
    assert outs[0][1] == outs[1][1]
    err = float(np.abs(outs[0][0] - want).max())
    assert err <= LOGIT_TOL * max(1.0, float(want.std()))


@pytest.mark.parametrize("wt_variant", ["q4_0", "q4_0_bias_conj"])
def test_prompt_with_rmsnorm_folded_into_the_gemms_matches_oracle(hip, orc, tmp_path, monkeypatch, wt_variant):
    # Long Q4_0 prompts run without bnorm launches: the WO / down GEMM epilogues emit the next GEMM's fragments (x * g) and
    # per-64-row sums of squares, the consuming GEMM scales its output by inv (go/quant.go:597-607 restated as the decode
    # GEMV restates it).  1100 tokens of a 3-layer D = 512 GQA model (every GEMM grid >= 128 workgroups, the folding
    # condition): logits and the next greedy ids against the oracle, and against the same prompt with NL_FOLD_NORM=0
    # (close, but not bit-identical -- which also shows that the folded path is the one that ran).
    shape = synth.ModelShape("fold_probe", 3, 512, 8, 4, 1024, seq_len=1200, interm=1536,
                             rope_conjugate=wt_variant != "q4_0", attn_bias=wt_variant != "q4_0")
    p = tmp_path / "f.gguf"
    synth.generate_gguf(str(p), shape, "q4_0", 29)
    g = gguf.load_gguf(str(p))
    n = 1100
    toks = synth.prompt_ids(n, shape.vocab, seed=6)
    ref = orc.OracleModel(g)
    orc.set_threads(min(16, os.cpu_count() or 1))
    for pos, t in enumerate(toks):
        want = ref.forward(t, pos).copy()
    ref_ids = []
    tok = int(orc.argmax(want))
    for k in range(4):
        ref_ids.append(tok)
        tok = int(orc.argmax(ref.forward(tok, n + k)))
    orc.set_threads(1)
    outs = {}
    for knob in ("1", "0"):
        monkeypatch.setenv("NL_FOLD_NORM", knob)
        dev = hip.load_llama_model(g)
        dev.prefill(toks)
        logits = dev.state.logits.copy()
        first = int(np.argmax(logits))
        outs[knob] = (logits, [first] + dev.decode_greedy(first, n, 3))
        dev.close()
    scale = max(1.0, float(want.std()))
    for knob, (logits, ids) in outs.items():
        err = float(np.abs(logits - want).max())
        print(f"\nfolded norm {wt_variant} NL_FOLD_NORM={knob}: max|gpu-oracle|={err:.2e} (logit std {scale:.2f})")
        assert err <= LOGIT_TOL * scale
        assert ids == ref_ids
    assert not np.array_equal(outs["1"][0], outs["0"][0])
    ref.close()


@pytest.mark.parametrize("stream_scale", [2.0 ** -6, 2.0 ** 15])
def test_folded_rmsnorm_at_small_and_large_residual_magnitudes(hip, orc, tmp_path, monkeypatch, stream_scale):
    # (advisor, round 3) the folded norm hands x * g to the consumer as fp16 hi / lo fragments BEFORE 1 / rms is known.  A
    # residual stream of rms ~1e-2 puts every lo half into the fp16 denormals (absolute instead of relative precision; smaller
    # still and RMSNorm's eps takes the model over), one of rms ~3e4
    # has elements beyond 65504 (inf - inf = NaN).  The producer therefore pre-scales by an exact power of two near the
    # token's previous 1 / rms (norm_prescale, nl_qgemm.h) and the consumer divides it out of inv: same tolerance against
    # the oracle as at magnitude 1, and as close to the unfolded path as there.
    shape = synth.ModelShape("fold_mag", 3, 512, 8, 4, 1024, seq_len=1200, interm=1536, tied=False)
    p = tmp_path / "f.gguf"
    synth.generate_gguf(str(p), shape, "q4_0", 31, stream_scale=stream_scale)
    g = gguf.load_gguf(str(p))
    n = 1100                                   # (every GEMM grid >= 128 workgroups: the folding condition)
    toks = synth.prompt_ids(n, shape.vocab, seed=8)
    ref = orc.OracleModel(g)
    orc.set_threads(min(16, os.cpu_count() or 1))
    for pos, t in enumerate(toks):
        want = ref.forward(t, pos).copy()
    orc.set_threads(1)
    ref.close()
    scale = max(1.0, float(want.std()))
    outs = {}
    for knob in ("1", "0"):
        monkeypatch.setenv("NL_FOLD_NORM", knob)
        dev = hip.load_llama_model(g)
        dev.prefill(toks)
        outs[knob] = dev.state.logits.copy()
        dev.close()
        assert np.isfinite(outs[knob]).all(), knob
        err = float(np.abs(outs[knob] - want).max())
        print(f"\nfolded norm, stream x {stream_scale:g}, NL_FOLD_NORM={knob}: max|gpu-oracle|={err:.2e} (logit std {scale:.2f})")
        assert err <= LOGIT_TOL * scale
    assert not np.array_equal(outs["1"], outs["0"])          # (the folded path is the one that ran)


def test_mini_full_size_long_prefill_matches_oracle(hip, orc, tmp_path, monkeypatch):
    # BASELINE.json configs[2] at its own shape (mini, 173M, Q4_0): a 1920-token prompt through the matrix-core
    # path in ONE step vs the CPU oracle fed token by token (1920 prompt positions for parity), then a
    # greedy continuation on the decode path from the cache the prefill wrote.
    shape = synth.TIERS["mini"]
    p = tmp_path / "mini.gguf"
    synth.generate_gguf(str(p), shape, "q4_0", mode="float")
    g = gguf.load_gguf(str(p))
    dev = hip.load_llama_model(g)
    ref = orc.OracleModel(g)
    toks = synth.prompt_ids(1920, shape.vocab, seed=6)
    orc.set_threads(min(32, os.cpu_count() or 1))
    for pos, t in enumerate(toks):
        want = ref.forward(t, pos)
    want = want.copy()              # (the oracle reuses its logits buffer)
    dev.prefill(toks)
    err = float(np.abs(dev.state.logits - want).max())
    scale = max(1.0, float(want.std()))
    print(f"\nmini 1920-token prefill: max|gpu-oracle|={err:.2e} (logit std {scale:.2f})")
    assert err <= LOGIT_TOL * scale
    cur = int(np.argmax(want))
    ids = dev.decode_greedy(cur, len(toks), 8)
    ref_ids = []
    for k in range(8):
        cur = int(np.argmax(ref.forward(cur, len(toks) + k)))
        ref_ids.append(cur)
    orc.set_threads(1)
    assert ids == ref_ids
    # ... and the single-product precision mode (NL_PREFILL_PRECISION=fp16x1: the long-prompt GEMMs and the prompt attention
    # multiply only the fp16 hi half of every activation / probability -- half / a third of the matrix work).  Its OWN stated
    # tolerance: logits within FP16X1_TOL * max(1, std) of the oracle, the same greedy continuation wherever the oracle's own
    # top-2 margin exceeds twice that tolerance.
    monkeypatch.setenv("NL_PREFILL_PRECISION", "fp16x1")
    dev.reset()
    dev.prefill(toks)
    err1 = float(np.abs(dev.state.logits - want).max())
    print(f"mini 1920-token prefill, fp16x1: max|gpu-oracle|={err1:.2e} (logit std {scale:.2f}; hi/lo mode {err:.2e})")
    assert err1 <= FP16X1_TOL * scale
    assert err1 > err                       # (the mode really took the single-product path)
    top2 = np.sort(want)[-2:]
    if top2[1] - top2[0] > 2 * FP16X1_TOL * scale:
        assert int(np.argmax(dev.state.logits)) == int(np.argmax(want))
        assert dev.decode_greedy(int(np.argmax(want)), len(toks), 8) == ref_ids
    dev.close()


def test_full_length_prefill_is_consistent_with_the_decode_path(hip, tmp_path):
    # BASELINE size (2047 prompt positions, the most the Go loop prefills): properties that do not need the oracle.
    # (1) one 2047-token step == a 1000-token step followed by a 1047-token step (pos0 > 0, different tiling);
    # (2) == 2040 tokens through the matrix-core path + 7 tokens through the single-token decode kernels.
    shape = synth.TIERS["mini"]
    p = tmp_path / "mini.gguf"
    synth.generate_gguf(str(p), shape, "q4_0", mode="qrand")
    g = gguf.load_gguf(str(p))
    toks = synth.prompt_ids(2047, shape.vocab, seed=8)
    a = hip.load_llama_model(g)
    a.prefill(toks)
    la = a.state.logits.copy()
    scale = max(1.0, float(la.std()))
    a.reset()
    a.prefill(toks[:1000], want_logits=False)
    a.prefill(toks[1000:], pos0=1000)
    assert np.abs(a.state.logits - la).max() <= LOGIT_TOL * scale
    a.reset()
    a.prefill(toks[:2040], want_logits=False)
    for i in range(2040, 2047):
        a.forward(toks[i], i)
    assert np.abs(a.state.logits - la).max() <= LOGIT_TOL * scale
    assert int(np.argmax(a.state.logits)) == int(np.argmax(la))
    from nanollama_amd._lib import NlError
    with pytest.raises(NlError, match="exceeds seq_len"):
        a.prefill([1, 2], pos0=2047)
    a.close()


def test_forward_batch_matches_individual_forwards(hip):
    g = gguf.load_gguf(os.path.join(GOLDEN, "tiny_q8_0.gguf"))
    dev = hip.load_llama_model(g, max_streams=4)
    solo = hip.load_llama_model(g, max_streams=4)
    seqs = [[1, 7, 8, 9], [1, 300, 301, 302], [1, 44, 45, 46]]   # ragged: stream 2 joins late
    for step in range(4):
        streams = [s for s in range(3) if not (s == 2 and step == 0)]
        toks = [seqs[s][step - (1 if s == 2 else 0)] for s in streams]
        pos = [step - (1 if s == 2 else 0) for s in streams]
        ids, lg = dev.forward_batch(streams, toks, pos, want_logits=True)
        for k, s in enumerate(streams):
            solo.forward(toks[k], pos[k], stream=s)
            assert np.abs(lg[k] - solo.state.logits).max() <= LOGIT_TOL
            assert ids[k] == int(np.argmax(solo.state.logits))
    from nanollama_amd._lib import NlError
    with pytest.raises(NlError, match="twice"):
        dev.forward_batch([0, 0], [1, 1], [0, 0])
    with pytest.raises(NlError, match="max_streams"):
        dev.forward_batch([0, 1, 2, 3, 0], [1] * 5, [0] * 5)
    assert dev.forward_batch([], [], []) == ([], None)
    dev.close(); solo.close()


@pytest.mark.parametrize("variant", ["conj_gqa", "bias_mha", "hd32_g3"])
def test_batched_decode_rotates_inside_the_attention_launch(hip, orc, tmp_path, monkeypatch, variant):
    # decode batches below position 128: the attention launch sums the Q|K|V GEMM's split-K slabs, adds the biases,
    # rotates (standard / conjugate RoPE) and stores the K / V rows itself (attn_rope_prologue) -- no brope_kv launch.
    # Five streams at ragged positions against the oracle, and bit-identical to the same steps with NL_ROPE_IN_ATTN=0
    # (the separate brope_kv launch does the same arithmetic in the same order).
    shape = {"conj_gqa": synth.ModelShape("ria_conj", 2, 256, 4, 2, 512, seq_len=96, interm=512, rope_conjugate=True),
             "bias_mha": synth.ModelShape("ria_bias", 2, 192, 3, 3, 512, seq_len=96, attn_bias=True),
             "hd32_g3": synth.ModelShape("ria_hd32", 2, 192, 6, 2, 512, seq_len=96, interm=384)}[variant]
    p = tmp_path / "r.gguf"
    synth.generate_gguf(str(p), shape, "q4_0" if variant != "bias_mha" else "q8_0", 43)
    g = gguf.load_gguf(str(p))
    ns, nsteps = 5, 6
    rng = np.random.Generator(np.random.PCG64(77))
    start = [0, 3, 9, 1, 20]
    seqs = [[int(t) for t in rng.integers(3, shape.vocab, size=start[s] + nsteps)] for s in range(ns)]
    refs = []
    for s in range(ns):
        ref = orc.OracleModel(g)
        lg = [ref.forward(t, pos).copy() for pos, t in enumerate(seqs[s])]
        refs.append(lg[start[s]:])
        ref.close()
    outs = {}
    for knob in ("1", "0"):
        monkeypatch.setenv("NL_ROPE_IN_ATTN", knob)
        dev = hip.load_llama_model(g, max_streams=ns)
        for s in range(ns):
            for pos in range(start[s]):
                dev.forward(seqs[s][pos], pos, stream=s)
        got = []
        for k in range(nsteps):
            ids, lg = dev.forward_batch(list(range(ns)), [seqs[s][start[s] + k] for s in range(ns)],
                                        [start[s] + k for s in range(ns)], want_logits=True)
            got.append(lg.copy())
            for s in range(ns):
                assert np.abs(lg[s] - refs[s][k]).max() <= LOGIT_TOL * max(1.0, float(refs[s][k].std())), (knob, s, k)
        outs[knob] = np.stack(got)
        dev.close()
    assert np.array_equal(outs["1"], outs["0"])


@pytest.mark.parametrize("variant", ["conj_gqa", "bias_mha"])
def test_batched_decode_rotates_inside_the_split_attention_launch(hip, orc, tmp_path, monkeypatch, variant):
    # decode batches at positions >= 128 (several 128-key splits per token): every split's workgroup rotates its own copy
    # of q, the split that holds the step's position also rotates k, stores the K / V rows and stages its own K -- still no
    # brope_kv launch.  Four streams at ragged positions that cross the 128- and 256-key boundaries during the run, against
    # the oracle and bit-identical to the same steps with NL_ROPE_IN_ATTN=0.
    shape = {"conj_gqa": synth.ModelShape("rsa_conj", 2, 256, 4, 2, 512, seq_len=320, interm=256, rope_conjugate=True),
             "bias_mha": synth.ModelShape("rsa_bias", 2, 192, 3, 3, 512, seq_len=320, interm=256, attn_bias=True)}[variant]
    p = tmp_path / "r.gguf"
    synth.generate_gguf(str(p), shape, "q4_0" if variant != "bias_mha" else "q8_0", 47)
    g = gguf.load_gguf(str(p))
    ns, nsteps = 4, 5
    rng = np.random.Generator(np.random.PCG64(78))
    start = [125, 127, 254, 300]
    seqs = [[int(t) for t in rng.integers(3, shape.vocab, size=start[s] + nsteps)] for s in range(ns)]
    refs = []
    for s in range(ns):
        ref = orc.OracleModel(g)
        lg = [ref.forward(t, pos).copy() for pos, t in enumerate(seqs[s])]
        refs.append(lg[start[s]:])
        ref.close()
    outs = {}
    for knob in ("1", "0"):
        monkeypatch.setenv("NL_ROPE_IN_ATTN", knob)
        dev = hip.load_llama_model(g, max_streams=ns)
        for s in range(ns):
            dev.prefill(seqs[s][:start[s]], stream=s, want_logits=False)
        got = []
        for k in range(nsteps):
            ids, lg = dev.forward_batch(list(range(ns)), [seqs[s][start[s] + k] for s in range(ns)],
                                        [start[s] + k for s in range(ns)], want_logits=True)
            got.append(lg.copy())
            for s in range(ns):
                assert np.abs(lg[s] - refs[s][k]).max() <= LOGIT_TOL * max(1.0, float(refs[s][k].std())), (knob, s, k)
                assert ids[s] == int(np.argmax(refs[s][k]))
        outs[knob] = np.stack(got)
        dev.close()
    assert np.array_equal(outs["1"], outs["0"])


@pytest.mark.parametrize("heads,kv", [(4, 4), (8, 4)])
def test_batched_decode_at_long_contexts(hip, orc, tmp_path, heads, kv):
    # 64 streams at ragged positions around 400-520 (four / five 128-key splits per token, streams 0 and 1 cross into a
    # fifth split during the run): split attention + brope_kv + battn_merge; three streams against the oracle.
    shape = synth.ModelShape(f"lc_{heads}_{kv}", 2, heads * 64, heads, kv, 512, seq_len=640, interm=128)
    p = tmp_path / "l.gguf"
    synth.generate_gguf(str(p), shape, "q8_0", 53)
    g = gguf.load_gguf(str(p))
    ns, nsteps = 64, 3
    rng = np.random.Generator(np.random.PCG64(79))
    start = [int(x) for x in rng.integers(390, 520, size=ns)]
    start[0], start[1] = 510, 511
    seqs = [[int(t) for t in rng.integers(3, shape.vocab, size=start[s] + nsteps)] for s in range(ns)]
    check = [0, 1, 37]
    refs = {}
    for s in check:
        ref = orc.OracleModel(g)
        lg = [ref.forward(t, pos).copy() for pos, t in enumerate(seqs[s])]
        refs[s] = lg[start[s]:]
        ref.close()
    dev = hip.load_llama_model(g, max_streams=ns)
    for s in range(ns):
        dev.prefill(seqs[s][:start[s]], stream=s, want_logits=False)
    for k in range(nsteps):
        ids, lg = dev.forward_batch(list(range(ns)), [seqs[s][start[s] + k] for s in range(ns)],
                                    [start[s] + k for s in range(ns)], want_logits=True)
        for s in check:
            assert np.abs(lg[s] - refs[s][k]).max() <= LOGIT_TOL * max(1.0, float(refs[s][k].std())), (s, k)
            assert ids[s] == int(np.argmax(refs[s][k]))
    dev.close()


def test_batched_decode_across_the_128_position_split(hip, orc, tmp_path):
    # three streams stepped together from position 0 to 135: below 128 every row has one attention split and the
    # attention kernel writes the WO fragments itself; from 128 on the split partials go through battn_merge.
    # Stream 1 is checked against the oracle on both sides of the boundary.
    shape = synth.ModelShape("split_probe", 2, 256, 4, 2, 1024, seq_len=192, interm=768)
    p = tmp_path / "s.gguf"
    synth.generate_gguf(str(p), shape, "q8_0", 37)
    g = gguf.load_gguf(str(p))
    dev = hip.load_llama_model(g, max_streams=3)
    ref = orc.OracleModel(g)
    rng = np.random.Generator(np.random.PCG64(12))
    seqs = rng.integers(3, shape.vocab, size=(3, 136))
    worst = 0.0
    for step in range(136):
        ids, lg = dev.forward_batch([0, 1, 2], [int(seqs[s, step]) for s in range(3)], [step] * 3, want_logits=True)
        want = ref.forward(int(seqs[1, step]), step)
        if step >= 120:
            worst = max(worst, float(np.abs(lg[1] - want).max()))
            assert ids[1] == int(np.argmax(want)), step
    print(f"\n3 streams across pos 128: max|gpu-oracle|={worst:.2e}")
    assert worst <= LOGIT_TOL
    dev.close()


def test_64_concurrent_streams_match_oracle(hip, orc, tmp_path):
    # BASELINE config 4 in miniature: 64 decode streams stepped together (GQA model, Q4_0), every stream's
    # logits checked against its own oracle run
    shape = synth.ModelShape("batch_probe", 2, 256, 4, 2, 1024, seq_len=64, interm=768)
    p = tmp_path / "b.gguf"
    synth.generate_gguf(str(p), shape, "q4_0", 31)
    g = gguf.load_gguf(str(p))
    dev = hip.load_llama_model(g, max_streams=64)
    rng = np.random.Generator(np.random.PCG64(8))
    seqs = rng.integers(3, shape.vocab, size=(64, 6))
    refs = [orc.OracleModel(g) for _ in range(4)]   # spot-check 4 of the 64 streams step by step
    check = [0, 21, 42, 63]
    worst = 0.0
    for step in range(6):
        ids, lg = dev.forward_batch(list(range(64)), [int(seqs[s, step]) for s in range(64)], [step] * 64, want_logits=True)
        for k, s in enumerate(check):
            want = refs[k].forward(int(seqs[s, step]), step)
            worst = max(worst, float(np.abs(lg[s] - want).max()))
            assert ids[s] == int(np.argmax(want))
    print(f"\n64 streams: max|gpu-oracle|={worst:.2e}")
    assert worst <= LOGIT_TOL
    dev.close()


def test_goldie_full_shape_64_streams_match_oracle(hip, orc, tmp_path):
    # BASELINE.json configs[3] at its OWN shape: goldie (841M: 28 layers, D 1536, 24 heads / 6 kv heads -> G = 4,
    # FFN 4096, vocabulary 48000) Q4_0 with 64 decode streams stepped together -- the batched GQA attention
    # instantiation and the split-K GEMM grids bench.py's goldie line launches (go/model.go:510-612, one layer,
    # batched).  Three streams are checked against their own oracle run: stream 5 at positions 0.., stream 40 a
    # few positions in, stream 63 walking 122..129 -- steps 0-5 have every row below position 128 (one attention
    # split, the attention kernel writes the WO fragments itself), steps 6-7 cross the 128-position split.
    shape = synth.TIERS["goldie"]
    p = os.path.join(os.environ.get("NL_BENCH_DIR", "/tmp"), "nl_bench_goldie_q4_0_qrand.gguf")   # bench.py's file when present
    if not os.path.exists(p):
        synth.generate_gguf(p + ".tmp", shape, "q4_0", mode="qrand")
        os.replace(p + ".tmp", p)
    g = gguf.load_gguf(p)
    assert (g.meta.num_layers, g.meta.embed_dim, g.meta.num_heads, g.meta.num_kv_heads, g.meta.interm_size,
            g.meta.vocab_size) == (28, 1536, 24, 6, 4096, 48000)
    ns, nsteps = 64, 8
    rng = np.random.Generator(np.random.PCG64(64))
    start = [int(v) for v in rng.integers(0, 24, size=ns)]   # ragged: every stream at its own position
    check = {5: 0, 40: 7, 63: 122}
    for s, p0 in check.items():
        start[s] = p0
    seqs = [[int(t) for t in rng.integers(3, shape.vocab, size=start[s] + nsteps)] for s in range(ns)]
    orc.set_threads(min(32, os.cpu_count() or 1))
    wants = {}
    for s in check:
        ref = orc.OracleModel(g)
        for pos, t in enumerate(seqs[s]):
            lg = ref.forward(t, pos)
            if pos >= start[s]:
                wants[(s, pos - start[s])] = lg.copy()
        ref.close()
    orc.set_threads(1)
    dev = hip.load_llama_model(g, max_streams=ns)
    for s in range(ns):
        if start[s]:
            dev.prefill(seqs[s][:start[s]], stream=s, want_logits=False)
    worst, scale = 0.0, 1.0
    for k in range(nsteps):
        ids, lg = dev.forward_batch(list(range(ns)), [seqs[s][start[s] + k] for s in range(ns)],
                                    [start[s] + k for s in range(ns)], want_logits=True)
        for s in check:
            want = wants[(s, k)]
            scale = max(scale, float(want.std()))
            worst = max(worst, float(np.abs(lg[s] - want).max()))
            top2 = np.partition(want, -2)[-2:]
            if float(top2[1] - top2[0]) > 10 * LOGIT_TOL * scale:   # (a tie inside the tolerance may go either way)
                assert ids[s] == int(orc.argmax(want)), (s, k)
    print(f"\ngoldie x 64 streams: max|gpu-oracle|={worst:.2e} (logit std {scale:.2f})")
    assert worst <= LOGIT_TOL * scale
    dev.close()


@pytest.mark.parametrize("wtype", ["q4_0", "q8_0", "f16"])
@pytest.mark.parametrize("rows,cols,ntok", [(128, 256, 64), (576, 576, 5), (1536, 576, 64), (192, 768, 17),
                                             (100, 96, 3), (4096, 4096, 64), (2304, 1536, 130),
                                             # the one / two / four 16-token-tile variants at their boundaries
                                             (320, 320, 16), (320, 320, 32), (320, 320, 33), (768, 2048, 48)])
def test_mfma_multi_token_matmul_matches_oracle(hip, orc, wtype, rows, cols, ntok):
    # the batched / prefill path: fp16-hi/lo activations x exact integer quants on the matrix cores, block
    # sums scaled by d in f32 -- must hold the same GEMV tolerance as the VALU decode kernel
    rng = np.random.Generator(np.random.PCG64(rows + cols * 3 + ntok))
    raw = _rand_matrix(rng, rows, cols, wtype)
    x = rng.standard_normal((ntok, cols), dtype=np.float32)
    x[0, :8] = [0.0, 1e-6, -1e-6, 300.0, -150.0, 1e-3, 7.0, -7.0]      # spread of magnitudes in the hi/lo split
    if ntok > 1:
        x[1, :4] = [6.0e4, -3.5e4, 2.0e-7, 1.0]                          # near the fp16 range limits
    t = synth.WTYPES[wtype]
    got = hip.op_matmul_batch(raw, t, x, rows, cols)
    worst = 0.0
    for n in range(ntok):
        want = orc.matmul(raw, t, x[n], rows, cols)
        # row 1 carries 6e4-sized inputs: both paths are float32 sums of ~3e3-sized terms, so bound it by the
        # float32 summation error of those terms instead of by the (much smaller) result
        scale = np.abs(want) + (np.abs(x[n]).max() * 0.05 * 127 * 1e-2 if n == 1 else 0.0)
        worst = max(worst, float((np.abs(got[n] - want) / (1 + scale)).max()))
    assert worst <= 2e-5, worst


def _rand_blocks(rng, rows, cols, wtype):
    """raw random blocks of the formats that have no reference quantiser (Q5_0, Q4_K, Q6_K)"""
    shape = synth.ModelShape("x", 1, cols, 1, 1, 32)
    return synth._draw_qrand(shape, int(rng.integers(1 << 30)), "t", (rows, cols), "matrix", synth.WTYPES[wtype])


@pytest.mark.parametrize("wtype", ["q5_0", "q4_k", "q6_k"])
@pytest.mark.parametrize("rows,cols", [(16, 256), (100, 512), (576, 768), (1536, 2048), (4096, 4096)])
def test_gemv_remaining_formats_match_oracle(hip, orc, wtype, rows, cols):
    rng = np.random.Generator(np.random.PCG64(rows * 17 + cols))
    raw = _rand_blocks(rng, rows, cols, wtype)
    x = rng.standard_normal(cols, dtype=np.float32)
    t = synth.WTYPES[wtype]
    want = orc.matmul(raw, t, x, rows, cols)
    got = hip.op_matmul(raw, t, x, rows, cols)
    err = np.abs(got - want) / (1 + np.abs(want))
    assert err.max() <= 2e-5, (err.max(), int(err.argmax()))


def test_q5_0_rows_not_multiple_of_64(hip, orc):
    rng = np.random.Generator(np.random.PCG64(12))
    raw = _rand_blocks(rng, 48, 96, "q5_0")
    x = rng.standard_normal(96, dtype=np.float32)
    want = orc.matmul(raw, gguf.GGML_Q5_0, x, 48, 96)
    got = hip.op_matmul(raw, gguf.GGML_Q5_0, x, 48, 96)
    assert np.all(np.abs(got - want) <= 2e-5 * (1 + np.abs(want)))


@pytest.mark.parametrize("wtype", ["q5_0", "q4_k", "q6_k"])
def test_forward_remaining_formats_match_oracle(hip, orc, tmp_path, wtype):
    # whole Forward (embedding rows included) on the formats of go/quant.go:171-484
    shape = synth.ModelShape("kq_probe", 2, 256, 4, 2, 1024, seq_len=64)
    p = tmp_path / "kq.gguf"
    synth.generate_gguf(str(p), shape, wtype, 5)
    worst, scale = _run_teacher_forced(hip, orc, str(p), synth.prompt_ids(20, shape.vocab, seed=6))
    print(f"\n{wtype}: max|gpu-oracle|={worst:.2e} (logit std {scale:.2f})")
    assert worst <= LOGIT_TOL * scale
    g = gguf.load_gguf(str(p))
    dev = hip.load_llama_model(g)
    ref = orc.OracleModel(g)
    prompt = synth.prompt_ids(6, shape.vocab, seed=1)
    dev.prefill(prompt)
    nxt = int(np.argmax(dev.state.logits))
    ref_ids, _ = ref.generate_greedy(prompt, 10)
    assert [nxt] + dev.decode_greedy(nxt, len(prompt), 9) == ref_ids
    dev.close()


def test_kquant_shape_constraint_is_reported(hip):
    from nanollama_amd._lib import NlError
    with pytest.raises(NlError):
        hip.op_matmul(np.zeros(144 // 2 * 16, np.uint8), gguf.GGML_Q4_K, np.zeros(128, np.float32), 16, 128)


@pytest.mark.parametrize("wtype", ["q8_0", "q4_0"])
def test_attention_biases_match_oracle(hip, orc, tmp_path, wtype):
    # Qwen-style attn_q/k/v/output biases (go/model.go:244-247,525-527,591): decode path, prefill path, tp
    shape = synth.ModelShape("bias_probe", 2, 128, 4, 2, 512, seq_len=64, interm=512, attn_bias=True)
    p = tmp_path / "b.gguf"
    synth.generate_gguf(str(p), shape, wtype, 41)
    g = gguf.load_gguf(str(p))
    assert "blk.0.attn_q.bias" in g.tensors and "blk.1.attn_output.bias" in g.tensors
    dev = hip.load_llama_model(g)
    ref = orc.OracleModel(g)
    toks = synth.prompt_ids(14, shape.vocab, seed=9)
    worst = 0.0
    for pos, t in enumerate(toks):
        dev.forward(t, pos)
        want = ref.forward(t, pos).copy()
        worst = max(worst, float(np.abs(dev.state.logits - want).max()))
    assert worst <= LOGIT_TOL
    pre = hip.load_llama_model(g)
    pre.prefill(toks)
    assert np.abs(pre.state.logits - want).max() <= LOGIT_TOL
    grp = hip.LocalTPGroup(g, 2)
    for pos, t in enumerate(toks):
        lg = grp.forward(t, pos)
    assert np.abs(lg - want).max() <= LOGIT_TOL
    dev.close(); pre.close(); grp.close()


def test_extra_tensors_are_skipped_and_biases_are_independently_optional(hip, orc, tmp_path):
    # The Go loader fetches tensors by name and ignores the rest (go/model.go:177-265), and each attention bias is
    # optional on its own (getF32TensorOptional, :244-247): a file with rope_freqs.weight and only a q bias loads.
    shape = synth.ModelShape("bias_probe2", 2, 128, 4, 2, 512, seq_len=64, interm=512, attn_bias=True)
    p = tmp_path / "b2.gguf"
    synth.generate_gguf(str(p), shape, "q8_0", 43)
    g = gguf.load_gguf(str(p))
    g.tensor_order = [n for n in g.tensor_order if not n.endswith(("attn_k.bias", "attn_v.bias", "attn_output.bias"))]
    ref = orc.OracleModel(g)                         # the oracle sees the same reduced tensor list
    g.tensors["rope_freqs.weight"] = g.tensors["output_norm.weight"]
    g.tensor_order = g.tensor_order + ["rope_freqs.weight"]
    dev = hip.load_llama_model(g)
    for pos, t in enumerate(synth.prompt_ids(10, shape.vocab, seed=5)):
        dev.forward(t, pos)
        assert np.abs(dev.state.logits - ref.forward(t, pos)).max() <= LOGIT_TOL
    dev.close(); ref.close()


def test_gamma_injection_matches_oracle(hip, orc, tmp_path):
    # go/gamma.go: embed[token] += gamma[token] for the listed tokens only; f32 and f16 value files
    from nanollama_amd import gamma as gm
    g = gguf.load_gguf(os.path.join(GOLDEN, "tiny_q8_0.gguf"))
    rng = np.random.Generator(np.random.PCG64(77))
    idx = np.array([1, 17, 45, 301, 300], dtype=np.int32)
    for dtype in (np.float32, np.float16):
        vals = (rng.standard_normal((len(idx), 128)) * 0.3).astype(dtype)
        path = tmp_path / f"gamma_{np.dtype(dtype).name}.npz"
        np.savez(path, indices=idx, values=vals, vocab_size=np.array(512), embed_dim=np.array(128))
        ge = gm.load_gamma(str(path))
        assert ge.num_tokens == 5 and ge.embed_dim == 128 and ge.is_f16 == (dtype == np.float16)
        dev = hip.load_llama_model(g)
        plain = hip.load_llama_model(g)
        ref = orc.OracleModel(g)
        dev.set_gamma(ge.indices, ge.values)          # after finalize: plan + graph are rebuilt
        ref.set_gamma(ge.indices, ge.values)
        toks = [1, 17, 9, 301, 5, 45]
        diff_seen = False
        for pos, t in enumerate(toks):
            dev.forward(t, pos)
            plain.forward(t, pos)
            want = ref.forward(t, pos)
            assert np.abs(dev.state.logits - want).max() <= LOGIT_TOL
            diff_seen |= bool(np.abs(dev.state.logits - plain.state.logits).max() > 1e-3)
        assert diff_seen                               # gamma really changes the output
        pre = hip.load_llama_model(g)
        pre.set_gamma(ge.indices, ge.values)
        pre.prefill(toks)                              # multi-token path applies it too
        assert np.abs(pre.state.logits - want).max() <= LOGIT_TOL
        dev.set_gamma([], np.zeros((0, 128), np.float32))   # removing it restores the plain model
        dev.forward(17, 0); plain.forward(17, 0)
        assert dev.state.logits.tobytes() == plain.state.logits.tobytes()
        with pytest.raises(ValueError):
            dev.set_gamma([1], np.zeros((1, 64), np.float32))   # embed_dim mismatch, go/main.go:75-77
        # token ids outside the vocabulary never match a lookup in the Go map (go/gamma.go IndexMap): they are
        # dropped, the in-range rows apply, and the engine keeps running on valid tables
        wide = np.concatenate([idx, np.array([100000, -3], np.int32)])
        wvals = np.concatenate([vals, np.ones((2, 128), dtype)])
        dev.set_gamma(wide, wvals)
        ref2 = orc.OracleModel(g)
        ref2.set_gamma(ge.indices, ge.values)
        for pos, t in enumerate(toks):
            dev.forward(t, pos)
            assert np.abs(dev.state.logits - ref2.forward(t, pos)).max() <= LOGIT_TOL
        ref2.close()
        dev.close(); plain.close(); pre.close()


def test_cli_end_to_end_text_generation(hip, orc, tmp_path, capsys):
    # `nanollama --model ... --prompt ... --temp 0 --rep-penalty 1.0` through tokenizer, engine and the C ABI:
    # the generated text must be the oracle's greedy ids decoded with the same vocabulary
    from nanollama_amd import cli
    from nanollama_amd.tokenizer import Tokenizer
    path = os.path.join(GOLDEN, "tiny_q8_0.gguf")
    g = gguf.load_gguf(path)
    tok = Tokenizer(g.meta)
    prompt_ids = [1, 40, 41, 42, 300]
    prompt = "".join(g.meta.token_list[i] for i in prompt_ids[1:])   # synthetic vocab: one code point per token
    assert tok.encode(prompt, True) == prompt_ids
    rc = cli.main(["--model", path, "--prompt", prompt, "--temp", "0", "--rep-penalty", "1.0", "--rep-window", "128",
                   "--max-tokens", "12"])
    assert rc == 0
    out = capsys.readouterr().out
    ref_ids, _ = orc.OracleModel(g).generate_greedy(prompt_ids, 12)
    want = tok.decode(ref_ids)
    assert want in out and "[12 tokens," in out and "[model] loaded: 2 layers, 128 dim" in out


def test_sampling_paths_are_well_formed(hip):
    # top-k / top-p / repetition penalty run on host over device logits (go/main.go:177-187,294-398); with a fixed
    # seed they are deterministic, stay inside the vocabulary and differ from greedy at temperature > 0
    from nanollama_amd.engine import Engine, GenParams
    g = gguf.load_gguf(os.path.join(GOLDEN, "tiny_q4_0.gguf"))
    dev = hip.load_llama_model(g)
    runs = []
    for _ in range(2):
        eng = Engine(dev, eos_id=g.meta.eos_id, rep_penalty=1.15, rep_window=64, seed=1234)
        runs.append(eng.generate_ids([1, 5, 6], GenParams(max_tokens=20, temperature=0.9, top_p=0.9)))
    assert runs[0] == runs[1] and all(0 <= t < 512 for t in runs[0]) and len(runs[0]) == 20
    eng = Engine(dev, eos_id=g.meta.eos_id, rep_penalty=1.15, rep_window=64, seed=7)
    topk = eng.generate_ids([1, 5, 6], GenParams(max_tokens=20, temperature=0.9, top_p=1.0, top_k=5))
    assert len(topk) == 20 and eng.last_tokens == 20
    greedy = Engine(dev, eos_id=g.meta.eos_id, rep_penalty=1.0).generate_ids([1, 5, 6], GenParams(max_tokens=20, temperature=0.0))
    assert greedy != runs[0]
    # --rep-window caps the reference's token counter (go/main.go:198-200,223): reproduce the quirk
    eng = Engine(dev, eos_id=g.meta.eos_id, rep_penalty=1.15, rep_window=8, seed=1)
    eng.generate_ids([1, 5, 6], GenParams(max_tokens=20, temperature=0.5))
    assert eng.last_tokens == 8
    dev.close()


# ---- fused attention block (nanollama_amd/csrc/nl_block.h): small models, short contexts -------------------------

def _kinds(dev, pos):
    return {k: c for k, (ms, c) in dev.profile_forward(5, pos, iters=1).items() if c}


@pytest.mark.parametrize("ffn", ["ffn_block", "gate_up+down"])
@pytest.mark.parametrize("wtype", ["q8_0", "q4_0"])
@pytest.mark.parametrize("variant", ["mha", "gqa", "qknorm_conj_bias"])
def test_fused_attention_block_matches_oracle(hip, orc, tmp_path, monkeypatch, wtype, variant, ffn):
    # one launch per layer for RMSNorm + Q/K/V + RoPE + KV store + attention + WO (go/model.go:517-594) against the
    # CPU oracle, teacher-forced across the 128-position pass boundary (NL_FUSED_MAX_POS lifted so the block runs at
    # every position), for MHA, a GQA group of 2 and the optional QK-norm / conjugate RoPE / bias branches
    shape = {"mha": synth.ModelShape("fb_mha", 3, 192, 3, 3, 1024, seq_len=160),
             "gqa": synth.ModelShape("fb_gqa", 2, 256, 4, 2, 512, seq_len=160, interm=512),
             "qknorm_conj_bias": synth.ModelShape("fb_var", 2, 256, 4, 2, 512, seq_len=160, interm=512, qk_norm=True,
                                                  rope_conjugate=True, attn_bias=True)}[variant]
    p = tmp_path / "m.gguf"
    synth.generate_gguf(str(p), shape, wtype, 61)
    g = gguf.load_gguf(str(p))
    monkeypatch.setenv("NL_FUSED_MAX_POS", "4096")
    # ffn_block: the feed-forward half is one launch as well (gate/up rows, SiLU * up, exchange, W_down column slice per
    # cluster of eight workgroups); the other case keeps gate/up and down as the two GEMV launches
    monkeypatch.setenv("NL_FUSED_FFN", "1" if ffn == "ffn_block" else "0")
    dev = hip.load_llama_model(g)
    monkeypatch.setenv("NL_FUSED_ATTN", "0")
    plain = hip.load_llama_model(g)
    ref = orc.OracleModel(g)
    toks = synth.prompt_ids(150, shape.vocab, seed=21)
    worst = gap = 0.0
    for pos, t in enumerate(toks):
        dev.forward(t, pos)
        plain.forward(t, pos)
        want = ref.forward(t, pos)
        worst = max(worst, float(np.abs(dev.state.logits - want).max()) / max(1.0, float(want.std())))
        gap = max(gap, float(np.abs(dev.state.logits - plain.state.logits).max()))
    print(f"\nfused block {variant}/{wtype}: max|gpu-oracle|={worst:.2e}, max|fused-unfused|={gap:.2e}")
    assert worst <= LOGIT_TOL
    k = _kinds(dev, 20)
    assert k.get("attn_block") == shape.n_layer and "qkv_rope" not in k and "wo_resid" not in k, k
    if ffn == "ffn_block":   # (the last layer keeps the two GEMV launches so that the LM head reads a complete residual stream)
        assert k.get("ffn_block") == shape.n_layer - 1 and k.get("gate_up_swiglu") == 1 and k.get("down_resid") == 1, k
    else:
        assert "ffn_block" not in k and k.get("gate_up_swiglu") == shape.n_layer, k
    assert "attn_block" not in _kinds(plain, 20)
    # the K/V rows the block stored are the ones the five-launch plan stores
    n = shape.n_layer * shape.n_kv_head * shape.seq_len * 64
    for which in ("k_cache", "v_cache"):
        a, b = dev.debug_read(which, n).reshape(-1, shape.seq_len, 64), plain.debug_read(which, n).reshape(-1, shape.seq_len, 64)
        assert np.abs(a[:, :149] - b[:, :149]).max() <= 2e-5
    dev.close(); plain.close(); ref.close()


@pytest.mark.parametrize("case", ["small_tier_blocks", "wide_tier_projection_attention", "wide_tier_attention_wo"])
def test_fused_plan_timeout_falls_back_to_the_general_plan(hip, orc, tmp_path, monkeypatch, case):
    # HIP promises nothing about which workgroups of a launch are resident together; on a shared GPU a cluster member can
    # wait for a peer that was never dispatched.  NL_FUSED_SPIN_LIMIT=0 makes every exchange poll give up at once: the
    # call must still succeed (Forward cannot fail in the reference, go/model.go:490) with the oracle's logits -- the
    # step is redone on the general plan -- the handle must stay on the general plan afterwards, and nl_last_error
    # carries the one-time note.  Covered for every entry point that steps the plan.
    if case == "small_tier_blocks":
        shape = synth.ModelShape("fb_fallback", 3, 192, 3, 3, 1024, seq_len=96)
    else:
        shape = synth.ModelShape("fb_fallback_wide", 2, 2048, 32, 8, 1024, seq_len=96, interm=1024)
        if case == "wide_tier_projection_attention":
            monkeypatch.setenv("NL_ATTN_WO", "0")                # mode 2; the default on one GPU is mode 4 (WO in the launch)
    p = tmp_path / "m.gguf"
    synth.generate_gguf(str(p), shape, "q4_0", 67, mode="qrand" if case != "small_tier_blocks" else "float")
    g = gguf.load_gguf(str(p))
    ref = orc.OracleModel(g)
    toks = synth.prompt_ids(12, shape.vocab, seed=3)
    wants = [ref.forward(t, pos).copy() for pos, t in enumerate(toks)]
    ref_ids, _ = ref.generate_greedy(toks, 20)
    probe = hip.load_llama_model(g)
    assert "attn_block" in _kinds(probe, 4)                      # the fused plan is what normally serves this model
    probe.close()
    monkeypatch.setenv("NL_FUSED_SPIN_LIMIT", "0")
    monkeypatch.setenv("NL_QUIET", "1")

    def fresh():
        dev = hip.load_llama_model(g)
        assert dev.last_error() == ""
        return dev

    def retired(dev):
        assert "timed out" in dev.last_error() and "general plan" in dev.last_error()
        assert "attn_block" not in _kinds(dev, 4) and "ffn_block" not in _kinds(dev, 4)

    tol = lambda w: LOGIT_TOL * max(1.0, float(w.std()))
    dev = fresh()                                                # nl_forward
    for pos, t in enumerate(toks):
        dev.forward(t, pos)
        assert np.abs(dev.state.logits - wants[pos]).max() <= tol(wants[pos]), pos
    retired(dev)
    dev.close()
    dev = fresh()                                                # nl_prefill, two tokens: the single-token plan
    dev.prefill(toks[:2])
    assert np.abs(dev.state.logits - wants[1]).max() <= tol(wants[1])
    retired(dev)
    dev.close()
    dev = fresh()                                                # nl_forward_batch below the batch threshold
    ids, lg = dev.forward_batch([0], [toks[0]], [0], want_logits=True)
    assert np.abs(lg[0] - wants[0]).max() <= tol(wants[0]) and ids[0] == int(np.argmax(wants[0]))
    retired(dev)
    dev.close()
    dev = fresh()                                                # nl_forward_argmax
    assert dev.forward_argmax(toks[0], 0) == int(np.argmax(wants[0]))
    retired(dev)
    dev.close()
    dev = fresh()                                                # nl_decode_greedy: the chained 16-step graph hits the timeout
    dev.prefill(toks)                                            # (12 tokens: the multi-token path, no fused launch yet)
    assert dev.last_error() == ""
    first = int(np.argmax(dev.state.logits))
    assert [first] + dev.decode_greedy(first, len(toks), 19) == ref_ids
    retired(dev)
    dev.close()
    # nl_sample_decode: same uniforms as a handle that ran the general plan from the start -> identical ids and window
    us = np.random.default_rng(9).random(24, dtype=np.float32)
    dev = fresh()
    dev.prefill(toks)
    got = dev.sample_decode(len(toks), 24, 0.9, 0.9, 50, 1.15, 8, us, [3, 4])
    retired(dev)
    dev.close()
    monkeypatch.delenv("NL_FUSED_SPIN_LIMIT")
    monkeypatch.setenv("NL_FUSED_ATTN", "0")
    plain = hip.load_llama_model(g)
    plain.prefill(toks)
    assert plain.sample_decode(len(toks), 24, 0.9, 0.9, 50, 1.15, 8, us, [3, 4]) == got
    assert plain.last_error() == ""
    plain.close(); ref.close()


def test_greedy_chain_switches_from_the_fused_plan_to_the_split_attention_plan(hip, orc, tmp_path):
    # the block serves positions below NL_FUSED_MAX_POS (default 512 for the per-head blocks); a chained greedy run that
    # crosses it keeps producing the oracle's ids (16-step graphs of either plan, single steps at the seam)
    shape = synth.ModelShape("fb_switch", 2, 192, 3, 3, 1024, seq_len=608)
    p = tmp_path / "m.gguf"
    synth.generate_gguf(str(p), shape, "q8_0", 71)
    g = gguf.load_gguf(str(p))
    dev = hip.load_llama_model(g)
    ref = orc.OracleModel(g)
    prompt = synth.prompt_ids(458, shape.vocab, seed=3)
    want, _ = ref.generate_greedy(prompt, 100)
    dev.prefill(prompt)
    first = int(np.argmax(dev.state.logits))
    got = [first] + dev.decode_greedy(first, len(prompt), 99)
    assert got == want
    assert "attn_block" in _kinds(dev, 100) and "attn_block" not in _kinds(dev, 580)
    dev.close(); ref.close()


def test_wide_tier_chain_crosses_the_attention_passes_and_the_plan_switch(hip, orc, tmp_path):
    # fused mode 4 serves positions below 768: its attention runs 1 .. 3 passes of 256 positions inside the launch (all sixteen
    # wavefronts of the runner hold cache rows).  A prompt of 718 tokens, then 100 chained greedy steps: the third pass, the switch
    # to the split-attention plan at 768 (16-step graphs of either plan, single steps at the seam) -- the oracle's ids throughout,
    # and its logits at positions in every pass
    shape = synth.ModelShape("m4_switch", 2, 1024, 16, 4, 1024, seq_len=864, interm=2048)
    p = tmp_path / "m.gguf"
    synth.generate_gguf(str(p), shape, "q4_0", 73, mode="qrand")
    g = gguf.load_gguf(str(p))
    dev = hip.load_llama_model(g)
    assert dev.plan_info()["fused_mode"] == 4 and dev.plan_info()["fused_max_pos"] == 768
    ref = orc.OracleModel(g)
    orc.set_threads(min(16, os.cpu_count() or 1))
    prompt = synth.prompt_ids(718, shape.vocab, seed=5)
    want, _ = ref.generate_greedy(prompt, 100)
    dev.prefill(prompt)
    first = int(np.argmax(dev.state.logits))
    got = [first] + dev.decode_greedy(first, len(prompt), 99)
    assert got == want
    for pos in (130, 255, 256, 300, 511, 512, 700, 767):      # first, second and third pass of the in-launch attention and their edges
        lg = ref.forward(want[3], pos).copy()
        dev.forward(want[3], pos)
        assert np.abs(dev.state.logits - lg).max() <= LOGIT_TOL * max(1.0, float(lg.std())), pos
    orc.set_threads(1)
    assert "attn_block" in _kinds(dev, 100) and "attn_block" in _kinds(dev, 760) and "attn_block" not in _kinds(dev, 800)
    dev.close(); ref.close()


@pytest.mark.parametrize("wo", ["wo_own_launch", "wo_in_the_launch"])
@pytest.mark.parametrize("variant,wtype", [("g4", "q4_0"), ("g4", "q8_0"), ("g8", "q4_0"), ("mha16", "q4_0"),
                                           ("g4_two_tiles", "q4_0"), ("g4_qknorm_conj_bias", "q8_0"), ("goldie_width", "q4_0")])
def test_fused_projection_attention_launch_matches_oracle(hip, orc, tmp_path, monkeypatch, variant, wtype, wo):
    # nl_group.h: Q/K/V + RoPE + KV store + attention as one launch for models too wide for the per-head block
    # (clusters of workgroups per kv group, granule exchange, G attention workgroups), against the oracle across the
    # 128-position pass boundary; "g4_two_tiles" is wide enough that a workgroup holds two tiles and a wavefront two
    # column groups (the 7.9B tier's geometry).  wo_in_the_launch (fused mode 4, the default on one GPU): nl_tp.h's
    # attention half with a direct seam -- WO + residual behind a second exchange in the same launch, no wo_resid launch;
    # wo_own_launch (NL_ATTN_WO=0): mode 2, the WO GEMV merges the launch's partials.
    if wo == "wo_own_launch":
        monkeypatch.setenv("NL_ATTN_WO", "0")
    shape = {"g4": synth.ModelShape("fg_g4", 2, 1024, 16, 4, 512, seq_len=160, interm=1024),
             "goldie_width": synth.ModelShape("fg_goldie", 2, 1536, 24, 6, 512, seq_len=160, interm=4096),   # BASELINE config 4's tier: six column groups, six kv groups
             "g8": synth.ModelShape("fg_g8", 2, 1024, 16, 2, 512, seq_len=160, interm=1024),
             "mha16": synth.ModelShape("fg_mha", 2, 1024, 16, 16, 512, seq_len=160, interm=1024),
             "g4_two_tiles": synth.ModelShape("fg_wide", 2, 2560, 40, 10, 512, seq_len=160, interm=1024),
             "g4_qknorm_conj_bias": synth.ModelShape("fg_var", 2, 1024, 16, 4, 512, seq_len=160, interm=1024, qk_norm=True,
                                                     rope_conjugate=True, attn_bias=True)}[variant]
    p = tmp_path / "m.gguf"
    synth.generate_gguf(str(p), shape, wtype, 63, mode="qrand")
    g = gguf.load_gguf(str(p))
    monkeypatch.setenv("NL_FUSED_MAX_POS", "4096")
    dev = hip.load_llama_model(g)
    monkeypatch.setenv("NL_FUSED_ATTN", "0")
    plain = hip.load_llama_model(g)
    ref = orc.OracleModel(g)
    orc.set_threads(min(16, os.cpu_count() or 1))
    toks = synth.prompt_ids(140, shape.vocab, seed=23)
    worst = gap = 0.0
    for pos, t in enumerate(toks):
        dev.forward(t, pos)
        plain.forward(t, pos)
        want = ref.forward(t, pos)
        worst = max(worst, float(np.abs(dev.state.logits - want).max()) / max(1.0, float(want.std())))
        gap = max(gap, float(np.abs(dev.state.logits - plain.state.logits).max()))
    orc.set_threads(1)
    print(f"\nfused projection+attention {variant}/{wtype}/{wo}: max|gpu-oracle|={worst:.2e}, max|fused-unfused|={gap:.2e}")
    assert worst <= LOGIT_TOL
    k = _kinds(dev, 20)
    assert k.get("attn_block") == shape.n_layer and "qkv_rope" not in k and "attention" not in k, k
    assert k.get("wo_resid") == (shape.n_layer if wo == "wo_own_launch" else None), k
    assert dev.plan_info()["fused_mode"] == (2 if wo == "wo_own_launch" else 4), dev.plan_info()
    assert "attn_block" not in _kinds(plain, 20)      # (both engines now hold the probe's row 20)
    n = shape.n_layer * shape.n_kv_head * shape.seq_len * 64
    for which in ("k_cache", "v_cache"):
        a, b = dev.debug_read(which, n).reshape(-1, shape.seq_len, 64), plain.debug_read(which, n).reshape(-1, shape.seq_len, 64)
        assert np.abs(a[:, :139] - b[:, :139]).max() <= 2e-5
    first = int(np.argmax(dev.state.logits))
    assert dev.decode_greedy(first, 140, 16) == plain.decode_greedy(first, 140, 16)
    dev.close(); plain.close(); ref.close()


@pytest.mark.parametrize("name,wtype", [("d3072_i8192", "q4_0"), ("d4096_i5632", "q4_0"), ("d3072_i8192", "q8_0")])
def test_wide_layer_as_two_launches_matches_oracle(hip, orc, tmp_path, monkeypatch, name, wtype):
    # One GPU, a layer as wide as the 7.9B tier's: projection + attention + WO in one launch (fused mode 4) and gate || up +
    # SwiGLU + down in one launch (wide_ffn_kernel, nl_tp.h: one workgroup per W_down tile, compile-time rounds of gate / up
    # tiles, h through tagged granules) -- 2 launches per layer.  Against the oracle across the 128-position pass boundary,
    # against the same model with the two GEMV launches (NL_WIDE_FFN=0), greedy continuation, and the timeout fallback.
    shape = {"d3072_i8192": synth.ModelShape("wl_a", 2, 3072, 48, 12, 2048, seq_len=160, interm=8192),      # 3 rounds, 32 W_down groups
             "d4096_i5632": synth.ModelShape("wl_b", 2, 4096, 64, 16, 2048, seq_len=160, interm=5632)}[name]  # 2 rounds (the second partial), 22 groups
    p = tmp_path / "m.gguf"
    synth.generate_gguf(str(p), shape, wtype, 91, mode="qrand")
    g = gguf.load_gguf(str(p))
    dev = hip.load_llama_model(g)
    info = dev.plan_info()
    assert info["fused_mode"] == 4 and info["launches_fused"] == 2 * shape.n_layer + 3, info
    monkeypatch.setenv("NL_WIDE_FFN", "0")
    split = hip.load_llama_model(g)
    assert split.plan_info()["launches_fused"] == 3 * shape.n_layer + 3
    monkeypatch.delenv("NL_WIDE_FFN")
    ref = orc.OracleModel(g)
    orc.set_threads(min(16, os.cpu_count() or 1))
    toks = synth.prompt_ids(134, shape.vocab, seed=37)
    worst = gap = 0.0
    for pos, t in enumerate(toks):
        dev.forward(t, pos)
        split.forward(t, pos)
        want = ref.forward(t, pos)
        worst = max(worst, float(np.abs(dev.state.logits - want).max()) / max(1.0, float(want.std())))
        gap = max(gap, float(np.abs(dev.state.logits - split.state.logits).max()))
    orc.set_threads(1)
    print(f"\nwide layer as two launches {name}/{wtype}: max|gpu-oracle|={worst:.2e}, max|fused-split|={gap:.2e}")
    assert worst <= LOGIT_TOL
    first = int(np.argmax(dev.state.logits))
    assert dev.decode_greedy(first, len(toks), 12) == split.decode_greedy(first, len(toks), 12)
    assert dev.last_error() == ""
    dev.close(); split.close()
    # every exchange poll gives up at once: the step is redone on the general plan, same logits as the oracle's
    monkeypatch.setenv("NL_FUSED_SPIN_LIMIT", "0")
    monkeypatch.setenv("NL_QUIET", "1")
    fb = hip.load_llama_model(g)
    ref2 = orc.OracleModel(g)
    for pos, t in enumerate(toks[:3]):
        fb.forward(t, pos)
        want = ref2.forward(t, pos)
        assert np.abs(fb.state.logits - want).max() <= LOGIT_TOL * max(1.0, float(want.std()))
    assert "timed out" in fb.last_error() and fb.plan_info()["fused_mode"] == 0
    fb.close(); ref.close(); ref2.close()


def test_f16_prompt_prefill_runs_on_the_matrix_cores(hip, orc, tmp_path):
    # F16 files (BASELINE.json configs[0]'s format, go/quant.go:527-563) take the multi-token MFMA path too: the
    # fp16 weights are their own exact operand, no scale step; a prompt long enough for the fused RoPE and SwiGLU
    # epilogues and for two attention splits, checked against the oracle and against token-at-a-time decode
    shape = synth.ModelShape("f16_prefill", 2, 256, 4, 2, 1024, seq_len=256, interm=512)
    p = tmp_path / "m.gguf"
    synth.generate_gguf(str(p), shape, "f16", 83)
    g = gguf.load_gguf(str(p))
    toks = synth.prompt_ids(150, shape.vocab, seed=4)
    ref = orc.OracleModel(g)
    for pos, t in enumerate(toks):
        want = ref.forward(t, pos)
    dev = hip.load_llama_model(g)
    dev.prefill(toks)
    scale = max(1.0, float(want.std()))
    d = float(np.abs(dev.state.logits - want).max())
    print(f"\nf16 prefill 150 tokens: max|gpu-oracle|={d:.2e}")
    assert d <= LOGIT_TOL * scale
    step = hip.load_llama_model(g)
    for pos, t in enumerate(toks):
        step.forward(t, pos)
    n = shape.n_layer * shape.n_kv_head * shape.seq_len * 64
    for which in ("k_cache", "v_cache"):
        a, b = dev.debug_read(which, n).reshape(-1, shape.seq_len, 64), step.debug_read(which, n).reshape(-1, shape.seq_len, 64)
        assert np.abs(a[:, :150] - b[:, :150]).max() <= 2e-5
    nxt = int(np.argmax(want))
    assert dev.decode_greedy(nxt, 150, 12) == step.decode_greedy(nxt, 150, 12)
    # 64 concurrent F16 decode streams through the same GEMM
    many = hip.load_llama_model(g, max_streams=8)
    ids, lg = many.forward_batch(list(range(8)), toks[:8], [0] * 8, want_logits=True)
    solo = hip.load_llama_model(g)
    for i in range(8):
        solo.reset()
        solo.forward(toks[i], 0)
        assert np.abs(lg[i] - solo.state.logits).max() <= LOGIT_TOL * scale
    for m in (dev, step, many, solo, ref):
        m.close()


@pytest.mark.parametrize("shape", [synth.ModelShape("fb_streams", 2, 192, 3, 3, 1024, seq_len=96),
                                   synth.ModelShape("fg_streams", 2, 1024, 16, 4, 512, seq_len=96, interm=1024)])
def test_fused_launches_keep_streams_independent(hip, tmp_path, shape):
    # the fused attention launches address the KV cache of the stream named in the control block (several decode
    # streams per engine): two interleaved sequences must not see each other, bitwise, in both fused forms
    p = tmp_path / "m.gguf"
    synth.generate_gguf(str(p), shape, "q8_0", 91, mode="qrand")
    g = gguf.load_gguf(str(p))
    dev = hip.load_llama_model(g, max_streams=2)
    assert "attn_block" in _kinds(dev, 3)
    dev.reset(0); dev.reset(1)
    a, b = synth.prompt_ids(20, shape.vocab, seed=1), synth.prompt_ids(20, shape.vocab, seed=2)
    la = []
    for pos in range(20):
        dev.forward(a[pos], pos, stream=0)
        la.append(dev.state.logits.copy())
        dev.forward(b[pos], pos, stream=1)
    solo = hip.load_llama_model(g)
    for pos in range(20):
        solo.forward(a[pos], pos)
        assert la[pos].tobytes() == solo.state.logits.tobytes(), pos
    dev.close(); solo.close()


def test_batches_above_64_streams_chunked_loop_and_sub_batches_agree_bitwise(hip, tmp_path, monkeypatch):
    # more streams than one 64-token step holds: by default nl_forward_batch walks them in chunks on the engine's stream; with
    # NL_SUB_BATCHES > 1 the chunks step concurrently.  Same arithmetic per stream: bitwise the same logits and ids, and every
    # stream agrees with its own single-stream run.
    shape = synth.ModelShape("over64", 2, 256, 4, 2, 2048, seq_len=64, interm=768)
    p = tmp_path / "m.gguf"
    synth.generate_gguf(str(p), shape, "q4_0", 79, mode="qrand")
    g = gguf.load_gguf(str(p))
    ns, nsteps = 96, 5
    rng = np.random.Generator(np.random.PCG64(67))
    seqs = [[int(t) for t in rng.integers(3, shape.vocab, size=nsteps)] for _ in range(ns)]

    def run(sub):
        if sub:
            monkeypatch.setenv("NL_SUB_BATCHES", str(sub))
        else:
            monkeypatch.delenv("NL_SUB_BATCHES", raising=False)
        dev = hip.load_llama_model(g, max_streams=ns)
        out = []
        for k in range(nsteps):
            ids, lg = dev.forward_batch(list(range(ns)), [seqs[s][k] for s in range(ns)], [k] * ns, want_logits=True)
            out.append((list(ids), lg.copy()))
        dev.close()
        return out

    base, cut = run(0), run(2)
    for k, ((ids_a, lg_a), (ids_b, lg_b)) in enumerate(zip(base, cut)):
        assert ids_a == ids_b, k
        assert lg_a.tobytes() == lg_b.tobytes(), f"step {k}: the sub-batched step differs from the chunked loop"
    monkeypatch.delenv("NL_SUB_BATCHES", raising=False)
    solo = hip.load_llama_model(g)
    for s in (0, 63, 64, 95):
        solo.reset()
        for k in range(nsteps):
            solo.forward(seqs[s][k], k)
            d = float(np.abs(solo.state.logits - base[k][1][s]).max()) / max(1.0, float(solo.state.logits.std()))
            assert d <= 1e-4, (s, k, d)
    solo.close()


@pytest.mark.parametrize("groups", [2, 4])
def test_concurrent_sub_batches_equal_the_one_step_batch_bitwise(hip, tmp_path, monkeypatch, groups):
    # nl_forward_batch cuts a decode batch into groups that step concurrently on their own HIP streams (each group's step a
    # cached hipGraph); a stream's arithmetic does not depend on the group it steps in, so logits and ids are BITWISE those of
    # the whole batch stepped as one (go/model.go:510-612 per stream; go/serve.go:106-108 serialises requests only because it
    # has one CPU engine).  Ragged positions across the 128-position attention split, 64 and 37 streams, graph replays.
    shape = synth.ModelShape("subbatch", 3, 512, 8, 2, 4096, seq_len=192, interm=1536)
    p = tmp_path / "m.gguf"
    synth.generate_gguf(str(p), shape, "q4_0", 77, mode="qrand")
    g = gguf.load_gguf(str(p))
    ns, nsteps = 64, 12
    rng = np.random.Generator(np.random.PCG64(65))
    start = [int(v) for v in rng.integers(0, 30, size=ns)]
    start[3], start[50] = 120, 125                      # these cross position 128 during the run
    seqs = [[int(t) for t in rng.integers(3, shape.vocab, size=start[s] + nsteps)] for s in range(ns)]

    def run(sub):
        monkeypatch.setenv("NL_SUB_BATCHES", str(sub))
        dev = hip.load_llama_model(g, max_streams=ns)
        for s in range(ns):
            if start[s]:
                dev.prefill(seqs[s][:start[s]], stream=s, want_logits=False)
        out = []
        for k in range(nsteps):
            live = list(range(ns)) if k % 3 else list(range(0, ns, 2)) + [1, 3, 5, 7, 9]      # 64 streams, or 37
            ids, lg = dev.forward_batch(live, [seqs[s][start[s] + k] for s in live], [start[s] + k for s in live], want_logits=True)
            out.append((list(ids), lg.copy()))
            for s in set(range(ns)) - set(live):       # keep every stream's cache complete: the skipped ones step alone
                dev.forward(seqs[s][start[s] + k], start[s] + k, stream=s)
        dev.close()
        return out

    one, cut = run(1), run(groups)
    for k, ((ids_a, lg_a), (ids_b, lg_b)) in enumerate(zip(one, cut)):
        assert ids_a == ids_b, k
        assert lg_a.tobytes() == lg_b.tobytes(), f"step {k}: sub-batched logits differ from the one-step batch"
