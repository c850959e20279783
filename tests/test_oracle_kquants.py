"""Pins the oracle's Q5_0 / Q4_K / Q6_K block decoders (go/quant.go:171-484) against an independent numpy
restatement of the published ggml block formats (ggml-quants.c: block_q5_0, block_q4_K + get_scale_min_k4,
block_q6_K).  The reference has neither a quantiser nor test vectors for these three formats, so random
well-formed blocks are decoded by both and compared (float32, a few ulp: the two sides may associate d*sc*q
differently)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import oracle as orc  # noqa: E402

GGML_Q5_0, GGML_Q4_K, GGML_Q6_K = 6, 12, 14


def _f16_bits(rng, n, lo=1e-3, hi=0.5):
    v = rng.uniform(lo, hi, size=n) * rng.choice([-1.0, 1.0], size=n)
    return v.astype(np.float16).view(np.uint16)


def _h(bits):
    return np.asarray(bits, np.uint16).view(np.float16).astype(np.float32)


def np_q5_0(raw, nblk):
    b = raw.reshape(nblk, 22)
    d = _h(b[:, 0:2].copy().view(np.uint16)[:, 0])[:, None]
    qh = b[:, 2:6].copy().view(np.uint32)[:, 0][:, None]
    qs = b[:, 6:22].astype(np.int32)
    j = np.arange(16)[None, :]
    x0 = ((qs & 0xF) | (((qh >> j) & 1) << 4)).astype(np.int32) - 16
    x1 = ((qs >> 4) | (((qh >> (j + 16)) & 1) << 4)).astype(np.int32) - 16
    return np.concatenate([x0, x1], axis=1).astype(np.float32) * d


def np_q4_k(raw, nblk):
    out = np.empty((nblk, 256), np.float32)
    b = raw.reshape(nblk, 144)
    for i in range(nblk):
        d, dmin = _h(b[i, 0:2].copy().view(np.uint16))[0], _h(b[i, 2:4].copy().view(np.uint16))[0]
        sc_raw, q = b[i, 4:16].astype(np.int32), b[i, 16:144].astype(np.int32)

        def scale_min(j):
            if j < 4:
                return sc_raw[j] & 63, sc_raw[j + 4] & 63
            return (sc_raw[j + 4] & 0xF) | ((sc_raw[j - 4] >> 6) << 4), (sc_raw[j + 4] >> 4) | ((sc_raw[j] >> 6) << 4)

        for grp in range(4):   # 64 elements: low nibbles of 32 bytes, then their high nibbles
            s1, m1 = scale_min(2 * grp)
            s2, m2 = scale_min(2 * grp + 1)
            qq = q[32 * grp:32 * grp + 32]
            out[i, 64 * grp:64 * grp + 32] = np.float32(d * np.float32(s1)) * (qq & 0xF).astype(np.float32) - np.float32(dmin * np.float32(m1))
            out[i, 64 * grp + 32:64 * grp + 64] = np.float32(d * np.float32(s2)) * (qq >> 4).astype(np.float32) - np.float32(dmin * np.float32(m2))
    return out


def np_q6_k(raw, nblk):
    out = np.empty((nblk, 256), np.float32)
    b = raw.reshape(nblk, 210)
    for i in range(nblk):
        ql, qh = b[i, 0:128].astype(np.int32), b[i, 128:192].astype(np.int32)
        sc = b[i, 192:208].copy().view(np.int8).astype(np.float32)
        d = _h(b[i, 208:210].copy().view(np.uint16))[0]
        for half in range(2):
            l = np.arange(32)
            lq, hq, s = ql[64 * half:64 * half + 64], qh[32 * half:32 * half + 32], sc[8 * half:8 * half + 8]
            isub = l // 16
            q1 = ((lq[l] & 0xF) | (((hq[l] >> 0) & 3) << 4)) - 32
            q2 = ((lq[l + 32] & 0xF) | (((hq[l] >> 2) & 3) << 4)) - 32
            q3 = ((lq[l] >> 4) | (((hq[l] >> 4) & 3) << 4)) - 32
            q4 = ((lq[l + 32] >> 4) | (((hq[l] >> 6) & 3) << 4)) - 32
            base = 128 * half
            out[i, base + l] = d * s[isub + 0] * q1
            out[i, base + 32 + l] = d * s[isub + 2] * q2
            out[i, base + 64 + l] = d * s[isub + 4] * q3
            out[i, base + 96 + l] = d * s[isub + 6] * q4
    return out


def _blocks(rng, kind, nblk):
    if kind == "q5_0":
        raw = rng.integers(0, 256, size=(nblk, 22), dtype=np.uint8)
        raw[:, 0:2] = _f16_bits(rng, nblk).view(np.uint8).reshape(nblk, 2)
        raw[0, 2:6] = 0; raw[1, 2:6] = 0xFF          # all high bits clear / set
        raw[2, 6:22] = 0; raw[3, 6:22] = 0xFF
    elif kind == "q4_k":
        raw = rng.integers(0, 256, size=(nblk, 144), dtype=np.uint8)
        raw[:, 0:2] = _f16_bits(rng, nblk).view(np.uint8).reshape(nblk, 2)
        raw[:, 2:4] = _f16_bits(rng, nblk).view(np.uint8).reshape(nblk, 2)
        raw[0, 4:16] = 0xFF; raw[1, 4:16] = 0         # extreme 6-bit scales / mins
    else:
        raw = rng.integers(0, 256, size=(nblk, 210), dtype=np.uint8)
        raw[:, 208:210] = _f16_bits(rng, nblk).view(np.uint8).reshape(nblk, 2)
        raw[0, 192:208] = 0x80; raw[1, 192:208] = 0x7F  # int8 scales -128 / 127
        raw[2, 128:192] = 0xFF; raw[3, 0:128] = 0
    return np.ascontiguousarray(raw).reshape(-1)


@pytest.mark.parametrize("kind,gtype,per,npfn", [("q5_0", GGML_Q5_0, 32, np_q5_0), ("q4_k", GGML_Q4_K, 256, np_q4_k),
                                                  ("q6_k", GGML_Q6_K, 256, np_q6_k)])
def test_oracle_block_decoder_matches_published_format(kind, gtype, per, npfn):
    rng = np.random.Generator(np.random.PCG64(7))
    nblk = 24
    raw = _blocks(rng, kind, nblk)
    got = orc.dequant(raw, gtype, nblk * per).reshape(nblk, per)
    want = npfn(raw, nblk)
    np.testing.assert_allclose(got, want, rtol=3e-7, atol=1e-9)
    # the decoders use every bit of the block: flipping any single byte changes the output
    for pos in rng.integers(0, raw.size, size=16):
        flipped = raw.copy()
        flipped[pos] ^= 0x5A
        assert not np.array_equal(orc.dequant(flipped, gtype, nblk * per), got.reshape(-1))


def _kat_matrix(kind):
    """rows x 512 matrix from the hand-built super block (tests/kquant_kat.py): row 0 = [B, B x 4], row 1 = [B x 4, B]
    (x 4: the same bytes with d -- and dmin -- four times larger).  Returns (raw bytes, expected[2][512])."""
    import kquant_kat as kat
    raw, exp = kat.q4_k_block() if kind == "q4_k" else kat.q6_k_block()
    raw = bytearray(raw)
    big = bytearray(raw)
    import struct
    if kind == "q4_k":
        big[0:2] = struct.pack("<e", 2.0); big[2:4] = struct.pack("<e", 1.0)
    else:
        big[208:210] = struct.pack("<e", 1.0)
    e1, e4 = np.array(exp, np.float32), np.array(exp, np.float32) * np.float32(4)
    return (np.frombuffer(bytes(raw + big + big + raw), np.uint8).copy(),
            np.stack([np.concatenate([e1, e4]), np.concatenate([e4, e1])]))


@pytest.mark.parametrize("kind,gtype", [("q4_k", GGML_Q4_K), ("q6_k", GGML_Q6_K)])
def test_oracle_reproduces_the_hand_built_super_blocks_exactly(kind, gtype):
    # first-principles KAT (no reference vectors exist for these formats): every expected value is an exact binary
    # fraction, so dequantisation AND any-order dot products with small-integer x must match bit for bit
    raw, want = _kat_matrix(kind)
    got = orc.dequant(raw, gtype, 2 * 512).reshape(2, 512)
    assert np.array_equal(got, want)
    x = ((np.arange(512) * 7) % 5 - 2).astype(np.float32)
    assert np.array_equal(orc.matmul(raw, gtype, x, 2, 512), (want.astype(np.float64) @ x.astype(np.float64)).astype(np.float32))
