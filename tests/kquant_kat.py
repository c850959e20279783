"""Hand-built Q4_K and Q6_K super-blocks with exactly representable expected values (first principles: the
published ggml block formats the Go decoders restate, go/quant.go:171-204 Q6_K, :285-330 Q4_K; pure Python
integers and binary fractions, no numpy decoding, no oracle).

Every scale is a power-of-two fraction and every quant a small integer, so each dequantised element -- and any sum
of them in ANY order -- is exact in float32: the oracle and the device must reproduce these numbers bit for bit.
"""
import struct


def _h(v):
    return struct.pack("<e", v)


def q4_k_block():
    """One 144-byte Q4_K super block: d = 0.5, dmin = 0.25; sub-block scales s = 1,2,3,4, 21,6,7,8 and minimums
    m = 0,1,2,3, 4,35,6,7 (s[4] = 21 and m[5] = 35 need the two high bits that live in bytes 0..7 of the scale
    field); quants: byte b of 32-byte group g = low nibble (g + b) % 16, high nibble (15 - b) % 16.
    Returns (raw bytes, expected[256])."""
    d, dmin = 0.5, 0.25
    s = [1, 2, 3, 4, 21, 6, 7, 8]
    m = [0, 1, 2, 3, 4, 35, 6, 7]
    sc = [0] * 12
    for j in range(4):
        sc[j] = s[j] | ((s[j + 4] >> 4) << 6)          # low 6 bits: s[j]; top 2 bits: high bits of s[j + 4]
        sc[j + 4] = m[j] | ((m[j + 4] >> 4) << 6)      # low 6 bits: m[j]; top 2 bits: high bits of m[j + 4]
        sc[j + 8] = (s[j + 4] & 0xF) | ((m[j + 4] & 0xF) << 4)
    qs, exp = [], [0.0] * 256
    for g in range(4):
        for b in range(32):
            lo, hi = (g + b) % 16, (15 - b) % 16
            qs.append(lo | (hi << 4))
            exp[64 * g + b] = d * s[2 * g] * lo - dmin * m[2 * g]
            exp[64 * g + 32 + b] = d * s[2 * g + 1] * hi - dmin * m[2 * g + 1]
    return _h(d) + _h(dmin) + bytes(sc) + bytes(qs), exp


def q6_k_block():
    """One 210-byte Q6_K super block: d = 0.25, int8 sub-scales 1,-2,3,-4,5,-6,7,-8,9,10,-11,12,13,-14,15,-16;
    ql[i] = (7 i + 3) mod 256, qh[i] = (13 i + 5) mod 256.  Returns (raw bytes, expected[256])."""
    d = 0.25
    scales = [1, -2, 3, -4, 5, -6, 7, -8, 9, 10, -11, 12, 13, -14, 15, -16]
    ql = [(7 * i + 3) % 256 for i in range(128)]
    qh = [(13 * i + 5) % 256 for i in range(64)]
    exp = [0.0] * 256
    for half in range(2):
        for l in range(32):
            lo0, lo1, hb = ql[64 * half + l], ql[64 * half + l + 32], qh[32 * half + l]
            q = [(lo0 & 0xF) | (((hb >> 0) & 3) << 4), (lo1 & 0xF) | (((hb >> 2) & 3) << 4),
                 (lo0 >> 4) | (((hb >> 4) & 3) << 4), (lo1 >> 4) | (((hb >> 6) & 3) << 4)]
            for k in range(4):
                exp[128 * half + 32 * k + l] = d * scales[8 * half + l // 16 + 2 * k] * (q[k] - 32)
    return bytes(ql) + bytes(qh) + bytes(v & 0xFF for v in scales) + _h(d), exp
