// nl_kernels.h -- gfx950 device kernels of the nanollama decode path.
//
// Everything here is HBM/L2-streaming VALU work (decode GEMV over block-quantised
// weights, single-token GQA attention); there is deliberately no MFMA.
//
// Weight layout ("row-interleaved tiles", built once at upload by repack_kernel):
//   a tile  = TR(16) consecutive output rows;
//   a pair  = two 32-element quant blocks (64 columns);
//   a group = KL(4) consecutive pairs (256 columns); the last group of a row may hold 1-3 pairs.
//   For tile t, group g (gsz pairs), 16-byte chunk c of the pair, row-in-tile r, pair-in-group k:
//       quants : t*npairs*CPP*TR + g*KL*CPP*TR + (c*TR + r)*gsz + k     (x 16 bytes)
//       scales : t*npairs*TR     + g*KL*TR     +  r*gsz + k             (x 4 bytes: two fp16 d)
//   CPP = chunks per pair = 4 (Q8_0), 2 (Q4_0), 8 (F16), 16 (F32).
//   A wavefront owns one tile and walks groups; lane = r*4 + k.  Every global load instruction of
//   a wavefront is therefore ONE contiguous 1 KiB run, each lane accumulates whole blocks of its
//   own row (dot over 32 in-register, then * d, as go/quant.go:149-165 / :74-94 do per block), the
//   four partial sums of a row sit in one quad (two DPP adds), and no padding bytes are stored.
//   The fp16 scale bits are kept verbatim.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <type_traits>

namespace nl {

constexpr int TR = 16;           // rows per tile
constexpr int KL = 4;            // k-lanes per wavefront (64 / TR)
constexpr int PAIR = 64;         // columns per pair
constexpr int ATT_CH = 128;      // positions per attention split
constexpr int ATT_THREADS = 256;
constexpr int CTL_TOKEN = 0, CTL_POS = 1, CTL_CHAIN = 2, CTL_STEP = 3, CTL_STREAM = 4, CTL_HOSTOUT = 5, CTL_WORDS = 8;

// A word written by an EARLIER launch (ctl / epoch counters) read through the scalar cache: s_load_dword, counted on
// lgkmcnt.  A vector load of a wave-uniform value is followed by v_readfirstlane, i.e. by an immediate
// `s_waitcnt vmcnt(0)` -- a whole memory round trip in front of every load the kernel issues after it.  (The scalar
// cache is invalidated at every dispatch, like the vector L1.)
__device__ __forceinline__ int sload_i32(const int *p) {
    return *reinterpret_cast<const __attribute__((address_space(4))) int *>(reinterpret_cast<unsigned long long>(p));
}
// All kernel arguments a launch reads, requested as ONE batch of s_load: hipcc otherwise fetches each field right before
// its first use with a wait behind it -- four or five dependent scalar-cache round trips before the first weight load.
// "Defined, value irrelevant": registers that only a conditional load fills and whose unfilled use is masked later.  An
// explicit zero costs a v_mov per register per wavefront (81 of them in front of the first load of the attention block).
__device__ __forceinline__ void undef_regs(float4 &v) { asm volatile("" : "=v"(v.x), "=v"(v.y), "=v"(v.z), "=v"(v.w)); }
__device__ __forceinline__ void undef_regs(uint4 &v) { asm volatile("" : "=v"(v.x), "=v"(v.y), "=v"(v.z), "=v"(v.w)); }
__device__ __forceinline__ void undef_regs(uint2 &v) { asm volatile("" : "=v"(v.x), "=v"(v.y)); }
// load at "uniform base + unsigned 32-bit byte offset": the base stays in SGPRs and the load takes the saddr form (a signed
// index costs a 64-bit VALU add per load)
template <typename T>
__device__ __forceinline__ T ld_off(const void *base, unsigned byte_off) {
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off);
}
// n / d for a runtime divisor known on the host: inv = ceil(2^32 / d) (udiv_inv), q = mulhi(n, inv), exact while
// n < 2^32 / d.  hipcc's generic 32-bit division is ~35 instructions with a v_rcp in a dependent chain, in front of the
// first load of a launch that derives its tile from a wavefront or workgroup index.
__host__ __device__ inline unsigned udiv_inv(unsigned d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + d - 1) / d); }
__device__ __forceinline__ unsigned udiv_by(unsigned n, unsigned d, unsigned inv) { return d <= 1 ? n : __umulhi(n, inv); }
#define NL_KARGS2(a, b) asm volatile("" :: "s"(a), "s"(b))
#define NL_KARGS4(a, b, c, d) asm volatile("" :: "s"(a), "s"(b), "s"(c), "s"(d))
#define NL_KARGS8(a, b, c, d, e, f, g, h) asm volatile("" :: "s"(a), "s"(b), "s"(c), "s"(d), "s"(e), "s"(f), "s"(g), "s"(h))


// ggml ids (go/gguf.go:43-57).  Device layouts: Q5_0 is expanded to the Q8_0 layout at upload (5-bit value - 16 as
// int8, same fp16 d); Q6_K to WT_Q6_K = int8 (6-bit value - 32) with four int8 sub-scales per 64 columns;
// Q4_K keeps its nibbles with an 8-byte {d, dmin, sc0, m0, sc1, m1} entry per 64 columns.
enum { WT_F32 = 0, WT_F16 = 1, WT_Q4_0 = 2, WT_Q5_0 = 6, WT_Q8_0 = 8, WT_Q4_K = 12, WT_Q6_K = 14 };
enum { PRO_PLAIN = 0, PRO_NORM = 1, PRO_ATTN = 2, PRO_NORM_PARTS = 3 };
constexpr int MAX_PARTS = 12;     // PRO_NORM_PARTS: partial vectors added to x (one per attention head, nl_block.h)
enum { EPI_STORE = 0, EPI_RESID = 1, EPI_SWIGLU = 2, EPI_QKV = 3, EPI_P2P = 4 };
enum { ROWMAP_IDENT = 0, ROWMAP_HEADPERM = 1 };

template <int WT> struct WTraits;
template <> struct WTraits<WT_Q8_0> { static constexpr int CPP = 4; static constexpr bool SCALED = true; };
template <> struct WTraits<WT_Q4_0> { static constexpr int CPP = 2; static constexpr bool SCALED = true; };
template <> struct WTraits<WT_F16> { static constexpr int CPP = 8; static constexpr bool SCALED = false; };
template <> struct WTraits<WT_F32> { static constexpr int CPP = 16; static constexpr bool SCALED = false; };
template <> struct WTraits<WT_Q4_K> { static constexpr int CPP = 2; static constexpr bool SCALED = true; };
template <> struct WTraits<WT_Q6_K> { static constexpr int CPP = 4; static constexpr bool SCALED = true; };
__host__ __device__ constexpr int scale_words(int wt) { return (wt == WT_Q4_K || wt == WT_Q6_K) ? 2 : 1; }

__device__ __forceinline__ float h2f_bits(uint32_t h) {
    // exact IEEE binary16 -> binary32 (subnormals kept), == go/gguf.go:603-636
    return __half2float(__ushort_as_half((unsigned short)h));
}

// float32(exp(float64(x))) -- the exponential of the reference's Softmax and SiLU (go/quant.go:619, :629-631) -- with a
// short dependent chain.  The library exp is ~150 instructions, most of them dependent; in the fused block launches one
// wavefront computes the softmax of a head while eleven wait for it, so its chain is launch time.  Cody-Waite
// reduction (k = rint(x log2 e), r = x - k ln2 in two exact steps), degree-13 Taylor polynomial in Estrin form (|r| <=
// 0.347: truncation 4e-18), v_ldexp_f64.  <= 2 ulp in float64, i.e. the float32 result differs from the library's only
// when the exact value lies within 2^-51 of a float32 rounding boundary (never in 4e6 test arguments,
// tests/test_gpu_parity.py::test_fast_exp_matches_float64_exp).  NaN in, NaN out; -inf and x < -745 give 0, x > 709 inf.
__device__ __forceinline__ float exp_f64_as_f32(float xf) {
    const double x = fmin(fmax((double)xf, -750.0), 710.0);
    const double k = rint(x * 1.44269504088896338700e+00);
    double r = fma(k, -6.93147180369123816490e-01, x);
    r = fma(k, -1.90821492927058770002e-10, r);
    const double r2 = r * r, r4 = r2 * r2, r8 = r4 * r4;
    const double a0 = 1.0 + r;
    const double a1 = fma(r, 1.0 / 6, 1.0 / 2), a2 = fma(r, 1.0 / 120, 1.0 / 24), a3 = fma(r, 1.0 / 5040, 1.0 / 720);
    const double a4 = fma(r, 1.0 / 362880, 1.0 / 40320), a5 = fma(r, 1.0 / 39916800, 1.0 / 3628800);
    const double a6 = fma(r, 1.0 / 6227020800.0, 1.0 / 479001600);
    const double b0 = fma(a1, r2, a0), b1 = fma(a3, r2, a2), b2 = fma(a5, r2, a4);
    const double d0 = fma(b1, r4, b0), d1 = fma(a6, r4, b2);
    const double p = fma(d1, r8, d0);
    const float res = (float)ldexp(p, (int)k);
    return xf == xf ? res : xf;
}

// ---------------------------------------------------------------- repack ---

struct RepackParams {
    const uint8_t *src;   // raw GGUF tensor bytes on device (whole tensor)
    uint8_t *q;           // destination chunk plane
    uint32_t *s;          // destination scale plane (may be null)
    int wtype;
    int src_cols;         // columns of the source tensor
    int row0, nrows;      // row slice taken
    int col0, ncols;      // column slice taken (multiples of 32)
    int ntiles, npairs;
    int rowmap, head_dim;
};

__device__ __forceinline__ int map_row(int rowmap, int head_dim, int tile, int r) {
    if (rowmap == ROWMAP_HEADPERM) {
        // RoPE partners (i, i + hd/2) share a tile: rows 0-7 hold i, rows 8-15 hold i + hd/2
        int tph = head_dim / 16;
        int head = tile / tph, j = tile % tph;
        return head * head_dim + j * 8 + (r & 7) + (r >> 3) * (head_dim / 2);
    }
    return tile * TR + r;
}

// go/quant.go:285-294
__device__ __forceinline__ void scale_min_k4(int j, const uint8_t *sc, uint32_t &s, uint32_t &m) {
    if (j < 4) { s = sc[j] & 63; m = sc[j + 4] & 63; }
    else { s = (sc[j + 4] & 0x0F) | ((sc[j - 4] >> 6) << 4); m = (sc[j + 4] >> 4) | ((sc[j] >> 6) << 4); }
}
// 6-bit value of element e (0..255) of a Q6_K super block (go/quant.go:193-203) and its scale index
__device__ __forceinline__ int q6k_value(const uint8_t *blk, int e, int &sidx) {
    const int n = e >> 7, within = e & 127, which = within >> 5, l = within & 31;
    const uint8_t *ql = blk + n * 64, *qh = blk + 128 + n * 32;
    sidx = n * 8 + (l >> 4) + 2 * which;
    const int lo = (which & 1) ? ql[l + 32] : ql[l];
    const int nib = (which >> 1) ? (lo >> 4) : (lo & 0x0F);
    return nib | (((qh[l] >> (2 * which)) & 3) << 4);
}

__global__ void repack_kernel(RepackParams P) {
    // P.wtype is the SOURCE ggml type; the destination layout is Q8_0-like for Q5_0 / Q6_K (see the enum)
    const int src = P.wtype;
    const int cpp = (src == WT_Q8_0 || src == WT_Q5_0 || src == WT_Q6_K) ? 4 : (src == WT_Q4_0 || src == WT_Q4_K) ? 2
                    : src == WT_F16 ? 8 : 16;
    const int sw = scale_words(src);
    const int per_tile = P.npairs * cpp * TR;
    const long long nchunks = (long long)P.ntiles * per_tile;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < nchunks;
         idx += (long long)gridDim.x * blockDim.x) {
        const int tile = (int)(idx / per_tile);
        int rem = (int)(idx - (long long)tile * per_tile);
        const int g = rem / (KL * cpp * TR);
        rem -= g * (KL * cpp * TR);
        const int gsz = min(KL, P.npairs - g * KL);
        const int k = rem % gsz;
        rem /= gsz;
        const int r = rem % TR, c = rem / TR;
        const int p = g * KL + k;
        int row = map_row(P.rowmap, P.head_dim, tile, r);
        uint4 v = make_uint4(0, 0, 0, 0);
        uint8_t *vb = reinterpret_cast<uint8_t *>(&v);
        uint32_t sc0 = 0, sc1 = 0;
        const bool in_row = row < P.nrows;
        const long long srow = (long long)(P.row0 + (in_row ? row : 0));
        if (in_row) {
            if (src == WT_Q8_0) {
                int b = 2 * p + (c >> 1);
                if (b * 32 < P.ncols) {
                    const uint8_t *blk = P.src + (srow * (P.src_cols / 32) + (P.col0 / 32 + b)) * 34;
                    for (int j = 0; j < 16; j++) vb[j] = blk[2 + (c & 1) * 16 + j];
                }
            } else if (src == WT_Q4_0) {
                int b = 2 * p + c;
                if (b * 32 < P.ncols) {
                    const uint8_t *blk = P.src + (srow * (P.src_cols / 32) + (P.col0 / 32 + b)) * 18;
                    for (int j = 0; j < 16; j++) vb[j] = blk[2 + j];
                }
            } else if (src == WT_Q5_0) {
                // go/quant.go:405-420: 5-bit q = nibble | high bit << 4, value (q - 16) * d
                int b = 2 * p + (c >> 1);
                if (b * 32 < P.ncols) {
                    const uint8_t *blk = P.src + (srow * (P.src_cols / 32) + (P.col0 / 32 + b)) * 22;
                    const uint32_t qh = (uint32_t)blk[2] | ((uint32_t)blk[3] << 8) | ((uint32_t)blk[4] << 16) | ((uint32_t)blk[5] << 24);
                    for (int j = 0; j < 16; j++) {
                        int e = (c & 1) * 16 + j;
                        int nib = e < 16 ? (blk[6 + e] & 0x0F) : (blk[6 + e - 16] >> 4);
                        vb[j] = (uint8_t)(int8_t)((nib | (int)(((qh >> e) & 1u) << 4)) - 16);
                    }
                }
            } else if (src == WT_Q4_K) {
                // super block = 4 pairs; pair j holds qs[32j .. 32j+31] (low nibbles = first 32 columns, high = next 32)
                if (p * PAIR < P.ncols) {
                    const uint8_t *blk = P.src + (srow * (P.src_cols / 256) + (P.col0 / 256 + (p >> 2))) * 144;
                    for (int j = 0; j < 16; j++) vb[j] = blk[16 + 32 * (p & 3) + 16 * c + j];
                }
            } else if (src == WT_Q6_K) {
                if (p * PAIR < P.ncols) {
                    const uint8_t *blk = P.src + (srow * (P.src_cols / 256) + (P.col0 / 256 + (p >> 2))) * 210;
                    for (int j = 0; j < 16; j++) {
                        int sidx;
                        vb[j] = (uint8_t)(int8_t)(q6k_value(blk, 64 * (p & 3) + 16 * c + j, sidx) - 32);
                    }
                }
            } else {
                int esz = src == WT_F16 ? 2 : 4;
                int per = 16 / esz;
                int e0 = p * PAIR + c * per;
                const uint8_t *sp = P.src + (srow * P.src_cols + P.col0 + e0) * esz;
                for (int j = 0; j < per; j++)
                    if (e0 + j < P.ncols)
                        for (int bb = 0; bb < esz; bb++) vb[j * esz + bb] = sp[j * esz + bb];
            }
        }
        reinterpret_cast<uint4 *>(P.q)[idx] = v;
        if (P.s && c == 0) {
            if (in_row) {
                if (src == WT_Q8_0 || src == WT_Q4_0 || src == WT_Q5_0) {
                    const int bsz = src == WT_Q8_0 ? 34 : src == WT_Q4_0 ? 18 : 22;
                    for (int hb = 0; hb < 2; hb++) {
                        int b = 2 * p + hb;
                        if (b * 32 < P.ncols) {
                            const uint8_t *blk = P.src + (srow * (P.src_cols / 32) + (P.col0 / 32 + b)) * bsz;
                            sc0 |= ((uint32_t)blk[0] | ((uint32_t)blk[1] << 8)) << (16 * hb);
                        }
                    }
                } else if (src == WT_Q4_K && p * PAIR < P.ncols) {
                    const uint8_t *blk = P.src + (srow * (P.src_cols / 256) + (P.col0 / 256 + (p >> 2))) * 144;
                    uint32_t s0, m0, s1, m1;
                    scale_min_k4(2 * (p & 3), blk + 4, s0, m0);
                    scale_min_k4(2 * (p & 3) + 1, blk + 4, s1, m1);
                    sc0 = (uint32_t)blk[0] | ((uint32_t)blk[1] << 8) | ((uint32_t)blk[2] << 16) | ((uint32_t)blk[3] << 24);  // d | dmin
                    sc1 = s0 | (m0 << 8) | (s1 << 16) | (m1 << 24);
                } else if (src == WT_Q6_K && p * PAIR < P.ncols) {
                    const uint8_t *blk = P.src + (srow * (P.src_cols / 256) + (P.col0 / 256 + (p >> 2))) * 210;
                    sc0 = (uint32_t)blk[208] | ((uint32_t)blk[209] << 8);
                    for (int i = 0; i < 4; i++) {
                        int sidx;
                        q6k_value(blk, 64 * (p & 3) + 16 * i, sidx);
                        sc1 |= (uint32_t)blk[192 + sidx] << (8 * i);
                    }
                }
            }
            const long long si = (long long)tile * P.npairs * TR + g * KL * TR + r * gsz + k;
            if (sw == 1) P.s[si] = sc0;
            else { P.s[2 * si] = sc0; P.s[2 * si + 1] = sc1; }
        }
    }
}

// ------------------------------------------------------------- pair dots ---

// Four independent accumulators per 32-element block: a lone wavefront on a SIMD (small models keep
// the chip nearly empty) is bound by the dependent-FMA chain, not by issue rate.
struct Acc4 { float a, b, c, d; };

__device__ __forceinline__ void dot4_i8(uint32_t w, float4 x, Acc4 &s) {
    s.a = fmaf((float)(int)(int8_t)(w & 0xff), x.x, s.a);
    s.b = fmaf((float)(int)(int8_t)((w >> 8) & 0xff), x.y, s.b);
    s.c = fmaf((float)(int)(int8_t)((w >> 16) & 0xff), x.z, s.c);
    s.d = fmaf((float)(int)(int8_t)(w >> 24), x.w, s.d);
}

template <int WT> struct PairDot;

template <> struct PairDot<WT_Q8_0> {
    // out = sum_b d_b * sum_j q_bj x_j  (go/quant.go:149-165)
    static __device__ __forceinline__ float run(const uint4 *c, uint2 sc2, const float *xp, float acc) {
        const uint32_t sc = sc2.x;
        const float4 *x4 = reinterpret_cast<const float4 *>(xp);
#pragma unroll
        for (int b = 0; b < 2; b++) {
            Acc4 s{0.f, 0.f, 0.f, 0.f};
            uint4 lo = c[2 * b], hi = c[2 * b + 1];
            dot4_i8(lo.x, x4[8 * b + 0], s);
            dot4_i8(lo.y, x4[8 * b + 1], s);
            dot4_i8(lo.z, x4[8 * b + 2], s);
            dot4_i8(lo.w, x4[8 * b + 3], s);
            dot4_i8(hi.x, x4[8 * b + 4], s);
            dot4_i8(hi.y, x4[8 * b + 5], s);
            dot4_i8(hi.z, x4[8 * b + 6], s);
            dot4_i8(hi.w, x4[8 * b + 7], s);
            acc = fmaf((s.a + s.b) + (s.c + s.d), h2f_bits((sc >> (16 * b)) & 0xffff), acc);
        }
        return acc;
    }
};

typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ h2_t bits_h2(uint32_t u) { return __builtin_bit_cast(h2_t, u); }

// (a & mask) | c in one VALU op (hipcc splits it because VOP3 takes no 32-bit literal: mask rides in an SGPR)
__device__ __forceinline__ uint32_t and_or_b32(uint32_t a, uint32_t mask, uint32_t c) {
    uint32_t r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(mask), "v"(c));
    return r;
}
// acc + float(half) * x with the fp16 operand read straight from the low / high half of a packed register
__device__ __forceinline__ float fma_mix_lo(uint32_t h2, float x, float acc) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h2), "v"(x), "v"(acc));
    return r;
}
__device__ __forceinline__ float fma_mix_hi(uint32_t h2, float x, float acc) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h2), "v"(x), "v"(acc));
    return r;
}
__device__ __forceinline__ uint32_t h2_bits(h2_t v) { return __builtin_bit_cast(uint32_t, v); }

__device__ __forceinline__ void dot_q4_word(uint32_t w, float4 xl, float4 xh, Acc4 &s) {
    // byte k of w: low nibble = element k, high nibble = element k+16 (go/quant.go:84-88).
    // Nibble -> (n - 8) without integer unpacking: OR the nibble into the mantissa of fp16 1024.0
    // (0x6400): bits 0-3 give 1024+n, bits 4-7 give 1024+16n; one packed fp16 op then yields n-8
    // EXACTLY for two nibbles at once, and v_fma_mix_f32 multiplies it into the f32 accumulator:
    // 17 VALU ops per 8 weights.
    const h2_t k1032 = {(_Float16)1032.0f, (_Float16)1032.0f};
    const h2_t k16th = {(_Float16)0.0625f, (_Float16)0.0625f};
    const h2_t km72 = {(_Float16)-72.0f, (_Float16)-72.0f};
    const uint32_t magic = 0x64006400u;
    const uint32_t w8 = w >> 8;
    uint32_t e02 = h2_bits(bits_h2(and_or_b32(w, 0x000F000Fu, magic)) - k1032);                                    // elems 0, 2
    uint32_t e13 = h2_bits(bits_h2(and_or_b32(w8, 0x000F000Fu, magic)) - k1032);                                   // elems 1, 3
    uint32_t f02 = h2_bits(__builtin_elementwise_fma(bits_h2(and_or_b32(w, 0x00F000F0u, magic)), k16th, km72));    // elems 16, 18
    uint32_t f13 = h2_bits(__builtin_elementwise_fma(bits_h2(and_or_b32(w8, 0x00F000F0u, magic)), k16th, km72));   // elems 17, 19
    s.a = fma_mix_lo(e02, xl.x, s.a);
    s.b = fma_mix_lo(f02, xh.x, s.b);
    s.c = fma_mix_lo(e13, xl.y, s.c);
    s.d = fma_mix_lo(f13, xh.y, s.d);
    s.a = fma_mix_hi(e02, xl.z, s.a);
    s.b = fma_mix_hi(f02, xh.z, s.b);
    s.c = fma_mix_hi(e13, xl.w, s.c);
    s.d = fma_mix_hi(f13, xh.w, s.d);
}

template <> struct PairDot<WT_Q4_0> {
    static __device__ __forceinline__ float run(const uint4 *c, uint2 sc2, const float *xp, float acc) {
        const uint32_t sc = sc2.x;
        const float4 *x4 = reinterpret_cast<const float4 *>(xp);
#pragma unroll
        for (int b = 0; b < 2; b++) {
            Acc4 s{0.f, 0.f, 0.f, 0.f};
            uint4 q = c[b];
            dot_q4_word(q.x, x4[8 * b + 0], x4[8 * b + 4], s);
#ifndef NL_FAKE_DOT   // developer experiment (tools/collect_r05.sh fakedot): a quarter of the vector work, wrong results -- how much of a step is it?
            dot_q4_word(q.y, x4[8 * b + 1], x4[8 * b + 5], s);
            dot_q4_word(q.z, x4[8 * b + 2], x4[8 * b + 6], s);
            dot_q4_word(q.w, x4[8 * b + 3], x4[8 * b + 7], s);
#else
            s.b += __uint_as_float(q.y & 1u) + __uint_as_float(q.z & 1u) + __uint_as_float(q.w & 1u);      // (the loads stay)
#endif
            acc = fmaf((s.a + s.b) + (s.c + s.d), h2f_bits((sc >> (16 * b)) & 0xffff), acc);
        }
        return acc;
    }
};

// One 32-element block of a pair (the fused block kernels split a pair's two blocks over two lanes): chunks c[0 .. CPP/2),
// this block's fp16 scale, the block's 32 inputs.  Same arithmetic per block as PairDot.
template <int WT> struct BlockDot;
template <> struct BlockDot<WT_Q8_0> {
    static __device__ __forceinline__ float run(const uint4 *c, uint32_t d16, const float *xp) {
        const float4 *x4 = reinterpret_cast<const float4 *>(xp);
        Acc4 s{0.f, 0.f, 0.f, 0.f};
        const uint4 lo = c[0], hi = c[1];
        dot4_i8(lo.x, x4[0], s); dot4_i8(lo.y, x4[1], s); dot4_i8(lo.z, x4[2], s); dot4_i8(lo.w, x4[3], s);
        dot4_i8(hi.x, x4[4], s); dot4_i8(hi.y, x4[5], s); dot4_i8(hi.z, x4[6], s); dot4_i8(hi.w, x4[7], s);
        return ((s.a + s.b) + (s.c + s.d)) * h2f_bits(d16);
    }
};
template <> struct BlockDot<WT_Q4_0> {
    static __device__ __forceinline__ float run(const uint4 *c, uint32_t d16, const float *xp) {
        const float4 *x4 = reinterpret_cast<const float4 *>(xp);
        Acc4 s{0.f, 0.f, 0.f, 0.f};
        const uint4 q = c[0];
        dot_q4_word(q.x, x4[0], x4[4], s); dot_q4_word(q.y, x4[1], x4[5], s);
        dot_q4_word(q.z, x4[2], x4[6], s); dot_q4_word(q.w, x4[3], x4[7], s);
        return ((s.a + s.b) + (s.c + s.d)) * h2f_bits(d16);
    }
};
// chunks and scale of block `blk` (0 / 1) of pair k in a tile: the block-level counterpart of load_pair
template <int WT>
__device__ __forceinline__ void load_block(const uint8_t *q, const uint32_t *s, long long tile_pair0, int g, int gsz, int r, int k,
                                           int blk, uint4 *c, uint32_t &d16) {
    constexpr int CPP = WTraits<WT>::CPP, CPB = CPP / 2;
    const uint4 *qb = reinterpret_cast<const uint4 *>(q) + (tile_pair0 * (CPP * TR) + (long long)g * (KL * CPP * TR));
    const unsigned off = __umul24((unsigned)r, (unsigned)gsz) + (unsigned)k;
#pragma unroll
    for (int j = 0; j < CPB; j++) c[j] = (qb + (blk * CPB + j) * TR * gsz)[off];
    const long long sb = tile_pair0 * TR + (long long)g * (KL * TR);
    d16 = ((s + sb)[off] >> (16 * blk)) & 0xffffu;
}

template <> struct PairDot<WT_F16> {
    static __device__ __forceinline__ float run(const uint4 *c, uint2, const float *xp, float acc) {
        const float4 *x4 = reinterpret_cast<const float4 *>(xp);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            uint4 q = c[k];
            float4 a = x4[2 * k], b = x4[2 * k + 1];
            a0 = fmaf(h2f_bits(q.x & 0xffff), a.x, a0);
            a1 = fmaf(h2f_bits(q.x >> 16), a.y, a1);
            a2 = fmaf(h2f_bits(q.y & 0xffff), a.z, a2);
            a3 = fmaf(h2f_bits(q.y >> 16), a.w, a3);
            a0 = fmaf(h2f_bits(q.z & 0xffff), b.x, a0);
            a1 = fmaf(h2f_bits(q.z >> 16), b.y, a1);
            a2 = fmaf(h2f_bits(q.w & 0xffff), b.z, a2);
            a3 = fmaf(h2f_bits(q.w >> 16), b.w, a3);
        }
        return acc + ((a0 + a1) + (a2 + a3));
    }
};

template <> struct PairDot<WT_F32> {
    static __device__ __forceinline__ float run(const uint4 *c, uint2, const float *xp, float acc) {
        const float4 *x4 = reinterpret_cast<const float4 *>(xp);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            uint4 q = c[k];
            float4 a = x4[k];
            a0 = fmaf(__uint_as_float(q.x), a.x, a0);
            a1 = fmaf(__uint_as_float(q.y), a.y, a1);
            a2 = fmaf(__uint_as_float(q.z), a.z, a2);
            a3 = fmaf(__uint_as_float(q.w), a.w, a3);
        }
        return acc + ((a0 + a1) + (a2 + a3));
    }
};

template <> struct PairDot<WT_Q4_K> {
    // go/quant.go:364-396: out += (d*sc * q - dmin*m) * x per element; this pair = sub-blocks 2j, 2j+1 of a super block
    static __device__ __forceinline__ float run(const uint4 *c, uint2 sc, const float *xp, float acc) {
        const float d = h2f_bits(sc.x & 0xffff), dmin = h2f_bits(sc.x >> 16);
        const float d1 = d * (float)(sc.y & 0xff), m1 = dmin * (float)((sc.y >> 8) & 0xff);
        const float d2 = d * (float)((sc.y >> 16) & 0xff), m2 = dmin * (float)(sc.y >> 24);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int ch = 0; ch < 2; ch++) {
            const uint32_t w[4] = {c[ch].x, c[ch].y, c[ch].z, c[ch].w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float *xl = xp + 16 * ch + 4 * i, *xh = xp + 32 + 16 * ch + 4 * i;
                const uint32_t u = w[i];
                a0 = fmaf(d1 * (float)(u & 0xF) - m1, xl[0], a0);
                a1 = fmaf(d2 * (float)((u >> 4) & 0xF) - m2, xh[0], a1);
                a2 = fmaf(d1 * (float)((u >> 8) & 0xF) - m1, xl[1], a2);
                a3 = fmaf(d2 * (float)((u >> 12) & 0xF) - m2, xh[1], a3);
                a0 = fmaf(d1 * (float)((u >> 16) & 0xF) - m1, xl[2], a0);
                a1 = fmaf(d2 * (float)((u >> 20) & 0xF) - m2, xh[2], a1);
                a2 = fmaf(d1 * (float)((u >> 24) & 0xF) - m1, xl[3], a2);
                a3 = fmaf(d2 * (float)(u >> 28) - m2, xh[3], a3);
            }
        }
        return acc + ((a0 + a1) + (a2 + a3));
    }
};

template <> struct PairDot<WT_Q6_K> {
    // go/quant.go:239-276: out += d * sc_i * (q - 32) * x; four 16-column sub-blocks per pair, q - 32 stored as int8
    static __device__ __forceinline__ float run(const uint4 *c, uint2 sc, const float *xp, float acc) {
        const float4 *x4 = reinterpret_cast<const float4 *>(xp);
        const float d = h2f_bits(sc.x & 0xffff);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            Acc4 s{0.f, 0.f, 0.f, 0.f};
            dot4_i8(c[i].x, x4[4 * i + 0], s);
            dot4_i8(c[i].y, x4[4 * i + 1], s);
            dot4_i8(c[i].z, x4[4 * i + 2], s);
            dot4_i8(c[i].w, x4[4 * i + 3], s);
            const float si = d * (float)(int)(int8_t)((sc.y >> (8 * i)) & 0xff);
            acc = fmaf((s.a + s.b) + (s.c + s.d), si, acc);
        }
        return acc;
    }
};

// ------------------------------------------------------------------ GEMV ---

struct GemvParams {
    // weights (matrix 1 only for EPI_SWIGLU: gate = 0, up = 1)
    const uint8_t *q0, *q1;
    const uint32_t *s0, *s1;
    int rows, cols, npairs, ntiles, tw, kw;
    unsigned kw_inv, kw2_inv;   // udiv_inv(kw), udiv_inv(2 * kw): wavefront -> tile slot without a division (launch_gemv_t)
    // prologue: input vector
    const float *x;      // PRO_PLAIN / PRO_NORM input [cols]
    const float *add;    // optional addend (tensor-parallel: all-reduced partial of the previous block)
    const float *add_src;  // what the kernel actually loads: add, or x when there is no addend (set by launch_gemv_t)
    float *x_out;        // optional: block 0 writes x (+ add) here (the updated residual stream)
    const float *normw;  // PRO_NORM weights
    float eps;
    // PRO_ATTN: combine attention split partials into the input vector
    const float *part_o;   // [heads][nsplit_max][hd]
    const float *part_ml;  // [heads][nsplit_max][2]  (max, sum)
    int nsplit_max, head_dim;
    const int *ctl;
    // epilogue
    float *out;          // EPI_STORE / EPI_SWIGLU destination, EPI_RESID destination
    const float *resid;  // EPI_RESID: out[row] = resid[row] + v
    // EPI_QKV
    const float *rope_cos, *rope_sin;  // [seq][hd/2]
    float *qbuf;                       // [n_q_heads*hd]
    float *kcache, *vcache;            // this layer, stream 0: [kv][seq][hd]
    long long kv_stream_stride;        // floats between streams
    int n_q_heads, n_kv_heads, seq_len, rope_conj;
    // EPI_STORE: optional fused partial argmax (one slot per workgroup, or per wave when tw*16 > 64)
    float *amax_val;
    int *amax_idx;
    // optional attention biases (go/model.go:244-247,525-527,591): EPI_QKV adds bias_q/k/v[head*hd+e] before RoPE,
    // EPI_RESID / EPI_STORE add bias_out[row]
    const float *bias_q, *bias_k, *bias_v, *bias_out;
    long long *dbg;  // optional phase timestamps (clock64) written by workgroup 0, lane 0 of each wave
    // tensor-parallel push (nl_p2p.h).  EPI_P2P: this rank's partial of out[row] leaves as a tagged granule into the
    // receive slot of every rank (p2p_dst[r] = rank r's slot for THIS rank's partial); the SAME lane then waits for the
    // p2p_n granules the ranks pushed for its row (p2p_slots: this rank's receive slots of the seam, [G][rows]) and
    // stores out[row] = resid[row] + sum over ranks in rank order -- the all-reduce is finished by the row's owner
    // inside the producing launch, there is no reduce launch.  EPI_STORE with peer_out: the logits slice is also
    // written into every peer's gathered logits buffer.
    unsigned long long *p2p_dst[8];
    const unsigned long long *p2p_slots;
    const unsigned *p2p_epoch;
    unsigned *p2p_status;       // set non-zero when a poll gave up (host: NL_ERR_COMM)
    long long p2p_timeout;      // wall_clock64 ticks (100 MHz)
    int p2p_n;
    unsigned p2p_seam;
    float *peer_out[8];
    float *host_out;            // EPI_STORE, LM head on one GPU: device address of the pinned host logits buffer (used when ctl[CTL_HOSTOUT])
    // PRO_NORM_PARTS: x + sum_p parts[p][.] is the input (fixed order p = 0..nparts-1); block 0 stores it to x_out
    const float *parts;
    int nparts;
};

// ---- cross-lane helpers on DPP (hipcc lowers __shfl_xor to ds_bpermute: ~100+ cycles a hop) ----
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
constexpr int DPP_QUAD_XOR1 = 0xB1;   // quad_perm [1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;   // quad_perm [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141;
constexpr int DPP_ROW_MIRROR = 0x140;

// row_bcast15 / row_bcast31 (gfx9): lane 15 of each row of 16 into the next row / lane 31 into rows 2-3; rows outside the
// row mask receive 0.  With the four in-row steps they make a full-wave reduction without ds_bpermute (the result is in
// lane 63 and is broadcast through an SGPR).
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_bcast_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROWMASK, 0xF, false));
}
constexpr int DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;

// sum over the 4 lanes of each quad, result in every lane of the quad
__device__ __forceinline__ float quad_sum(float v) {
    v += dpp_f32<DPP_QUAD_XOR1>(v);
    v += dpp_f32<DPP_QUAD_XOR2>(v);
    return v;
}
// full-wave float64 sum, result valid in every lane
__device__ __forceinline__ double wave_sum_f64(double v) {
    v += dpp_f64<DPP_QUAD_XOR1>(v);
    v += dpp_f64<DPP_QUAD_XOR2>(v);
    v += dpp_f64<DPP_HALF_MIRROR>(v);
    v += dpp_f64<DPP_ROW_MIRROR>(v);   // every lane: sum of its row of 16
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

constexpr int XS_PAIR = PAIR + 4;            // LDS floats per pair (16-byte pad: the 4 pairs of a group hit distinct banks)
constexpr int XS_WAVE = KL * XS_PAIR;        // LDS floats per wavefront (one 256-column group)

// The 4 input-vector elements (columns col..col+3) this lane stages for its wavefront's current group.
// BRANCH-FREE on purpose: the caller passes an in-range column (clamped) and zeroes the result itself.  An early
// "if (col >= cols) return 0" makes every loaded value a phi, and hipcc then waits vmcnt(0) INSIDE the branch --
// i.e. the x / norm-weight round trip completed before the first weight load was even issued.
template <int PRO>
// Addresses are a wave-UNIFORM base (the group's first column, scalar arithmetic) plus an unsigned 32-bit lane
// offset: no 64-bit VALU address arithmetic, no sign extensions.
__device__ __forceinline__ float4 load_x4(const GemvParams &P, int gcol, unsigned lcol, float4 &g, float4 &addv, int ns) {
    g = make_float4(1.f, 1.f, 1.f, 1.f);
    addv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (PRO == PRO_ATTN) {
        // merge the position-split attention partials (online softmax): x = sum_c w_c o_c / sum_c w_c l_c.
        // Split 0 is fetched unconditionally; the loop over further splits only runs once pos >= 128.
        const int col = gcol + (int)lcol, hd = P.head_dim;
        const int h = (hd & (hd - 1)) == 0 ? col >> (__ffs(hd) - 1) : col / hd, d = col - h * hd;
        const float *ml = P.part_ml + (long long)h * P.nsplit_max * 2;
        const float *po = P.part_o + (long long)h * P.nsplit_max * hd + d;
        const float4 o0 = *reinterpret_cast<const float4 *>(po);
        const float m0 = ml[0], l0 = ml[1];
        if (ns == 1) {
            float il = 1.0f / l0;
            return make_float4(o0.x * il, o0.y * il, o0.z * il, o0.w * il);
        }
        // Long contexts (this plan serves positions >= 256 and the models the fused launches do not cover): the (max, sum)
        // pairs and partial rows of EIGHT splits are fetched per memory round trip (clamped split index, surplus ones get
        // weight 0 -- a runtime "for c < ns: load" loop is ns dependent round trips), and the merge weights
        // exp(m_c - M) use the f32 exponential: they are this engine's own construct (the reference has no splits), and
        // sixteen float64 exps per lane in every wavefront of the launch cost more than the GEMV itself at pos 2047.
        const float2 *ml2 = reinterpret_cast<const float2 *>(ml);
        float M = m0;
        for (int c0 = 0; c0 < ns; c0 += 8) {
            float2 t[8];
#pragma unroll
            for (int j = 0; j < 8; j++) t[j] = ml2[min(c0 + j, ns - 1)];
#pragma unroll
            for (int j = 0; j < 8; j++) M = fmaxf(M, t[j].x);
        }
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        float L = 0.f;
        for (int c0 = 0; c0 < ns; c0 += 8) {
            float2 t[8];
            float4 o[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int c = min(c0 + j, ns - 1);
                t[j] = ml2[c];
                o[j] = *reinterpret_cast<const float4 *>(po + (long long)c * hd);
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float w = c0 + j < ns ? __expf(t[j].x - M) : 0.f;
                L += w * t[j].y;
                v.x += w * o[j].x; v.y += w * o[j].y; v.z += w * o[j].z; v.w += w * o[j].w;
            }
        }
        float il = 1.0f / L;
        return make_float4(v.x * il, v.y * il, v.z * il, v.w * il);
    }
    float4 v = *reinterpret_cast<const float4 *>((P.x + gcol) + lcol);
    if (PRO == PRO_NORM || PRO == PRO_NORM_PARTS) g = *reinterpret_cast<const float4 *>((P.normw + gcol) + lcol);
    // optional addend (gamma row / tensor-parallel residual): always LOADED (from x itself when absent: same
    // cache line, the launcher resolves
    // the pointer so there is no select here for hipcc to turn back into a branch) and handed back; the caller adds it when it consumes x, after every load of the round is out
    addv = *reinterpret_cast<const float4 *>((P.add_src + gcol) + lcol);
    return v;
}

template <int WT>
__device__ __forceinline__ void load_pair(const uint8_t *q, const uint32_t *s, long long tile_pair0, int g, int gsz,
                                          int r, int k, uint4 *c, uint2 &sc) {
    constexpr int CPP = WTraits<WT>::CPP;
    // uniform bases (tile, group) + one unsigned lane offset shared by the quant chunks and the scale entry
    const uint4 *qb = reinterpret_cast<const uint4 *>(q) + (tile_pair0 * (CPP * TR) + (long long)g * (KL * CPP * TR));
    const unsigned off = __umul24((unsigned)r, (unsigned)gsz) + (unsigned)k;
#ifdef NL_NT_WEIGHTS   // developer build (tools/r4_big.sh): non-temporal policy on the streamed weight chunks
#pragma unroll
    for (int j = 0; j < CPP; j++) {
        typedef unsigned u32x4n __attribute__((ext_vector_type(4)));
        const u32x4n t = __builtin_nontemporal_load(reinterpret_cast<const u32x4n *>(qb + j * TR * gsz) + off);
        c[j] = make_uint4(t.x, t.y, t.z, t.w);
    }
#else
#pragma unroll
    for (int j = 0; j < CPP; j++) c[j] = (qb + j * TR * gsz)[off];
#endif
    const long long sb = tile_pair0 * TR + (long long)g * (KL * TR);
    if (!WTraits<WT>::SCALED) sc = make_uint2(0u, 0u);
    else if (scale_words(WT) == 2) sc = (reinterpret_cast<const uint2 *>(s) + sb)[off];
    else sc = make_uint2((s + sb)[off], 0u);
}

// One wavefront = one 16-row tile x a 1/kw share of the 256-column groups; a workgroup holds
// tw tiles x kw shares.  Decode kernels form a dependent-launch chain, so for small models the
// critical path INSIDE a launch is what matters.  Per wavefront it is:
//   issue {x slice, norm weights, weight chunks, scales} together  ->  one memory latency
//   x*g -> wave-private LDS (no workgroup barrier) -> 128 cvt + 128 fma per lane (8 chains)
//   quad DPP adds -> LDS -> ONE workgroup barrier -> epilogue (inputs prefetched at entry).
template <int WT, int PRO, int EPI, int NP = 1>   // NP: partial vectors fetched per x element (PRO_NORM_PARTS)
__global__ void __launch_bounds__(512) gemv_kernel(GemvParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int CPP = WTraits<WT>::CPP;
    const int nwaves = blockDim.x >> 6;
    float *xs_all = reinterpret_cast<float *>(smem);          // [waves][XS_WAVE]
    float *red = xs_all + nwaves * XS_WAVE;                   // [waves][TR]
    double *dred = reinterpret_cast<double *>(red + nwaves * TR);  // [waves]

    // the wavefront index as an SGPR value: tile, matrix and group bases become scalar arithmetic + a 32-bit lane offset
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // the arguments every variant reads, requested as one batch (the parameter block spans eight cache lines; fetched
    // field by field they were three dependent scalar round trips)
    NL_KARGS8(P.q0, P.q1, P.s0, P.s1, P.x, P.add_src, P.normw, P.out);
    NL_KARGS8(P.rows, P.cols, P.npairs, P.ntiles, P.tw, P.kw, P.eps, P.dbg);
    if (PRO == PRO_ATTN) NL_KARGS4(P.part_o, P.part_ml, P.nsplit_max, P.ctl);
    if (PRO == PRO_NORM_PARTS) NL_KARGS2(P.parts, P.nparts);
    if (EPI == EPI_RESID) NL_KARGS2(P.resid, P.bias_out);
    if (EPI == EPI_P2P) NL_KARGS8(P.resid, P.bias_out, P.p2p_slots, P.p2p_epoch, P.p2p_status, P.p2p_timeout, P.p2p_n, P.p2p_seam);
    if (EPI == EPI_STORE) NL_KARGS4(P.amax_val, P.amax_idx, P.bias_out, P.x_out);
#define NL_STAMP(k) do { if (P.dbg && blockIdx.x == 0 && lane == 0) P.dbg[wave * 8 + (k)] = clock64(); } while (0)
    NL_STAMP(0);
    const int r = lane >> 2, k = lane & 3;
    // EPI_SWIGLU: a tile slot is served by 2*kw wavefronts, the first kw on the gate matrix, the rest on up
    const int wpt = EPI == EPI_SWIGLU ? 2 * P.kw : P.kw;
    const int tin = (int)udiv_by((unsigned)wave, (unsigned)wpt, EPI == EPI_SWIGLU ? P.kw2_inv : P.kw_inv);
    const int msel = EPI == EPI_SWIGLU ? (int)udiv_by((unsigned)(wave - tin * wpt), (unsigned)P.kw, P.kw_inv) : 0;
    const int kw = (wave - tin * wpt) - msel * P.kw;
    const uint8_t *const Wq = msel ? P.q1 : P.q0;
    const uint32_t *const Ws = msel ? P.s1 : P.s0;
    const int tile = blockIdx.x * P.tw + tin;
    const bool live = tile < P.ntiles;
    const long long tp0 = (long long)(live ? tile : 0) * P.npairs;
    const int ngroups = (P.npairs + KL - 1) / KL;
    float *xs = xs_all + wave * XS_WAVE;

    // ---- epilogue inputs, fetched now so their latency hides under the weight stream ----
    const int t = threadIdx.x;
    const int nact = P.tw * TR;
    const int e_tin = t / TR, e_rr = t % TR;
    const int e_tile = blockIdx.x * P.tw + e_tin;
    const bool e_act = t < nact && e_tile < P.ntiles;
    float e_resid = 0.f, e_cos = 0.f, e_sin = 0.f;
    int e_pos = 0;
    if (EPI == EPI_RESID || EPI == EPI_P2P) {
        if (e_act && e_tile * TR + e_rr < P.rows) e_resid = P.resid[e_tile * TR + e_rr];
    }
    if (EPI == EPI_QKV) {
        if (e_act) {
            e_pos = sload_i32(P.ctl + CTL_POS);   // (scalar cache: the weight requests below do not queue behind a vector wait)
            const int half = P.head_dim >> 1, tph = P.head_dim / 16;
            const int i = (e_tile % tph) * 8 + (e_rr & 7);
            e_cos = P.rope_cos[e_pos * half + i];
            e_sin = P.rope_sin[e_pos * half + i];
        }
    }
    unsigned e_tag = 0;
    if (EPI == EPI_P2P) e_tag = ((unsigned)sload_i32(reinterpret_cast<const int *>(P.p2p_epoch)) << 8) | P.p2p_seam;
    // LM head of a per-call Forward (nl_forward): the logits also go straight into the caller-facing pinned host buffer --
    // 64-byte posted writes over the link while the launch runs, instead of a DMA operation behind it (~13 us per call)
    int e_hostout = 0;
    if (EPI == EPI_STORE && P.host_out) e_hostout = sload_i32(P.ctl + CTL_HOSTOUT);
    int ns = 1;
    if (PRO == PRO_ATTN) ns = sload_i32(P.ctl + CTL_POS) / ATT_CH + 1;

    float acc0 = 0.f;
    double ss = 0.0;
    NL_STAMP(1);
    // NF groups in flight per wavefront (all loads of a round are issued before the first dot product)
#ifdef NL_NF
    constexpr int NF = NL_NF;
#else
    constexpr int NF = CPP <= 8 ? 2 : 1;  // 4 in flight for Q4_0 cost occupancy: 5.36 -> 4.77 TB/s on big's LM head
#endif
    // One round = NFR groups of this wavefront.  ALL loads of the round are issued before anything consumes one,
    // activations first (L2-resident: the staging below starts while the weights are still streaming in), and
    // none sits behind a branch: out-of-range lanes load a clamped address and are masked at the use.
    auto round = [&](auto nf_tag, int g0) {
        constexpr int NFR = decltype(nf_tag)::value;
        float4 xv[NFR], gv[NFR], av[NFR];
        float4 pv[NFR][NP];
        uint4 cw[NFR][CPP];
        uint2 sw[NFR];
        bool lv[NFR], inb[NFR];
#pragma unroll
        for (int f = 0; f < NFR; f++) {
            const int gcol = (g0 + f * P.kw) * (KL * PAIR);
            inb[f] = gcol + lane * 4 < P.cols;
            xv[f] = load_x4<PRO>(P, gcol, inb[f] ? (unsigned)lane * 4u : 0u, gv[f], av[f], ns);
            if (PRO == PRO_NORM_PARTS) {
                // all NP loads are issued (clamped part index: no branch around a load), surplus ones are masked at the add
#pragma unroll
                for (int p = 0; p < NP; p++)
                    pv[f][p] = *reinterpret_cast<const float4 *>((P.parts + (size_t)min(p, P.nparts - 1) * P.cols + gcol) + (inb[f] ? (unsigned)lane * 4u : 0u));
            }
        }
#pragma unroll
        for (int f = 0; f < NFR; f++) {
            const int g = g0 + f * P.kw;
            const int gs = min(KL, P.npairs - g * KL);
            lv[f] = live && k < gs;
            load_pair<WT>(Wq, Ws, tp0, g, gs, r, min(k, gs - 1), cw[f], sw[f]);
        }
#pragma unroll
        for (int f = 0; f < NFR; f++) {
            const int g = g0 + f * P.kw;
            float4 xa = xv[f];
            if (PRO != PRO_ATTN && PRO != PRO_NORM_PARTS && P.add) { xa.x += av[f].x; xa.y += av[f].y; xa.z += av[f].z; xa.w += av[f].w; }
            if (PRO == PRO_NORM_PARTS) {
#pragma unroll
                for (int p = 0; p < NP; p++) {
                    const bool on = p < P.nparts;
                    xa.x += on ? pv[f][p].x : 0.f; xa.y += on ? pv[f][p].y : 0.f;
                    xa.z += on ? pv[f][p].z : 0.f; xa.w += on ? pv[f][p].w : 0.f;
                }
            }
            if (!inb[f]) xa = make_float4(0.f, 0.f, 0.f, 0.f);
            if (PRO == PRO_NORM || PRO == PRO_NORM_PARTS) {
                // RMSNormInto go/quant.go:597-607.  inv = 1/sqrt(mean(x^2)+eps) multiplies the GEMV OUTPUT
                // (out = inv * sum_j w_ij (x_j g_j)), so its float64 reduction is off the critical path.
                if (tin == 0 && msel == 0) {
                    // (the product of two float32 values is exact in float64, so the fused form rounds exactly as
                    // the reference's separate multiply and add do)
                    ss = fma((double)xa.x, (double)xa.x, ss); ss = fma((double)xa.y, (double)xa.y, ss);
                    ss = fma((double)xa.z, (double)xa.z, ss); ss = fma((double)xa.w, (double)xa.w, ss);
                    if (P.x_out && blockIdx.x == 0 && inb[f])
                        *reinterpret_cast<float4 *>(P.x_out + g * (KL * PAIR) + lane * 4) = xa;
                }
                xa.x *= gv[f].x; xa.y *= gv[f].y; xa.z *= gv[f].z; xa.w *= gv[f].w;
            }
            *reinterpret_cast<float4 *>(xs + (lane >> 4) * XS_PAIR + (lane & 15) * 4) = xa;
            __builtin_amdgcn_wave_barrier();
            const float a1 = PairDot<WT>::run(cw[f], sw[f], xs + k * XS_PAIR, acc0);
            acc0 = lv[f] ? a1 : acc0;
            __builtin_amdgcn_wave_barrier();
        }
    };
    {
        int g0 = kw;
        if (NF > 1)
            for (; g0 + (NF - 1) * P.kw < ngroups; g0 += NF * P.kw) round(std::integral_constant<int, NF>{}, g0);
        for (; g0 < ngroups; g0 += P.kw) round(std::integral_constant<int, 1>{}, g0);
    }
    NL_STAMP(4);
    // the 4 pair-lanes of a row form a quad
    acc0 = quad_sum(acc0);
    if (k == 0) red[wave * TR + r] = acc0;
    if ((PRO == PRO_NORM || PRO == PRO_NORM_PARTS) && tin == 0 && msel == 0) {
        ss = wave_sum_f64(ss);
        if (lane == 0) dred[kw] = ss;
    }
    NL_STAMP(5);
    __syncthreads();
    NL_STAMP(6);

    if ((t & ~63) >= nact) return;  // this wave holds no output rows
    const int rr = e_rr, otile = e_tile;
    const bool act = e_act;
    float v = 0.f, v1 = 0.f;
    if (act)
        for (int j = 0; j < P.kw; j++) {  // fixed order: deterministic
            v += red[(e_tin * wpt + j) * TR + rr];
            if (EPI == EPI_SWIGLU) v1 += red[(e_tin * wpt + P.kw + j) * TR + rr];
        }
    if (PRO == PRO_NORM || PRO == PRO_NORM_PARTS) {
        double tot = 0.0;
        for (int w = 0; w < P.kw; w++) tot += dred[w];
        float inv = (float)(1.0 / sqrt(tot / (double)P.cols + (double)P.eps));
        v *= inv;
        if (EPI == EPI_SWIGLU) v1 *= inv;
    }
    if (EPI == EPI_QKV) {
        // RoPE (go/model.go:449-477), KV store (:552-554).  Tile rows 0-7 hold
        // element i, rows 8-15 element i + hd/2 of the same head (ROWMAP_HEADPERM).
        if (!act) return;  // 16-lane groups are uniformly active, so RoPE partners stay together
        const int hd = P.head_dim, half = hd >> 1, tph = hd / 16;
        const int head = otile / tph, j = otile % tph;
        const int i = j * 8 + (rr & 7);
        const int e = i + (rr >> 3) * half;
        const int pos = e_pos;
        if (P.bias_q) {  // addBias before RoPE, go/model.go:525-527
            if (head < P.n_q_heads) v += P.bias_q[head * hd + e];
            else if (head < P.n_q_heads + P.n_kv_heads) v += P.bias_k[(head - P.n_q_heads) * hd + e];
            else v += P.bias_v[(head - P.n_q_heads - P.n_kv_heads) * hd + e];
        }
        float partner = __shfl_xor(v, 8);
        float outv = v;
        if (head < P.n_q_heads + P.n_kv_heads) {
            float c = e_cos, sn = e_sin;
            float x0 = (rr < 8) ? v : partner, x1 = (rr < 8) ? partner : v;
            if (!P.rope_conj) outv = (rr < 8) ? (x0 * c - x1 * sn) : (x0 * sn + x1 * c);
            else outv = (rr < 8) ? (x0 * c + x1 * sn) : (-x0 * sn + x1 * c);
        }
        if (head < P.n_q_heads) {
            P.qbuf[head * hd + e] = outv;
        } else {
            const long long soff = (long long)sload_i32(P.ctl + CTL_STREAM) * P.kv_stream_stride;
            if (head < P.n_q_heads + P.n_kv_heads) {
                int kvh = head - P.n_q_heads;
                P.kcache[soff + ((long long)kvh * P.seq_len + pos) * hd + e] = outv;
            } else {
                int kvh = head - P.n_q_heads - P.n_kv_heads;
                P.vcache[soff + ((long long)kvh * P.seq_len + pos) * hd + e] = outv;
            }
        }
        return;
    }
    const int row = otile * TR + rr;
    if (P.bias_out && act && row < P.rows && (EPI == EPI_STORE || EPI == EPI_RESID || EPI == EPI_P2P)) v += P.bias_out[row];
    if (EPI == EPI_P2P) {
        // one 8-byte {tag, value} granule per rank, system scope: the value is its own arrival flag
        if (!act || row >= P.rows) return;
        const unsigned long long gran = ((unsigned long long)e_tag << 32) | __float_as_uint(v);
#pragma unroll
        for (int pr = 0; pr < 8; pr++)
            if (pr < P.p2p_n) __hip_atomic_store(P.p2p_dst[pr] + row, gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        // ... and the all-reduce is finished here, by the row's owner: wait for the granule every rank pushed for this
        // row (the peers run the same launch at the same time, so the wait is the wire's latency while the other
        // workgroups of this launch still stream their weights), add them in rank order (deterministic; bitwise what
        // the in-process shard group computes) and store the new residual.  Bounded: a missing rank sets the status word.
        const bool dead = __hip_atomic_load(P.p2p_status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        unsigned long long g[8];
        const long long t0 = wall_clock64();
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int pr = 0; pr < 8; pr++)
                g[pr] = __hip_atomic_load(P.p2p_slots + (size_t)min(pr, P.p2p_n - 1) * P.rows + row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
            for (int pr = 0; pr < 8; pr++) ok &= (unsigned)(g[pr] >> 32) == e_tag;
            if (ok) break;
            if (dead || wall_clock64() - t0 > P.p2p_timeout) { atomicOr(P.p2p_status, 1u); break; }
            __builtin_amdgcn_s_sleep(2);
        }
        float sum = 0.f;
#pragma unroll
        for (int pr = 0; pr < 8; pr++) sum += pr < P.p2p_n ? __uint_as_float((unsigned)g[pr]) : 0.f;   // fixed rank order
        P.out[row] = e_resid + sum;
        return;
    }
    if (EPI == EPI_STORE) {
        const bool ok = act && row < P.rows;
        if (ok) {
            P.out[row] = v;
#pragma unroll
            for (int pr = 0; pr < 8; pr++)
                if (P.peer_out[pr]) P.peer_out[pr][row] = v;
            if (e_hostout) P.host_out[row] = v;
        }
        if (P.amax_val) {
            // fused partial argmax over this wave's rows (go/main.go:400-408: strict '>' => lowest index
            // wins ties); one slot per wave that holds rows, reduced by argmax_kernel
            float bv = ok ? v : -INFINITY;
            int bi = ok ? row : 0x7fffffff;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                float ov = __shfl_xor(bv, o);
                int oi = __shfl_xor(bi, o);
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            if ((t & 63) == 0) {
                const int spb = (nact + 63) >> 6;
                P.amax_val[blockIdx.x * spb + (t >> 6)] = bv;
                P.amax_idx[blockIdx.x * spb + (t >> 6)] = bi;
            }
        }
        return;
    }
    if (!act || row >= P.rows) return;
    if (EPI == EPI_RESID) {
        P.out[row] = e_resid + v;
    } else if (EPI == EPI_SWIGLU) {
        // SiLU go/quant.go:629-631: x / (1 + f32(exp(f64(-x)))), then * up (go/model.go:604-606)
        float ex = exp_f64_as_f32(-v);
        P.out[row] = (v / (1.0f + ex)) * v1;
    }
    NL_STAMP(7);
#undef NL_STAMP
}

// ------------------------------------------------------------- embedding ---

struct EmbedParams {
    const uint8_t *table;  // raw GGUF rows
    int wtype, dim;
    const int *ctl;
    float *x;
    // gamma injection (go/gamma.go:272-290, go/model.go:503-505): embed[token] += gamma[token] for listed tokens
    const int *gamma_row;     // [vocab] row in gamma_val or -1; nullptr = no gamma
    const float *gamma_val;   // [n][dim]
    unsigned *epoch;          // tensor-parallel push (nl_p2p.h): forward counter, advanced here once per Forward
};

// element i of row `token` of a raw GGUF tensor, dequantised (embedLookupInto go/model.go:389-446 and the
// block dequantisers go/quant.go:22-31,103-108,405-420,296-323,174-208)
__device__ __forceinline__ float embed_value(const uint8_t *table, int wtype, int dim, int token, int i) {
    if (wtype == WT_Q8_0) {
        const uint8_t *blk = table + ((long long)token * (dim / 32) + i / 32) * 34;
        float d = h2f_bits((uint32_t)blk[0] | ((uint32_t)blk[1] << 8));
        return (float)(int)(int8_t)blk[2 + (i & 31)] * d;
    }
    if (wtype == WT_Q4_0) {
        const uint8_t *blk = table + ((long long)token * (dim / 32) + i / 32) * 18;
        float d = h2f_bits((uint32_t)blk[0] | ((uint32_t)blk[1] << 8));
        int j = i & 31;
        int nib = j < 16 ? (blk[2 + j] & 0x0F) : (blk[2 + j - 16] >> 4);
        return (float)(nib - 8) * d;
    }
    if (wtype == WT_Q5_0) {
        const uint8_t *blk = table + ((long long)token * (dim / 32) + i / 32) * 22;
        float d = h2f_bits((uint32_t)blk[0] | ((uint32_t)blk[1] << 8));
        const uint32_t qh = (uint32_t)blk[2] | ((uint32_t)blk[3] << 8) | ((uint32_t)blk[4] << 16) | ((uint32_t)blk[5] << 24);
        int e = i & 31;
        int nib = e < 16 ? (blk[6 + e] & 0x0F) : (blk[6 + e - 16] >> 4);
        return (float)((nib | (int)(((qh >> e) & 1u) << 4)) - 16) * d;
    }
    if (wtype == WT_Q4_K) {
        const uint8_t *blk = table + ((long long)token * (dim / 256) + i / 256) * 144;
        const float d = h2f_bits((uint32_t)blk[0] | ((uint32_t)blk[1] << 8)), dmin = h2f_bits((uint32_t)blk[2] | ((uint32_t)blk[3] << 8));
        const int e = i & 255, j = e >> 6, within = e & 63;
        uint32_t sc, m;
        scale_min_k4(2 * j + (within >> 5), blk + 4, sc, m);
        const uint8_t qb = blk[16 + 32 * j + (within & 31)];
        const float d1 = d * (float)sc, m1 = dmin * (float)m;
        return d1 * (float)(within < 32 ? (qb & 0x0F) : (qb >> 4)) - m1;
    }
    if (wtype == WT_Q6_K) {
        const uint8_t *blk = table + ((long long)token * (dim / 256) + i / 256) * 210;
        const float d = h2f_bits((uint32_t)blk[208] | ((uint32_t)blk[209] << 8));
        int sidx;
        const int q = q6k_value(blk, i & 255, sidx);
        return d * (float)(int)(int8_t)blk[192 + sidx] * (float)(q - 32);
    }
    if (wtype == WT_F16) {
        const uint8_t *p = table + ((long long)token * dim + i) * 2;
        return h2f_bits((uint32_t)p[0] | ((uint32_t)p[1] << 8));
    }
    return reinterpret_cast<const float *>(table)[(long long)token * dim + i];
}

__global__ void embed_kernel(EmbedParams P) {
    if (P.epoch && threadIdx.x == 0) *P.epoch = *P.epoch + 1;
    const int token = sload_i32(P.ctl + CTL_TOKEN);
    const int gr = P.gamma_row ? P.gamma_row[token] : -1;
    // four elements per lane and round, their byte loads issued together (clamped index, masked store): one memory
    // latency per round instead of one per element
    for (int i0 = threadIdx.x; i0 < P.dim; i0 += 4 * blockDim.x) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = embed_value(P.table, P.wtype, P.dim, token, min(i0 + k * (int)blockDim.x, P.dim - 1));
        if (gr >= 0) {
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] += P.gamma_val[(long long)gr * P.dim + min(i0 + k * (int)blockDim.x, P.dim - 1)];
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (i0 + k * (int)blockDim.x < P.dim) P.x[i0 + k * blockDim.x] = v[k];
    }
}

// ------------------------------------------------------------- attention ---

// A multi-token GEMM result as its consumer sees it (nl_batch.h): final values, or -- when the GEMM ran split-K -- the ks
// partial-sum slabs, which the consumer adds in ascending z order (then bias), exactly as qgemm_sum_kernel does.
// Folding the reduction into the consumer removes one launch per GEMM from the batched-decode step.
struct GemmOut {
    const float *val;     // [N][ld], used when ks <= 1 (bias / residual already applied by the GEMM epilogue)
    const float *part;    // [ks][N][ld]
    int ks;
    long long zstride;    // N * ld
    const float *bias;    // optional [ld], applied after the partials when ks > 1
};

struct AttnParams {
    const float *qbuf;             // [n_q_heads][hd], RoPE applied
    const float *kcache, *vcache;  // this layer, stream 0: [kv][seq][hd]
    long long kv_stream_stride;
    float *part_o;                 // [n_q_heads][nsplit_max][hd]
    float *part_ml;                // [n_q_heads][nsplit_max][2]
    const int *ctl;
    int n_kv_heads, seq_len, nsplit_max;
    float scale;
    int single_stream;  // max_streams == 1: stream offset is 0 without reading ctl
    // multi-token step: blockIdx.z = item, per-item position / stream arrays and buffer strides (floats)
    const int *bpos, *bstream;
    long long q_item_stride, part_item_stride;
    // FIN variant (multi-token step whose positions are all < 128, i.e. one split): the normalised output leaves
    // as the WO GEMM's fp16 hi/lo fragments and battn_merge_kernel is not launched
    uint4 *fin_xf;
    int fin_nt16, fin_q4;
    int x1;   // attn_tile16_kernel: prompt precision mode fp16x1 -- one fp16 product per q.k and p.v term instead of hi*hi + lo*hi + hi*lo
    // attn_tile16_kernel: the step's positions are pos_base + item (a prompt) when pos_base_valid -- the workgroup then
    // knows its key range without reading bpos, and its K / V requests leave at entry
    int pos_base_valid, pos_base;
    // ... and the launch is COMPACT then: blockIdx.x indexes the host-built list of workgroups, each a kv head, a query tile
    // and a run of consecutive 128-key chunks under the causal diagonal (kv head << 24 | tile << 16 | partial slot << 12 |
    // first chunk << 6 | chunks) instead of a tiles x chunks grid half of whose workgroups -- the ones above the diagonal --
    // are dispatched, given LDS and exit.  The list is ordered so that the two workgroups of a CU (i and i + #CUs, measured)
    // are a long run and a short one.
    const int *live_map;
    // ... and K / V^T arrive already split into fp16 hi / lo halves and laid out as the workgroup's LDS image, one image per
    // (kv head, 128-key chunk), built once per layer by kv16_build_kernel (null: every workgroup converts its own chunks)
    const uint4 *kv16;
    // FIN variant with the RoPE prologue (decode batches, nl_batch.h attn_rope_prologue): q / k / v of the step's tokens are
    // still the Q|K|V GEMM's output -- packed row order, possibly split-K slabs -- and this kernel rotates and stores them
    struct Rope {
        int on;
        GemmOut qkv;               // [N][R]
        int R, n_q_heads, conj;
        const float *cos, *sin;    // [seq][hd/2]
        const float *bias_q, *bias_k, *bias_v;
        float *kcache_w, *vcache_w;   // = kcache / vcache (written here)
    } rp;
};

template <int HD, int G>
__device__ void attn_finalize(const AttnParams &P, const float *ored, const float *ml, int kvh, int item);   // nl_batch.h
template <int HD, int G>
__device__ void attn_rope_prologue(const AttnParams &P, int kvh, int item, int pos, long long soff, float *qs, float *krow, float *vcur, bool kv_part);   // nl_batch.h

// GQA decode attention for one token (go/model.go:557-587): one workgroup per
// (kv head, 128-position split); the G query heads of the group share every K
// and V element read.  Emits un-normalised partials (max, sum, sum p*v) that
// the WO GEMV prologue merges.  All global loads (q, this split's K and V rows)
// are issued up front so the launch pays ONE memory latency: K goes to LDS
// (padded rows, for the per-position dot), V stays in registers for P*V.
// full-wave f32 max / sum on DPP (row of 16) + two cross-row hops, result valid in every lane
__device__ __forceinline__ float wave_max_f32(float v) {
    v = fmaxf(v, dpp_f32<DPP_QUAD_XOR1>(v));
    v = fmaxf(v, dpp_f32<DPP_QUAD_XOR2>(v));
    v = fmaxf(v, dpp_f32<DPP_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_f32<DPP_ROW_MIRROR>(v));      // every lane: max of its row of 16
    // rows outside the mask receive 0 from the broadcast: take the max only where the broadcast applies
    const int lane = (int)(threadIdx.x & 63);
    const float b15 = dpp_bcast_f32<DPP_ROW_BCAST15, 0xA>(v);
    v = (lane & 16) ? fmaxf(v, b15) : v;           // rows 1, 3: max with rows 0, 2
    const float b31 = dpp_bcast_f32<DPP_ROW_BCAST31, 0xC>(v);
    v = (lane & 32) ? fmaxf(v, b31) : v;           // rows 2, 3: max with row 1 (= rows 0-1)
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_sum_f32(float v) {
    v += dpp_f32<DPP_QUAD_XOR1>(v);
    v += dpp_f32<DPP_QUAD_XOR2>(v);
    v += dpp_f32<DPP_HALF_MIRROR>(v);
    v += dpp_f32<DPP_ROW_MIRROR>(v);               // every lane: sum of its row of 16
    v += dpp_bcast_f32<DPP_ROW_BCAST15, 0xA>(v);   // rows 1, 3 += rows 0, 2
    v += dpp_bcast_f32<DPP_ROW_BCAST31, 0xC>(v);   // rows 2, 3 += row 1 (= rows 0-1): lane 63 holds the total
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

#ifdef NL_ATTN_STAMPS
__device__ long long g_attn_stamps[16];   // developer build (-DNL_ATTN_STAMPS): phase stamps of workgroup (0, 0, item 5), thread 64
#define ATTN_STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 5 && threadIdx.x == 64) g_attn_stamps[(i)] = clock64(); } while (0)
#else
#define ATTN_STAMP(i) do { } while (0)
#endif
template <int HD, int G, bool FIN = false, bool ROPE = FIN>   // ROPE: the instantiation carries the RoPE prologue (AttnParams::rp)
__global__ void __launch_bounds__(ATT_THREADS) attn_kernel(AttnParams P) {
    ATTN_STAMP(0);
    const int split = blockIdx.y, t0 = split * ATT_CH;
    const int kvh = blockIdx.x, tid = threadIdx.x;
    constexpr int KS = HD + 4;               // padded row stride, 16-byte aligned: a lane reads its key's row as float4s and
                                             // consecutive rows sit one 16-byte slot apart (conflict-free ds_read_b128)
    static_assert(ATT_THREADS == 2 * ATT_CH, "scores: thread = (key row, half of the group's heads)");
    constexpr int R4 = HD / 4;               // float4 per row
    constexpr int NG = ATT_THREADS / R4;     // row groups
    constexpr int NV = ATT_CH / NG;          // rows per thread
    // the per-row-group partial outputs (ored, written after the last read of the staged K) reuse K's storage when they fit:
    // 52 -> 37 KB of LDS for HD = 64, G = 4, i.e. four workgroups per CU instead of three -- at long contexts this kernel
    // is bound by the bytes it keeps in flight (goldie x 64 streams at position 1000: 2048 workgroups of 64 KB per layer)
    constexpr bool ORED_IN_K = NG * G * HD <= ATT_CH * KS;
    __shared__ __attribute__((aligned(16))) float Kt[ATT_CH * KS];
    __shared__ __attribute__((aligned(16))) float qs[G * HD];
    __shared__ float sc[G * ATT_CH];
    __shared__ __attribute__((aligned(16))) float ored_own[ORED_IN_K ? 4 : NG * G * HD];
    float *const ored = ORED_IN_K ? Kt : ored_own;
    __shared__ float ml[G * 2];
    __shared__ __attribute__((aligned(16))) float vcur[ROPE ? HD : 4];  // RoPE prologue: this step's V row

    const int c4 = tid % R4, tg = tid / R4;
    float4 kreg[NV], vreg[NV];
    int pos, n;
    // Split 0 of a single-stream engine does not need ctl to know WHERE its K/V rows are, so it loads
    // all 128 rows speculatively (rows > pos hold finite stale data and are masked below) and the
    // ctl round trip overlaps the K/V fetch instead of preceding it.
    const int item = blockIdx.z;
    const float *const qsrc = P.qbuf + (long long)item * P.q_item_stride;
    float *const part_o = P.part_o + (long long)item * P.part_item_stride * HD;
    float *const part_ml = P.part_ml + (long long)item * P.part_item_stride * 2;
    const bool spec = split == 0 && P.single_stream && !P.bpos;
    if (spec) {
        const float4 *K4 = reinterpret_cast<const float4 *>(P.kcache + (long long)kvh * P.seq_len * HD);
        const float4 *V4 = reinterpret_cast<const float4 *>(P.vcache + (long long)kvh * P.seq_len * HD);
        const int lim = min(ATT_CH, P.seq_len);
#pragma unroll
        for (int k = 0; k < NV; k++) {
            int row = tg + k * NG;
            if (row < lim) {
                kreg[k] = K4[row * R4 + c4];
                vreg[k] = V4[row * R4 + c4];
            } else {
                kreg[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                vreg[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        pos = sload_i32(P.ctl + CTL_POS);
        n = min(ATT_CH, pos + 1);
    } else {
        pos = sload_i32(P.bpos ? P.bpos + item : P.ctl + CTL_POS);   // (ctl through the scalar cache: no vector wait before the K / V requests)
        // (the stream is requested with the position, in front of the test on it: one scalar round trip for both -- behind the
        //  early return it was a second, dependent one: 0.3 us of a 4.7 us launch)
        const int strm = sload_i32(P.bstream ? P.bstream + item : P.ctl + CTL_STREAM);
        asm volatile("" :: "s"(pos), "s"(strm));
        if (t0 > pos) return;
        n = min(ATT_CH, pos + 1 - t0);
        const long long soff = (long long)strm * P.kv_stream_stride;
        const float4 *K4 = reinterpret_cast<const float4 *>(P.kcache + soff + ((long long)kvh * P.seq_len + t0) * HD);
        const float4 *V4 = reinterpret_cast<const float4 *>(P.vcache + soff + ((long long)kvh * P.seq_len + t0) * HD);
#pragma unroll
        for (int k = 0; k < NV; k++) {
            int row = tg + k * NG;
            if (row < n) {
                kreg[k] = K4[row * R4 + c4];
                vreg[k] = V4[row * R4 + c4];
            } else {
                kreg[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                vreg[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        if constexpr (ROPE) {
            // (every split rotates its own copy of q; the split that holds `pos` -- the last one -- also rotates k, writes
            // row `pos` of the cache and puts its K straight into the staged tile.  AFTER the cache rows have been requested:
            // the prologue stores into the cache, so placed first it kept the requests behind its own slab round trip -- two
            // round trips per launch instead of one.  What the requests above return for row `pos` is discarded below.)
            if (P.rp.on) {
                __builtin_amdgcn_sched_barrier(0);
                attn_rope_prologue<HD, G>(P, kvh, item, pos, soff, qs, Kt + (pos - t0) * KS, vcur, pos - t0 < ATT_CH);
            }
        }
    }
    ATTN_STAMP(1);
    const bool roped = ROPE && P.rp.on;      // q comes from the RoPE prologue above ...
    const int own = roped && pos - t0 < ATT_CH ? pos - t0 : -1;   // ... and so does this staged row of K / V (the step's own position)
    if (!roped)
        for (int i = tid; i < G * HD; i += ATT_THREADS) qs[i] = qsrc[kvh * G * HD + i];
#pragma unroll
    for (int k = 0; k < NV; k++) {
        int row = tg + k * NG;
        if (row < n && row != own) {
            *reinterpret_cast<float4 *>(Kt + row * KS + c4 * 4) = kreg[k];
        }
    }
    __syncthreads();
    if constexpr (ROPE) {
        if (own >= 0) {
#pragma unroll
            for (int k = 0; k < NV; k++)
                if (tg + k * NG == own) vreg[k] = *reinterpret_cast<const float4 *>(vcur + c4 * 4);
        }
    }

    ATTN_STAMP(2);
    // scores: thread (key row t, half gp of the group's heads) -> q.k over d.  The row is read once, as float4s, for all the
    // heads of the half; q is a broadcast read.  Per (t, g): four partial sums over d = 0, 4, 8, ... / 1, 5, ... / ..., the
    // order the one-float-at-a-time loop of rounds 1-2 used (256 ds_read_b32 per thread against 48 ds_read_b128 now).
    {
        constexpr int GH = (G + 1) / 2;
        const int t = tid & (ATT_CH - 1), gp = tid >> 7;
        if (t < n) {
            float d[GH][4];
#pragma unroll
            for (int gi = 0; gi < GH; gi++) d[gi][0] = d[gi][1] = d[gi][2] = d[gi][3] = 0.f;
            // (sixteen dims per round of LDS reads: one-float4-at-a-time left a dependent LDS round trip per step, 3.8k of a
            // decode-batch launch's 15k cycles)
#pragma unroll 1
            for (int d0 = 0; d0 < HD; d0 += 16) {      // (not unrolled: hipcc otherwise hoists every round's reads -- 230+ registers)
                float4 kv[4], qv[GH][4];
#pragma unroll
                for (int u = 0; u < 4; u++) kv[u] = *reinterpret_cast<const float4 *>(Kt + t * KS + d0 + 4 * u);
#pragma unroll
                for (int gi = 0; gi < GH; gi++)
#pragma unroll
                    for (int u = 0; u < 4; u++) qv[gi][u] = *reinterpret_cast<const float4 *>(qs + min(gp * GH + gi, G - 1) * HD + d0 + 4 * u);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 4; u++)
#pragma unroll
                    for (int gi = 0; gi < GH; gi++) {
                        d[gi][0] = fmaf(qv[gi][u].x, kv[u].x, d[gi][0]);
                        d[gi][1] = fmaf(qv[gi][u].y, kv[u].y, d[gi][1]);
                        d[gi][2] = fmaf(qv[gi][u].z, kv[u].z, d[gi][2]);
                        d[gi][3] = fmaf(qv[gi][u].w, kv[u].w, d[gi][3]);
                    }
            }
#pragma unroll
            for (int gi = 0; gi < GH; gi++) {
                const int g = gp * GH + gi;
                if (g < G) sc[g * ATT_CH + t] = ((d[gi][0] + d[gi][1]) + (d[gi][2] + d[gi][3])) * P.scale;
            }
        }
    }
    __syncthreads();

    ATTN_STAMP(3);
    // softmax pieces per head: wave w handles heads w, w+4, ...
    const int wave = tid >> 6, lane = tid & 63;
    for (int g = wave; g < G; g += ATT_THREADS / 64) {
        float s0 = lane < n ? sc[g * ATT_CH + lane] : -INFINITY;
        float s1 = lane + 64 < n ? sc[g * ATT_CH + lane + 64] : -INFINITY;
        float m = wave_max_f32(fmaxf(s0, s1));
        float p0 = lane < n ? exp_f64_as_f32(s0 - m) : 0.f;       // go/quant.go:619
        float p1 = lane + 64 < n ? exp_f64_as_f32(s1 - m) : 0.f;
        if (lane < n) sc[g * ATT_CH + lane] = p0;
        if (lane + 64 < n) sc[g * ATT_CH + lane + 64] = p1;
        float l = wave_sum_f32(p0 + p1);
        if (lane == 0) { ml[2 * g] = m; ml[2 * g + 1] = l; }
    }
    __syncthreads();

    ATTN_STAMP(4);
    // P*V from the V rows already in registers
    float4 o[G];
#pragma unroll
    for (int g = 0; g < G; g++) o[g] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < NV; k++) {
        int row = tg + k * NG;
        if (row < n) {
#pragma unroll
            for (int g = 0; g < G; g++) {
                float pw = sc[g * ATT_CH + row];
                o[g].x = fmaf(pw, vreg[k].x, o[g].x);
                o[g].y = fmaf(pw, vreg[k].y, o[g].y);
                o[g].z = fmaf(pw, vreg[k].z, o[g].z);
                o[g].w = fmaf(pw, vreg[k].w, o[g].w);
            }
        }
    }
#pragma unroll
    for (int g = 0; g < G; g++) *reinterpret_cast<float4 *>(ored + (tg * G + g) * HD + c4 * 4) = o[g];
    __syncthreads();
    ATTN_STAMP(5);
    if constexpr (FIN) {
        attn_finalize<HD, G>(P, ored, ml, kvh, item);
        ATTN_STAMP(6);
        return;
    }
    for (int i = tid; i < G * HD; i += ATT_THREADS) {
        int g = i / HD, dd = i % HD;
        float s = 0.f;
#pragma unroll 8
        for (int k = 0; k < NG; k++) s += ored[(k * G + g) * HD + dd];
        int h = kvh * G + g;
        part_o[((long long)h * P.nsplit_max + split) * HD + dd] = s;
        if (dd == 0) {
            part_ml[((long long)h * P.nsplit_max + split) * 2] = ml[2 * g];
            part_ml[((long long)h * P.nsplit_max + split) * 2 + 1] = ml[2 * g + 1];
        }
    }
}

// QK-norm (go/model.go:542-549): RMSNormBare per head on q and on the K row
// just stored, after RoPE.  Only launched when nanollama.qk_norm is set.
struct QkNormParams {
    float *qbuf, *kcache;
    long long kv_stream_stride;
    const int *ctl;
    int n_q_heads, n_kv_heads, head_dim, seq_len;
    float eps;
};

__global__ void qknorm_kernel(QkNormParams P) {
    const int head = blockIdx.x, lane = threadIdx.x;  // 64 threads
    const int hd = P.head_dim;
    float *vec;
    if (head < P.n_q_heads) vec = P.qbuf + head * hd;
    else {
        const long long soff = (long long)P.ctl[CTL_STREAM] * P.kv_stream_stride;
        vec = P.kcache + soff + ((long long)(head - P.n_q_heads) * P.seq_len + P.ctl[CTL_POS]) * hd;
    }
    double ss = 0.0;
    for (int i = lane; i < hd; i += 64) ss += (double)vec[i] * (double)vec[i];
    ss = wave_sum_f64(ss);
    float inv = (float)(1.0 / sqrt(ss / (double)hd + (double)P.eps));
    for (int i = lane; i < hd; i += 64) vec[i] = vec[i] * inv;
}

// ---------------------------------------------------------------- argmax ---

struct ArgmaxParams {
    const float *logits;   // full scan source (used when part_val == nullptr)
    int n;
    const float *part_val; // per-wave partial maxima written by the LM-head GEMV epilogue
    const int *part_idx;
    int npart;
    int *ctl;
    int *ids;     // ring of sampled ids (chained decode)
    int *result;  // last argmax
};

// argmax go/main.go:400-408: strict '>' => lowest index wins ties.
__global__ void __launch_bounds__(1024) argmax_kernel(ArgmaxParams P) {
    __shared__ float bv[16];
    __shared__ int bi[16];
    const int tid = threadIdx.x;
    float best = -INFINITY;
    int idx = 0x7fffffff;
    if (P.part_val) {
        for (int i = tid; i < P.npart; i += blockDim.x) {
            float v = P.part_val[i];
            int vi = P.part_idx[i];
            if (v > best || (v == best && vi < idx)) { best = v; idx = vi; }
        }
    } else {
        for (int i = tid; i < P.n; i += blockDim.x) {
            float v = P.logits[i];
            if (v > best || idx == 0x7fffffff) { best = v; idx = i; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(best, o);
        int oi = __shfl_xor(idx, o);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    if ((tid & 63) == 0) { bv[tid >> 6] = best; bi[tid >> 6] = idx; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); w++)
            if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
        if (idx == 0x7fffffff) idx = 0;  // all-NaN logits: the reference's loop never leaves index 0
        *P.result = idx;
        if (P.ctl[CTL_CHAIN]) {
            int step = P.ctl[CTL_STEP];
            P.ids[step] = idx;
            P.ctl[CTL_STEP] = step + 1;
            P.ctl[CTL_TOKEN] = idx;
            P.ctl[CTL_POS] = P.ctl[CTL_POS] + 1;
        }
    }
}

// argmax of step k + embedding lookup of step k+1 in one launch (chained greedy decode inside the multi-step
// graph): the token goes from the reduction to the row fetch through LDS instead of through ctl and a kernel boundary
__global__ void __launch_bounds__(1024) argmax_embed_kernel(ArgmaxParams P, EmbedParams E) {
    __shared__ float bv[16];
    __shared__ int bi[16];
    __shared__ int tok;
    const int tid = threadIdx.x;
    // the decode-state words this launch updates were written by the previous step's launch: requested now, through the
    // scalar cache, so that after the reduction only stores remain (they were four dependent round trips of thread 0)
    const int c_chain = sload_i32(P.ctl + CTL_CHAIN), c_step = sload_i32(P.ctl + CTL_STEP), c_pos = sload_i32(P.ctl + CTL_POS);
    const int c_epoch = E.epoch ? sload_i32(reinterpret_cast<const int *>(E.epoch)) : 0;
    float best = -INFINITY;
    int idx = 0x7fffffff;
    if (P.part_val) {
        for (int i = tid; i < P.npart; i += blockDim.x) {
            float v = P.part_val[i];
            int vi = P.part_idx[i];
            if (v > best || (v == best && vi < idx)) { best = v; idx = vi; }
        }
    } else {
        for (int i = tid; i < P.n; i += blockDim.x) {
            float v = P.logits[i];
            if (v > best || idx == 0x7fffffff) { best = v; idx = i; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(best, o);
        int oi = __shfl_xor(idx, o);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    if ((tid & 63) == 0) { bv[tid >> 6] = best; bi[tid >> 6] = idx; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); w++)
            if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
        if (idx == 0x7fffffff) idx = 0;  // all-NaN logits: the reference's loop never leaves index 0
        tok = idx;
        *P.result = idx;
        if (c_chain) {
            P.ids[c_step] = idx;
            P.ctl[CTL_STEP] = c_step + 1;
            P.ctl[CTL_TOKEN] = idx;
            P.ctl[CTL_POS] = c_pos + 1;
        }
        if (E.epoch) *E.epoch = (unsigned)c_epoch + 1u;   // this embedding opens the next Forward
    }
    __syncthreads();
    const int token = tok;
    const int gr = E.gamma_row ? E.gamma_row[token] : -1;
    for (int i = tid; i < E.dim; i += blockDim.x) {
        float v = embed_value(E.table, E.wtype, E.dim, token, i);
        if (gr >= 0) v += E.gamma_val[(long long)gr * E.dim + i];
        E.x[i] = v;
    }
}


// Sum of tensor-parallel partial vectors living in ONE process (the in-process stand-in for the
// RCCL all-reduce, used by nl_group_forward): every buffer ends up holding the sum.
struct PtrList8 { float *p[8]; };
__global__ void local_allreduce_kernel(PtrList8 bufs, int nranks, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int k = 0; k < nranks; k++) s += bufs.p[k][i];   // fixed rank order: deterministic
    for (int k = 0; k < nranks; k++) bufs.p[k][i] = s;
}

}  // namespace nl
