// nl_qgemm.h -- multi-token path: Y[n][row] = sum_k W[row][k] * X[n][k] for N tokens at once on the
// matrix cores (batched decode streams, prompt prefill).  Reads the SAME row-interleaved weight tiles as
// the decode GEMV (nl_kernels.h) -- no second copy of the model.
//
// Precision: the reference computes f32 dot products per 32-element block, then * d (go/quant.go:74-94,
// :149-165).  Here each block is one v_mfma_f32_16x16x32_f16 pair: the weight operand holds the block's
// integer quants EXACTLY in fp16 (n-8 or int8), the activation operand is split x = x_hi + x_lo into two
// fp16 values (two MFMAs, ~2^-22 relative), products accumulate in f32 inside the MFMA, and the f32 block
// sum is multiplied by the block's fp16 scale d on the VALU -- the same "dot, then * d" structure, so the
// result is float32-grade, not fp16-grade.
#pragma once
#include "nl_kernels.h"

namespace nl {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int QG_TOK = 64;      // tokens per workgroup tile (4 MFMA column tiles of 16)
constexpr int QG_KC = 8;        // 32-element blocks staged per chunk (256 columns)
constexpr int QG_WAVES = 4;
constexpr int QG_RT = 2;        // 16-row weight tiles per wavefront

struct QGemmParams {
    const uint8_t *q;
    const uint32_t *s;
    int rows, cols, npairs, ntiles;
    const float *x;      // [N][ldx] activations
    int ldx, n_tokens;
    float *out;          // [N][ldo]
    int ldo;
    const float *resid;  // optional [N][ldo]: out = resid + y
    const float *bias;   // optional [rows]: y += bias[row] (attention output bias)
    // split-K: blockIdx.z handles chunks z, z+ksplit, ...; with ksplit > 1 the kernel writes partial sums to
    // part[z][N][ldo] and qgemm_sum_kernel adds them in a fixed order (deterministic, no atomics)
    int ksplit;
    float *part;
};

// k-slot -> element-of-block map shared by both MFMA operands.
//   Q4_0: dword w of the 16 quant bytes holds elements 4w..4w+3 (low nibbles) and 16+4w..16+4w+3 (high);
//         slots (w, j): [4w, 4w+2, 4w+1, 4w+3, 16+4w, 16+4w+2, 16+4w+1, 16+4w+3]
//   Q8_0 / F16: slots (w, j) = 8w + j
template <int WT>
__device__ __forceinline__ void load_x_slots(const float *xb, int w, float (&v)[8]) {
    if (WT == WT_Q4_0) {
        float4 a = *reinterpret_cast<const float4 *>(xb + 4 * w);
        float4 b = *reinterpret_cast<const float4 *>(xb + 16 + 4 * w);
        v[0] = a.x; v[1] = a.z; v[2] = a.y; v[3] = a.w;
        v[4] = b.x; v[5] = b.z; v[6] = b.y; v[7] = b.w;
    } else {
        float4 a = *reinterpret_cast<const float4 *>(xb + 8 * w);
        float4 b = *reinterpret_cast<const float4 *>(xb + 8 * w + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
}

// weight fragment (8 fp16 k-slots of one row of one block) straight from the tile layout
template <int WT> struct WFrag;

template <> struct WFrag<WT_Q4_0> {
    // chunk c of pair p == block 2p+c; lane (row i, dword w) reads one dword
    static __device__ __forceinline__ half8_t load(const uint8_t *q, long long tile_pair0, int npairs, int blk, int i, int w) {
        const int p = blk >> 1, c = blk & 1, g = p >> 2, k = p & 3;
        const int gsz = min(KL, npairs - g * KL);
        const uint32_t *src = reinterpret_cast<const uint32_t *>(
            q + ((tile_pair0 * 2 * TR) + (long long)g * (KL * 2 * TR) + (c * TR + i) * gsz + k) * 16) + w;
        const uint32_t u = *src, u8 = u >> 8;
        const h2_t k1032 = {(_Float16)1032.0f, (_Float16)1032.0f};
        const h2_t k16th = {(_Float16)0.0625f, (_Float16)0.0625f};
        const h2_t km72 = {(_Float16)-72.0f, (_Float16)-72.0f};
        const uint32_t magic = 0x64006400u;
        h2_t e02 = bits_h2(and_or_b32(u, 0x000F000Fu, magic)) - k1032;
        h2_t e13 = bits_h2(and_or_b32(u8, 0x000F000Fu, magic)) - k1032;
        h2_t f02 = __builtin_elementwise_fma(bits_h2(and_or_b32(u, 0x00F000F0u, magic)), k16th, km72);
        h2_t f13 = __builtin_elementwise_fma(bits_h2(and_or_b32(u8, 0x00F000F0u, magic)), k16th, km72);
        half8_t r;
        r[0] = e02.x; r[1] = e02.y; r[2] = e13.x; r[3] = e13.y;
        r[4] = f02.x; r[5] = f02.y; r[6] = f13.x; r[7] = f13.y;
        return r;
    }
};

template <> struct WFrag<WT_Q8_0> {
    // block 2p+h = chunks 2h, 2h+1 of pair p; lane (row i, slot group w) reads bytes 8w..8w+7 of the block
    static __device__ __forceinline__ half8_t load(const uint8_t *q, long long tile_pair0, int npairs, int blk, int i, int w) {
        const int p = blk >> 1, c = (blk & 1) * 2 + (w >> 1), g = p >> 2, k = p & 3;
        const int gsz = min(KL, npairs - g * KL);
        const uint2 u = *reinterpret_cast<const uint2 *>(
            q + ((tile_pair0 * 4 * TR) + (long long)g * (KL * 4 * TR) + (c * TR + i) * gsz + k) * 16 + (w & 1) * 8);
        half8_t r;
        r[0] = (_Float16)(int)(int8_t)(u.x & 0xff); r[1] = (_Float16)(int)(int8_t)((u.x >> 8) & 0xff);
        r[2] = (_Float16)(int)(int8_t)((u.x >> 16) & 0xff); r[3] = (_Float16)(int)(int8_t)(u.x >> 24);
        r[4] = (_Float16)(int)(int8_t)(u.y & 0xff); r[5] = (_Float16)(int)(int8_t)((u.y >> 8) & 0xff);
        r[6] = (_Float16)(int)(int8_t)((u.y >> 16) & 0xff); r[7] = (_Float16)(int)(int8_t)(u.y >> 24);
        return r;
    }
};

template <int WT>
__device__ __forceinline__ float load_scale(const uint32_t *s, long long tile_pair0, int npairs, int blk, int i) {
    const int p = blk >> 1, g = p >> 2, k = p & 3;
    const int gsz = min(KL, npairs - g * KL);
    const uint32_t sc = s[tile_pair0 * TR + g * (KL * TR) + i * gsz + k];
    return h2f_bits((sc >> (16 * (blk & 1))) & 0xffff);
}

// Workgroup = 4 wavefronts x QG_RT row tiles (128 weight rows) x 64 tokens; K walked in 256-column chunks
// whose activations are split into fp16 hi/lo MFMA fragments in LDS once and shared by all wavefronts.
template <int WT>
__global__ void __launch_bounds__(QG_WAVES * 64) qgemm_kernel(QGemmParams P) {
    // fragment store: [block in chunk][token tile][hi/lo][lane] x 16 bytes
    __shared__ __attribute__((aligned(16))) uint4 xfrag[QG_KC * 4 * 2 * 64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int li = lane & 15, lw = lane >> 4;
    const int tok0 = blockIdx.y * QG_TOK;
    const int nblocks = P.cols / 32;
    const int tile0 = (blockIdx.x * QG_WAVES + wave) * QG_RT;

    f32x4_t acc[QG_RT][4];
#pragma unroll
    for (int rt = 0; rt < QG_RT; rt++)
#pragma unroll
        for (int t = 0; t < 4; t++) acc[rt][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    for (int b0 = blockIdx.z * QG_KC; b0 < nblocks; b0 += QG_KC * P.ksplit) {
        const int nb = min(QG_KC, nblocks - b0);
        // this wavefront's weight fragments + scales for the chunk (issued before the staging work)
        half8_t wf[QG_RT][QG_KC];
        float wd[QG_RT][QG_KC];
#pragma unroll
        for (int rt = 0; rt < QG_RT; rt++) {
            const int tile = tile0 + rt;
            const bool live = tile < P.ntiles;
            const long long tp0 = (long long)(live ? tile : 0) * P.npairs;
#pragma unroll
            for (int b = 0; b < QG_KC; b++) {
                if (live && b < nb) {
                    wf[rt][b] = WFrag<WT>::load(P.q, tp0, P.npairs, b0 + b, li, lw);
                    wd[rt][b] = load_scale<WT>(P.s, tp0, P.npairs, b0 + b, li);
                } else {
                    wf[rt][b] = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
                    wd[rt][b] = 0.f;
                }
            }
        }
        __syncthreads();  // previous chunk's fragments fully consumed
        // stage activations: fragment (block b, token tile t, lane (n, w)) = 8 k-slots of token tok0+16t+n
        for (int f = tid; f < nb * 4 * 64; f += QG_WAVES * 64) {
            const int fl = f & 63, t = (f >> 6) & 3, b = f >> 8;
            const int n = tok0 + t * 16 + (fl & 15), w = fl >> 4;
            float v[8];
            if (n < P.n_tokens) load_x_slots<WT>(P.x + (long long)n * P.ldx + (b0 + b) * 32, w, v);
            else
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = 0.f;
            half8_t hi, lo;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                hi[j] = (_Float16)v[j];
                lo[j] = (_Float16)(v[j] - (float)hi[j]);
            }
            xfrag[((b * 4 + t) * 2 + 0) * 64 + fl] = __builtin_bit_cast(uint4, hi);
            xfrag[((b * 4 + t) * 2 + 1) * 64 + fl] = __builtin_bit_cast(uint4, lo);
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < QG_KC; b++) {
            if (b < nb) {
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const half8_t xh = __builtin_bit_cast(half8_t, xfrag[((b * 4 + t) * 2 + 0) * 64 + lane]);
                    const half8_t xl = __builtin_bit_cast(half8_t, xfrag[((b * 4 + t) * 2 + 1) * 64 + lane]);
#pragma unroll
                    for (int rt = 0; rt < QG_RT; rt++) {
                        f32x4_t z = f32x4_t{0.f, 0.f, 0.f, 0.f};
                        z = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl, wf[rt][b], z, 0, 0, 0);
                        z = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, wf[rt][b], z, 0, 0, 0);
                        const float d = wd[rt][b];
                        acc[rt][t][0] = fmaf(z[0], d, acc[rt][t][0]);
                        acc[rt][t][1] = fmaf(z[1], d, acc[rt][t][1]);
                        acc[rt][t][2] = fmaf(z[2], d, acc[rt][t][2]);
                        acc[rt][t][3] = fmaf(z[3], d, acc[rt][t][3]);
                    }
                }
            }
        }
    }
    // D[token = (lane>>4)*4 + j][weight row = lane & 15]
#pragma unroll
    for (int rt = 0; rt < QG_RT; rt++) {
        const int row = (tile0 + rt) * TR + li;
        if (tile0 + rt >= P.ntiles || row >= P.rows) continue;
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int n = tok0 + t * 16 + lw * 4 + j;
                if (n < P.n_tokens) {
                    float v = acc[rt][t][j];
                    if (P.ksplit > 1) {
                        P.part[((long long)blockIdx.z * P.n_tokens + n) * P.ldo + row] = v;
                    } else {
                        if (P.bias) v += P.bias[row];
                        if (P.resid) v += P.resid[(long long)n * P.ldo + row];
                        P.out[(long long)n * P.ldo + row] = v;
                    }
                }
            }
    }
}

// out[n][row] = (resid) + sum_z part[z][n][row], z in ascending order
__global__ void qgemm_sum_kernel(const float *part, int ksplit, long long count, const float *resid, float *out,
                                 const float *bias, int ldo) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x) {
        float v = part[i];
        for (int z = 1; z < ksplit; z++) v += part[(long long)z * count + i];
        if (bias) v += bias[i % ldo];
        if (resid) v += resid[i];
        out[i] = v;
    }
}

}  // namespace nl
