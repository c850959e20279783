// nl_qgemm2.h -- the multi-token GEMM for LONG token runs (prompt prefill): weights through LDS, activations in registers.
//
// qgemm_kernel (nl_qgemm.h) gives every wavefront its own 16 weight rows and shares the activation fragments of 64 tokens
// through LDS: per 32-column block a wavefront issues 8 ds_read_b128 for 8 MFMAs, expands its weight fragment itself
// and spends 16 VALU FMAs on the block scale -- LDS reads, VALU and MFMA issue are co-limited (measured: MFMA pipe
// 29 % busy at 2047 tokens, and neither a 3-MFMA scaled-weight form nor two row tiles per wavefront helped).
// Here the roles are swapped.  A workgroup owns 128 weight rows x 128 tokens; its 512 threads expand the 128 x 128-column
// chunk of weights ONCE into fp16 MFMA fragments in LDS (one (row, block) item per thread), every wavefront owns one
// 16-token tile whose hi / lo activation fragments it loads straight from the fragment store into registers, and per
// block it multiplies them against all 8 row tiles: 8 ds_read_b128 now feed 16 MFMAs, the weight expansion costs 1/8 of
// the VALU work per MFMA, and the 32 scale FMAs ride under 256 cycles of MFMA issue.
// Same arithmetic per output as qgemm_kernel: exact integer quants in fp16, x = hi + lo, f32 block sums, * d in f32.
#pragma once
#include "nl_qgemm.h"

namespace nl {

constexpr int QG2_ROWS = 128, QG2_TOK = 128, QG2_WAVES = 8, QG2_RT = QG2_ROWS / TR;

// raw quant bytes of one (row, block): Q4_0 16 bytes, Q8_0 32 bytes
template <int WT> struct RowBlock;
template <> struct RowBlock<WT_Q4_0> {
    uint4 u;
    static __device__ __forceinline__ RowBlock load(const uint8_t *q, unsigned tile_group_pairs, int gsz, int i, int blk) {
        // chunk c of pair p == block 2p + c:  q + ((tile*npairs + g*KL)*2*TR + (c*TR + i)*gsz + k)*16
        RowBlock r;
        r.u = *reinterpret_cast<const uint4 *>(q + ((size_t)tile_group_pairs * (2 * TR) + (size_t)((blk & 1) * TR + i) * gsz + ((blk >> 1) & 3)) * 16);
        return r;
    }
    __device__ __forceinline__ half8_t frag(int w) const {
        const uint32_t d = w == 0 ? u.x : w == 1 ? u.y : w == 2 ? u.z : u.w;
        return WFrag<WT_Q4_0>::expand(d);
    }
};
template <> struct RowBlock<WT_Q8_0> {
    uint4 lo, hi;
    static __device__ __forceinline__ RowBlock load(const uint8_t *q, unsigned tile_group_pairs, int gsz, int i, int blk) {
        // block 2p + h = chunks 2h, 2h+1 of pair p:  q + ((tile*npairs + g*KL)*4*TR + (c*TR + i)*gsz + k)*16
        RowBlock r;
        const uint8_t *base = q + ((size_t)tile_group_pairs * (4 * TR) + (size_t)((blk & 1) * 2 * TR + i) * gsz + ((blk >> 1) & 3)) * 16;
        r.lo = *reinterpret_cast<const uint4 *>(base);
        r.hi = *reinterpret_cast<const uint4 *>(base + (size_t)TR * gsz * 16);
        return r;
    }
    __device__ __forceinline__ half8_t frag(int w) const {
        const uint4 &c = w < 2 ? lo : hi;
        return WFrag<WT_Q8_0>::expand((w & 1) ? make_uint2(c.z, c.w) : make_uint2(c.x, c.y));
    }
};

// Plain epilogue (out / resid / bias, or split-K partial slabs), as qgemm_kernel's.
template <int WT>
__global__ void __launch_bounds__(QG2_WAVES * 64, 2) qgemm2_kernel(QGemmParams P) {
    __shared__ __attribute__((aligned(16))) uint4 wfrag[2][QG_KC][QG2_RT][64];   // [buffer][block][row tile][lane] fp16 x 8
    __shared__ float wsc[2][QG_KC][QG2_ROWS];                                     // block scales
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int li = lane & 15;
    const int nblocks = P.cols / 32, nchunks = (nblocks + QG_KC - 1) / QG_KC;
    const int row0 = blockIdx.x * QG2_ROWS, tile0 = row0 / TR;
    const int ttile = blockIdx.y * (QG2_TOK / 16) + wave;            // this wavefront's 16-token tile
    const bool tlive = ttile * 16 < P.n_tokens;

    // staging item of this thread: (row s_row of the workgroup, block s_blk of the chunk)
    const int s_row = tid & (QG2_ROWS - 1), s_blk = tid >> 7, s_rt = s_row >> 4, s_i = s_row & 15;
    const int s_tile = min(tile0 + s_rt, P.ntiles - 1);
    auto stage_load = [&](int chunk, RowBlock<WT> &rb, uint32_t &sw) {
        const int blk = min(chunk * QG_KC + s_blk, nblocks - 1);
        const int g = blk >> 3, gsz = min(KL, P.npairs - g * KL);
        const unsigned gp = (unsigned)(s_tile * P.npairs + g * KL);
        rb = RowBlock<WT>::load(P.q, gp, gsz, s_i, blk);
        sw = P.s[(size_t)gp * TR + (size_t)s_i * gsz + ((blk >> 1) & 3)];
    };
    auto stage_store = [&](int chunk, int buf, const RowBlock<WT> &rb, uint32_t sw) {
        const int blk = chunk * QG_KC + s_blk;
#pragma unroll
        for (int w = 0; w < 4; w++) wfrag[buf][s_blk][s_rt][w * 16 + s_i] = __builtin_bit_cast(uint4, rb.frag(w));
        wsc[buf][s_blk][s_row] = blk < nblocks ? scale_of(sw, blk) : 0.f;   // a block past the end of K contributes nothing
    };
    // activation fragments of this wavefront's token tile: [block][tile][hi/lo][lane] x 16 B
    const uint4 *const xbase = P.xf + ((size_t)min(ttile, P.nt16 - 1) * 2) * QG_FRAG + lane;
    const size_t xblock = (size_t)P.nt16 * 2 * QG_FRAG;
    auto xload = [&](int chunk, uint4 (&xh)[QG_KC], uint4 (&xl)[QG_KC]) {
#pragma unroll
        for (int b = 0; b < QG_KC; b++) {
            const int blk = min(chunk * QG_KC + b, nblocks - 1);
            xh[b] = xbase[(size_t)blk * xblock];
            xl[b] = xbase[(size_t)blk * xblock + QG_FRAG];
        }
    };

    f32x4_t acc[QG2_RT];
#pragma unroll
    for (int rt = 0; rt < QG2_RT; rt++) acc[rt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    int chunk = blockIdx.z;
    RowBlock<WT> rb;
    uint32_t sw = 0;
    uint4 xh[QG_KC], xl[QG_KC], xhn[QG_KC], xln[QG_KC];
    if (chunk < nchunks) {
        stage_load(chunk, rb, sw);
        xload(chunk, xh, xl);
        stage_store(chunk, 0, rb, sw);
    }
    __syncthreads();
    int buf = 0;
    while (chunk < nchunks) {
        const int nxt = chunk + P.ksplit;
        const bool more = nxt < nchunks;
        // the next chunk's raw weights and activation fragments travel while this chunk is on the matrix cores
        stage_load(more ? nxt : chunk, rb, sw);
        xload(more ? nxt : chunk, xhn, xln);
#pragma unroll
        for (int b = 0; b < QG_KC; b++) {
            const half8_t ah = __builtin_bit_cast(half8_t, xh[b]), al = __builtin_bit_cast(half8_t, xl[b]);
            // four row tiles at a time: the lo-part MFMAs of all four, then the hi-part MFMAs -- each dependent pair is four
            // issues apart, so no MFMA waits for its predecessor's accumulator
            f32x4_t z[QG2_RT];
#pragma unroll
            for (int r0 = 0; r0 < QG2_RT; r0 += 4) {
                half8_t wf[4];
#pragma unroll
                for (int q = 0; q < 4; q++) wf[q] = __builtin_bit_cast(half8_t, wfrag[buf][b][r0 + q][lane]);
#pragma unroll
                for (int q = 0; q < 4; q++) z[r0 + q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, wf[q], (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; q++) z[r0 + q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wf[q], z[r0 + q], 0, 0, 0);
            }
#pragma unroll
            for (int rt = 0; rt < QG2_RT; rt++) {
                const float d = wsc[buf][b][rt * TR + li];
                acc[rt][0] = fmaf(z[rt][0], d, acc[rt][0]);
                acc[rt][1] = fmaf(z[rt][1], d, acc[rt][1]);
                acc[rt][2] = fmaf(z[rt][2], d, acc[rt][2]);
                acc[rt][3] = fmaf(z[rt][3], d, acc[rt][3]);
            }
        }
        if (more) stage_store(nxt, buf ^ 1, rb, sw);     // the other buffer: nobody reads it during this chunk
        __syncthreads();
#pragma unroll
        for (int b = 0; b < QG_KC; b++) { xh[b] = xhn[b]; xl[b] = xln[b]; }
        buf ^= 1;
        chunk = nxt;
    }

    // D[token = (lane>>4)*4 + j][weight row = lane & 15] per row tile
    if (!tlive) return;
    const bool split = P.ksplit > 1;
    float *const dst = split ? P.part + (long long)blockIdx.z * P.n_tokens * P.ldo : P.out;
    const float *const resid = split ? nullptr : P.resid, *const bias = split ? nullptr : P.bias;
    const int tok0 = ttile * 16 + (lane >> 4) * 4;
#pragma unroll
    for (int rt = 0; rt < QG2_RT; rt++) {
        const int row = row0 + rt * TR + li;
        if (tile0 + rt >= P.ntiles || row >= P.rows) continue;
        const float bv = bias ? bias[row] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int n = tok0 + j;
            if (n >= P.n_tokens) continue;
            const size_t off = (size_t)n * P.ldo + row;
            float v = acc[rt][j] + bv;
            if (resid) v += resid[off];
            dst[off] = v;
        }
    }
}

}  // namespace nl
