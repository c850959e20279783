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
#ifndef NL_QG_KC
#define NL_QG_KC 4
#endif
#ifndef NL_QG_OCC
#define NL_QG_OCC 4
#endif
constexpr int QG_KC = NL_QG_KC;  // 32-element blocks per K chunk (128 columns)
#ifndef NL_QG_WAVES
#define NL_QG_WAVES 8
#endif
#ifndef NL_QG_RT
#define NL_QG_RT 1
#endif
constexpr int QG_WAVES = NL_QG_WAVES;
constexpr int QG_RT = NL_QG_RT;  // 16-row weight tiles per wavefront
constexpr int QG_EPI_PLAIN = 0, QG_EPI_SWIGLU = 1, QG_EPI_ROPE = 2;
constexpr int QG_FRAG = 64;     // uint4 per activation fragment: 64 lanes x 16 bytes, lane-linear (= one LDS-DMA)

// Activation fragments live in global memory in MFMA operand order, produced ONCE per activation matrix
// (xsplit_kernel or the producing kernel's epilogue) instead of once per weight row block:
//   xf[block b][16-token tile t][hi/lo p][lane (token n%16, slot group w)] = 8 fp16 k-slots
// nt16 (the t extent) is padded to a multiple of 4 so a 64-token workgroup tile is 8 KB contiguous per block.
struct QGemmParams {
    const uint8_t *q;
    const uint32_t *s;
    int rows, cols, npairs, ntiles;
    const uint4 *xf;     // activation fragments
    int nt16, n_tokens;
    int x1;              // prompt precision mode fp16x1 (qgemm2_kernel): only the hi half of every activation is multiplied
    float *out;          // [N][ldo]
    int ldo;
    const float *resid;  // optional [N][ldo]: out = resid + y
    const float *bias;   // optional [rows]: y += bias[row] (attention output bias)
    // split-K: blockIdx.z handles chunks z, z+ksplit, ...; with ksplit > 1 the kernel writes partial sums to
    // part[z][N][ldo] and qgemm_sum_kernel adds them in a fixed order (deterministic, no atomics)
    int ksplit;
    float *part;
    // optional second matrix of the same shape and type sharing the input (gate + up in one launch): workgroup
    // columns [row_groups, 2*row_groups) run it and write out1 / part1
    const uint8_t *q1;
    const uint32_t *s1;
    float *out1, *part1;
    int row_groups;
    // fused SwiGLU epilogue (QG_EPI_SWIGLU): q = gate, q1 = up; the result leaves as fragments for the next GEMM
    uint4 *xf_out;
    int out_q4;          // k-slot order of the consumer's weight type
    // fused RoPE + KV-store epilogue (QG_EPI_ROPE): the matrix is a layer's packed Q|K|V (ROWMAP_HEADPERM tiles)
    struct Rope {
        const int *pos, *stream;             // per token
        const float *cos, *sin;              // [seq][hd/2]
        float *q;                            // [N][n_q_heads*hd], natural order
        float *kcache, *vcache;              // this layer, stream 0: [kv][seq][hd]
        long long kv_stream_stride;
        const float *bias_q, *bias_k, *bias_v;
        int head_dim, n_q_heads, n_kv_heads, seq_len, conj;
        // dgemm_kernel (nl_dgemm.h): the step's positions resolved once per token by the embedding launch (bembed_kernel) -- the RoPE
        // rows cos / sin[pos[n]][hd / 2] and stream[n] * kv_stream_stride + pos[n] * hd -- so that no launch of a layer chases pos -> table
        const float *tcos, *tsin;            // [N][hd/2]
        const long long *tkv;                // [N]
    } rope;
    // RMSNorm (go/quant.go:597-607) folded around the GEMMs of a prompt (nl_qgemm2.h).  A producer -- WO or down with
    // the plain epilogue, unsplit, so its lanes hold finished rows of the new residual stream -- also emits those rows
    // as the NEXT GEMM's fragments, multiplied by that GEMM's norm weights but not yet by 1 / rms, plus one float64
    // sum of squares per (token, 64-row block).  The consumer (Q|K|V with the RoPE epilogue, gate || up with the
    // SwiGLU epilogue) adds a token's partial sums in block order and scales its OUTPUT by inv, exactly as the decode
    // GEMV does (nl_kernels.h: out = inv * sum_j w_ij (x_j g_j)).  No bnorm launch, no second pass over x.
    struct NormOut {
        const float *w;      // [rows] norm weights of the consuming GEMM; nullptr = not folded
        uint4 *xf;           // fragment store the consumer reads
        double *ssq;         // [n_tokens][rows / 64]
        const float *scale;  // [n_tokens] exact power of two the fragments are multiplied by (norm_prescale below); nullptr = 1
    } nrm_out;               // (the consumer is a Q4_0 GEMM of nl_qgemm2.h: Q4_0 k-slot order)
    struct NormIn {
        const double *ssq;   // nullptr = the input fragments are already normalised
        int nrb;             // partial sums per token (dim / 64)
        int dim;
        float eps;
        const float *scale;  // the power of two the producer multiplied this token's fragments by (undone through inv); nullptr = 1
        float *scale_next;   // written by the workgroups of row block 0: the pre-scale of the NEXT producer, from this inv
    } nrm_in;
};

inline size_t xfrag_uint4(int cols, int n_tokens) {   // uint4 elements of a fragment store
    return (size_t)(cols / 32) * (size_t)(((n_tokens + 63) / 64) * 4) * 2 * QG_FRAG;
}

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

// weight fragment (8 fp16 k-slots of one row of one block) straight from the tile layout: the kernel issues the
// load (kept in its packed form while the previous chunk computes), expand() makes the fp16 operand
template <int WT> struct WFrag;

template <> struct WFrag<WT_Q4_0> {
    typedef uint32_t raw_t;
    // chunk c of pair p == block 2p+c; lane (row i, dword w) reads one dword at
    //   q + ((tile*npairs + g*KL)*2*TR + (c*TR + i)*gsz + k)*16 + 4w,   p = 4g + k, gsz = pairs in group g,
    // split into a wave-uniform part (group base + blk_off) and a per-lane part that only depends on the group's
    // size: the uniform part lives in SGPRs and the loads take the saddr form
    static constexpr int CPP = 2;
    static __device__ __forceinline__ unsigned lane_off(int i, int w, int gsz) { return (unsigned)(i * gsz) * 16u + (unsigned)w * 4u; }
    static __device__ __forceinline__ unsigned blk_off(int odd, int k, int gsz) { return (unsigned)(odd * TR * gsz + k) * 16u; }
    static __device__ __forceinline__ raw_t zero() { return 0x88888888u; }
    static __device__ __forceinline__ half8_t expand(raw_t u) {
        const uint32_t u8 = u >> 8;
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
    typedef uint2 raw_t;
    // block 2p+h = chunks 2h, 2h+1 of pair p; lane (row i, slot group w) reads bytes 8w..8w+7 of the block:
    //   q + ((tile*npairs + g*KL)*4*TR + (c*TR + i)*gsz + k)*16 + (w&1)*8,   c = 2h + (w>>1)
    static constexpr int CPP = 4;
    static __device__ __forceinline__ unsigned lane_off(int i, int w, int gsz) {
        return (unsigned)(((w >> 1) * TR + i) * gsz) * 16u + (unsigned)(w & 1) * 8u;
    }
    static __device__ __forceinline__ unsigned blk_off(int odd, int k, int gsz) { return (unsigned)(odd * 2 * TR * gsz + k) * 16u; }
    static __device__ __forceinline__ raw_t zero() { return make_uint2(0u, 0u); }
    static __device__ __forceinline__ half8_t expand(raw_t u) {
        half8_t r;
        r[0] = (_Float16)(int)(int8_t)(u.x & 0xff); r[1] = (_Float16)(int)(int8_t)((u.x >> 8) & 0xff);
        r[2] = (_Float16)(int)(int8_t)((u.x >> 16) & 0xff); r[3] = (_Float16)(int)(int8_t)(u.x >> 24);
        r[4] = (_Float16)(int)(int8_t)(u.y & 0xff); r[5] = (_Float16)(int)(int8_t)((u.y >> 8) & 0xff);
        r[6] = (_Float16)(int)(int8_t)((u.y >> 16) & 0xff); r[7] = (_Float16)(int)(int8_t)(u.y >> 24);
        return r;
    }
};

template <> struct WFrag<WT_F16> {
    typedef uint4 raw_t;
    // F16 weights (go/quant.go:527-563) are their own exact fp16 operand: no quants, no scale.  A pair is 8 chunks of 8
    // halves; block 2p+h = chunks 4h..4h+3, and lane (row i, slot group w) reads chunk 4h+w whole:
    //   q + ((tile*npairs + g*KL)*8*TR + ((4h + w)*TR + i)*gsz + k)*16
    static constexpr int CPP = 8;
    static __device__ __forceinline__ unsigned lane_off(int i, int w, int gsz) { return (unsigned)((w * TR + i) * gsz) * 16u; }
    static __device__ __forceinline__ unsigned blk_off(int odd, int k, int gsz) { return (unsigned)(odd * 4 * TR * gsz + k) * 16u; }
    static __device__ __forceinline__ raw_t zero() { return make_uint4(0u, 0u, 0u, 0u); }
    static __device__ __forceinline__ half8_t expand(raw_t u) { return __builtin_bit_cast(half8_t, u); }
};

// the scale word of (tile, pair, row i) holds both blocks' fp16 d: s[(tile*npairs + g*KL)*TR + i*gsz + k]; scale_of() picks one
__device__ __forceinline__ float scale_of(uint32_t word, int blk) { return h2f_bits((word >> (16 * (blk & 1))) & 0xffff); }

// The folded RMSNorm hands x * g to its consumer BEFORE 1 / rms is known, as fp16 hi / lo fragments.  fp16 has 5 exponent
// bits: a residual stream of rms 1e-3 would put every lo half into the denormals (absolute 2^-25 instead of relative 2^-22)
// and one of rms 1e5 would overflow to inf - inf.  So the producer multiplies by an EXACT power of two close to 1 / rms -- the
// largest one not above the inv of the token's previous norm (the stream's rms moves by a small factor from one norm to the
// next) -- and the consumer divides its inv by the same power: both are exact, so for streams that were in range anyway the
// results are bit-identical to the unscaled form.
__device__ __forceinline__ float norm_prescale(float inv) {
    const unsigned e = (__float_as_uint(inv) >> 23) & 0xffu;
    return __uint_as_float((unsigned)min(max((int)e, 127 - 100), 127 + 100) << 23);     // 2^floor(log2 inv), kept inside 2^+-100
}

// x = hi + lo, two fp16 values (~2^-22 relative).  The f32 value is pinned in a register first: when v is a
// product the compiler otherwise derives hi TWICE -- once as cvt(f32 product) for the store, once as
// v_fma_mixlo_f16 (one rounding of the exact product) inside the lo computation -- and on the rare double-rounding
// ties the stored hi and the lo no longer add up to v (error 2^-11 of that element; seen as sporadic 1e-4 logits).
__device__ __forceinline__ void split_hi_lo(float v, _Float16 &hi, _Float16 &lo) {
    asm("" : "+v"(v));
    const _Float16 h = (_Float16)v;
    hi = h;
    lo = (_Float16)(v - (float)h);
}

// x -> fp16 hi/lo fragments.  One thread per (block, 16-token tile, slot group w, token): reads the 8 k-slot
// elements of its token (two float4), writes the hi and the lo fragment entry (16 bytes each; the 16 tokens of a
// tile are adjacent, so a 16-lane group writes 256 contiguous bytes).  Tokens beyond n_tokens are zero.
__global__ void xsplit_kernel(const float *x, int ldx, int nblocks, int n_tokens, int nt16, int q4, uint4 *xf) {
    const long long total = (long long)nblocks * nt16 * 64;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int fl = (int)(i & 63), w = fl >> 4;
        const long long bt = i >> 6;
        const int t = (int)(bt % nt16), b = (int)(bt / nt16);
        const int n = t * 16 + (fl & 15);
        float v[8];
        if (n < n_tokens) {
            const float *xb = x + (long long)n * ldx + b * 32;
            if (q4) load_x_slots<WT_Q4_0>(xb, w, v);
            else load_x_slots<WT_Q8_0>(xb, w, v);
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = 0.f;
        }
        half8_t hi, lo;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            _Float16 h, l;
            split_hi_lo(v[j], h, l);
            hi[j] = h; lo[j] = l;
        }
        xf[(bt * 2 + 0) * QG_FRAG + fl] = __builtin_bit_cast(uint4, hi);
        xf[(bt * 2 + 1) * QG_FRAG + fl] = __builtin_bit_cast(uint4, lo);
    }
}

// the 8 k-slots (w, j) of a 32-column block from its two float4 groups a (columns offa..+3) and b (offb..+3),
// slot_offsets() below: only the two middle lanes of each group differ between the Q4_0 and the linear order
__device__ __forceinline__ void slots_from(int q4, const float4 &a, const float4 &b, float (&v)[8]) {
    v[0] = a.x; v[1] = q4 ? a.z : a.y; v[2] = q4 ? a.y : a.z; v[3] = a.w;
    v[4] = b.x; v[5] = q4 ? b.z : b.y; v[6] = q4 ? b.y : b.z; v[7] = b.w;
}
__device__ __forceinline__ void slot_offsets(int q4, int w, int &offa, int &offb) {
    offa = q4 ? 4 * w : 8 * w;
    offb = q4 ? 16 + 4 * w : 8 * w + 4;
}

// element (0..31) of a 32-column block that k-slot (w, j) of the MFMA operands holds (nl_qgemm.h load_x_slots)
__device__ __forceinline__ int slot_elem(int q4, int w, int j) {
    return q4 ? ((j >> 2) * 16 + 4 * w + ((j & 1) << 1) + ((j >> 1) & 1)) : 8 * w + j;
}

// the 8 k-slots (block blk, slot group w) of token n -> hi/lo entries of the fragment store
__device__ __forceinline__ void store_frag(uint4 *xf, int nt16, int n, int blk, int w, const float (&v)[8]) {
    half8_t hi, lo;
#pragma unroll
    for (int j = 0; j < 8; j++) {
            _Float16 h, l;
            split_hi_lo(v[j], h, l);
            hi[j] = h; lo[j] = l;
        }
    const long long bt = (long long)blk * nt16 + (n >> 4);
    const int fl = w * 16 + (n & 15);
    xf[(bt * 2 + 0) * QG_FRAG + fl] = __builtin_bit_cast(uint4, hi);
    xf[(bt * 2 + 1) * QG_FRAG + fl] = __builtin_bit_cast(uint4, lo);
}

// ... one 16-row half of it: the four k-slots j = 4 half .. 4 half + 3 of group w are 8 bytes of the hi and 8 bytes of the lo entry
// (Q4_0 slot order: rows 4w, 4w + 2, 4w + 1, 4w + 3 of the half -- slots_from)
__device__ __forceinline__ void store_frag_half(uint4 *xf, int nt16, int n, int blk, int w, int half, const float4 &r) {
    typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
    const float v[4] = {r.x, r.z, r.y, r.w};
    half4_t hi, lo;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        _Float16 h, l;
        split_hi_lo(v[j], h, l);
        hi[j] = h; lo[j] = l;
    }
    const long long bt = (long long)blk * nt16 + (n >> 4);
    const int fl = w * 16 + (n & 15);
    reinterpret_cast<uint2 *>(xf + (bt * 2 + 0) * QG_FRAG + fl)[half] = __builtin_bit_cast(uint2, hi);
    reinterpret_cast<uint2 *>(xf + (bt * 2 + 1) * QG_FRAG + fl)[half] = __builtin_bit_cast(uint2, lo);
}

// Workgroup = QG_WAVES wavefronts x QG_RT row tiles (8 x 1: 128 weight rows) x 64 tokens; K walked in 128-column
// chunks.  Eight light wavefronts (one row tile, ~100 VGPRs) at four per SIMD measured 5-15 % faster than four
// heavy ones at two per SIMD for 64-512 tokens and equal at 2047 (tools/qgemm_variants.sh).
// Software pipeline, one barrier per chunk: while chunk c is on the matrix cores, chunk c+1's activation
// fragments travel global -> LDS by LDS-DMA (global_load_lds_dwordx4: the fragment store is lane-linear, one
// instruction per 1 KB fragment, no VGPR round trip) and its packed weights + scales travel to registers.
template <int WT, int WAVES, int RT, int EPI, int NT = 4, bool RSTAGE = false>
__global__ void __launch_bounds__(WAVES * 64, WAVES == 8 ? NL_QG_OCC : 2) qgemm_kernel(QGemmParams P) {
    static_assert(EPI != QG_EPI_SWIGLU || (RT == 2 && NT == 4), "the fused epilogue pairs a gate tile with its up tile");
    static_assert(EPI != QG_EPI_ROPE || (RT == 1 && NT == 4), "one Q|K|V tile per wavefront: rotation partners are rows r and r^8 of a tile");
    static_assert(NT == 1 || NT == 2 || NT == 4, "16-token tiles of the 64-token group this workgroup computes");
    NL_KARGS8(P.q, P.s, P.xf, P.out, P.resid, P.bias, P.part, P.q1);   // one batch of s_load (nl_kernels.h)
    NL_KARGS8(P.rows, P.cols, P.npairs, P.ntiles, P.nt16, P.n_tokens, P.ldo, P.ksplit);
    NL_KARGS4(P.s1, P.out1, P.part1, P.row_groups);
    typedef typename WFrag<WT>::raw_t raw_t;
    // fragment buffers: [buffer][block in chunk][token tile < NT][hi/lo][lane] x 16 bytes.  NT < 4 (decode batches
    // and prompts of <= 16 / 32 tokens): only the first NT tiles of the group are fetched and multiplied.
    __shared__ __attribute__((aligned(16))) uint4 xfrag[2][QG_KC * 2 * NT * QG_FRAG];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int li = lane & 15, lw = lane >> 4;
    const int tok0 = blockIdx.y * QG_TOK;
    const int nblocks = P.cols / 32;
    const int nchunks = (nblocks + QG_KC - 1) / QG_KC;
    // plain: workgroup columns [row_groups, 2*row_groups) run the optional second matrix; wavefront tiles tile0 + rt.
    // fused: row tile rt = 0 is the gate tile and rt = 1 the up tile of the SAME 16 rows (q / q1).
    constexpr bool FUSED = EPI == QG_EPI_SWIGLU;
    const bool second = !FUSED && P.q1 && (int)blockIdx.x >= P.row_groups;
    const uint8_t *const Wq = second ? P.q1 : P.q;
    const uint8_t *const Ws = reinterpret_cast<const uint8_t *>(second ? P.s1 : P.s);
    float *const outp = second ? P.out1 : P.out, *const partp = second ? P.part1 : P.part;
    const int tile0 = FUSED ? (int)blockIdx.x * WAVES + wave : (((int)blockIdx.x - (second ? P.row_groups : 0)) * WAVES + wave) * RT;
    const uint8_t *const xsrc = reinterpret_cast<const uint8_t *>(P.xf + (long long)blockIdx.y * (8 * QG_FRAG));

    f32x4_t acc[RT][NT];
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
        for (int t = 0; t < NT; t++) acc[rt][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // Both prefetch lambdas are BRANCH-FREE: out-of-range blocks / tiles are clamped to a valid address and
    // neutralised afterwards (scale 0, or simply never consumed).  A per-element "load or zero" branch makes
    // hipcc wait vmcnt(0) after every load -- one dependent memory round trip per block instead of one per chunk.
    // Every address is "kernel-argument base + 32-bit offset", the offset being scalar arithmetic plus ONE VALU
    // add of the lane's part (a packed matrix and a fragment store are each < 4 GiB): the prefetch of a chunk
    // was ~130 VALU + ~80 SALU instructions of 64-bit address math in front of the MFMA stream, measured at a
    // third of the kernel's time.
    static_assert(QG_KC == 2 || QG_KC == 4 || QG_KC == 8, "a chunk is a whole number of pairs inside one group");
    const unsigned xlane = (unsigned)lane * 16u;
    const unsigned xblock = (unsigned)P.nt16 * (2 * QG_FRAG * 16);   // bytes of one block's fragments
    // RSTAGE (decode batches: one 64-token tile, a handful of chunks per workgroup): the next chunk's fragments travel
    // through REGISTERS (global_load_dwordx4 now, ds_write_b128 after this chunk's MFMAs) instead of LDS-DMA.  The guide's
    // verdict -- LDS-DMA wins -- is for GEMMs that spend their registers and issue slots on MFMA work; here a wavefront
    // is alone on its SIMD with ~50 instructions per block, and the register path measured 5-8 % faster per launch on
    // every goldie shape at 64 tokens (tools/qgemm_bench.hip, profiles/r03_rejected_qgemm_ring_bench.log has the series).
    constexpr int NFR = 2 * NT * QG_KC / WAVES;   // fragments of a chunk staged per wavefront
    [[maybe_unused]] uint4 xr[RSTAGE ? NFR : 1];
    auto stage = [&](int chunk, int buf) {
        const int b0 = chunk * QG_KC;
        static_assert((2 * NT * QG_KC) % WAVES == 0, "fragments of a chunk divide evenly over the wavefronts");
#pragma unroll
        for (int i = 0; i < NFR; i++) {
            const int f = wave + WAVES * i;        // fragment of the chunk: block f / 2NT, (tile, part) f % 2NT
            const unsigned uo = (unsigned)min(b0 + f / (2 * NT), nblocks - 1) * xblock + (unsigned)(f % (2 * NT)) * (QG_FRAG * 16);
            if constexpr (RSTAGE) xr[i] = ld_off<uint4>(xsrc, uo + xlane);
            else
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(xsrc + (uo + xlane)),
                    (__attribute__((address_space(3))) void *)&xfrag[buf][f * QG_FRAG], 16, 0, 0);
        }
    };
    auto stage_commit = [&](int buf) {     // RSTAGE: the staged registers into the buffer nobody reads during this chunk
        if constexpr (RSTAGE) {
#pragma unroll
            for (int i = 0; i < NFR; i++) xfrag[buf][(wave + WAVES * i) * QG_FRAG + lane] = xr[i];
        }
    };
    const uint8_t *const wq_base = Wq, *const ws_base = Ws;
    auto wload = [&](int chunk, raw_t (&wq)[RT][QG_KC], uint32_t (&wd)[RT][QG_KC / 2]) {
        const int b0 = chunk * QG_KC;
        const int g = b0 >> 3;                         // 8 blocks per group; a chunk never straddles groups
        const int gsz = min(KL, P.npairs - g * KL);
        const unsigned lq = WFrag<WT>::lane_off(li, lw, gsz), ls = (unsigned)(li * gsz) * 4u;
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            const unsigned gp = (unsigned)(min(FUSED ? tile0 : tile0 + rt, P.ntiles - 1) * P.npairs + g * KL);   // tiles past ntiles are never stored
            const unsigned gq = gp * (WFrag<WT>::CPP * TR * 16), gs = gp * (TR * 4);
            const uint8_t *const Wq = FUSED && rt ? P.q1 : wq_base, *const Ws = FUSED && rt ? reinterpret_cast<const uint8_t *>(P.s1) : ws_base;
#pragma unroll
            for (int b = 0; b < QG_KC; b++) {
                const int blk = min(b0 + b, nblocks - 1);
                wq[rt][b] = *reinterpret_cast<const raw_t *>(Wq + ((gq + WFrag<WT>::blk_off(blk & 1, (blk >> 1) & 3, gsz)) + lq));
            }
#pragma unroll
            for (int pp = 0; pp < QG_KC / 2; pp++) {
                const int blk = min(b0 + 2 * pp, nblocks - 1);
                if constexpr (WTraits<WT>::SCALED) wd[rt][pp] = *reinterpret_cast<const uint32_t *>(Ws + ((gs + (unsigned)((blk >> 1) & 3) * 4u) + ls));
                else wd[rt][pp] = 0u;
            }
        }
    };

    // One chunk: issue the next chunk's prefetch (on the last chunk: the same chunk again, into the idle buffers
    // -- no branch around the loads, so the scheduler may spread them through the MFMA stream), then straight-line
    // over the chunk's blocks (a block past the end of K was clamped to the last valid one by the prefetch and
    // gets scale 0).  Per block: the 8 lo-part MFMAs go first on 8 independent accumulators, then the 8 hi-part
    // MFMAs -- each dependent pair is 8 issues apart, so no MFMA waits for its predecessor -- then the 32 "* d"
    // FMAs.  The two register sets swap roles from call to call (no copies).
    auto body = [&](int chunk, int buf, raw_t (&wq)[RT][QG_KC], uint32_t (&wd)[RT][QG_KC / 2],
                    raw_t (&wqn)[RT][QG_KC], uint32_t (&wdn)[RT][QG_KC / 2]) {
        const int nb = min(QG_KC, nblocks - chunk * QG_KC);
        const int nxt = chunk + P.ksplit < nchunks ? chunk + P.ksplit : chunk;
        stage(nxt, buf ^ 1);
        wload(nxt, wqn, wdn);
#pragma unroll
        for (int b = 0; b < QG_KC; b++) {
            half8_t wf[RT], xh[NT], xl[NT];
            float dsc[RT];
#pragma unroll
            for (int t = 0; t < NT; t++) {
                xh[t] = __builtin_bit_cast(half8_t, xfrag[buf][((b * NT + t) * 2 + 0) * QG_FRAG + lane]);
                xl[t] = __builtin_bit_cast(half8_t, xfrag[buf][((b * NT + t) * 2 + 1) * QG_FRAG + lane]);
            }
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                wf[rt] = WFrag<WT>::expand(wq[rt][b]);
                const float d = WTraits<WT>::SCALED ? scale_of(wd[rt][b >> 1], b) : 1.0f;
                dsc[rt] = b < nb ? d : 0.f;
            }
#ifdef NL_QG_SCALEW
            if constexpr (WTraits<WT>::SCALED) {
                // Scale folded into the weight operand: w = d * q as fp16 hi + lo (d is fp16, q a small integer: the product
                // has <= 19 significant bits, so hi = round(d*q) and lo = fma(d, q, -hi) split it exactly; d is taken x 2^8
                // -- undone once after the K loop -- so that lo stays a normal fp16 number).  Three MFMAs per block
                // (hi*hi + hi*lo + lo*hi, the lo*lo term is 2^-22 relative) accumulate straight into the output
                // accumulators: no per-block f32 FMAs, no accumulator read-back between blocks.
#pragma unroll
                for (int rt = 0; rt < RT; rt++) {
                    const uint32_t word = wd[rt][b >> 1];
                    const _Float16 dh = (b < nb) ? (_Float16)(256.0f * h2f_bits((word >> (16 * (b & 1))) & 0xffff)) : (_Float16)0.0f;
                    const half8_t dv = {dh, dh, dh, dh, dh, dh, dh, dh};
                    const half8_t hi = wf[rt] * dv;
                    const half8_t lo = __builtin_elementwise_fma(wf[rt], dv, -hi);
#pragma unroll
                    for (int t = 0; t < NT; t++) acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl[t], hi, acc[rt][t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < NT; t++) acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[t], lo, acc[rt][t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < NT; t++) acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[t], hi, acc[rt][t], 0, 0, 0);
                }
                continue;
            }
#endif
            if constexpr (!WTraits<WT>::SCALED) {
                // no per-block scale: the products accumulate straight into the output accumulators (a block past the end
                // of K was clamped to a valid one by the prefetch: its weight operand is zeroed instead of its scale)
#pragma unroll
                for (int rt = 0; rt < RT; rt++)
                    if (b >= nb) wf[rt] = __builtin_bit_cast(half8_t, make_uint4(0u, 0u, 0u, 0u));
#pragma unroll
                for (int t = 0; t < NT; t++)
#pragma unroll
                    for (int rt = 0; rt < RT; rt++)
                        acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl[t], wf[rt], acc[rt][t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NT; t++)
#pragma unroll
                    for (int rt = 0; rt < RT; rt++)
                        acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[t], wf[rt], acc[rt][t], 0, 0, 0);
                continue;
            }
            f32x4_t z[RT][NT];
#pragma unroll
            for (int t = 0; t < NT; t++)
#pragma unroll
                for (int rt = 0; rt < RT; rt++)
                    z[rt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl[t], wf[rt], (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; t++)
#pragma unroll
                for (int rt = 0; rt < RT; rt++)
                    z[rt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[t], wf[rt], z[rt][t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; t++)
#pragma unroll
                for (int rt = 0; rt < RT; rt++) {
                    acc[rt][t][0] = fmaf(z[rt][t][0], dsc[rt], acc[rt][t][0]);
                    acc[rt][t][1] = fmaf(z[rt][t][1], dsc[rt], acc[rt][t][1]);
                    acc[rt][t][2] = fmaf(z[rt][t][2], dsc[rt], acc[rt][t][2]);
                    acc[rt][t][3] = fmaf(z[rt][t][3], dsc[rt], acc[rt][t][3]);
                }
        }
        stage_commit(buf ^ 1);
        __syncthreads();             // this chunk's buffer is free again; the next chunk's DMA has landed
    };

    raw_t wqa[RT][QG_KC], wqb[RT][QG_KC];
    uint32_t wda[RT][QG_KC / 2], wdb[RT][QG_KC / 2];
    int chunk = blockIdx.z;
    if (chunk < nchunks) {
        stage(chunk, 0);
        wload(chunk, wqa, wda);
        stage_commit(0);
    }
    __syncthreads();                 // (drains the LDS-DMA queue: chunk 0 has landed)
    while (chunk < nchunks) {
        body(chunk, 0, wqa, wda, wqb, wdb);
        chunk += P.ksplit;
        if (chunk >= nchunks) break;
        body(chunk, 1, wqb, wdb, wqa, wda);
        chunk += P.ksplit;
    }
#ifdef NL_QG_SCALEW
    if constexpr (WTraits<WT>::SCALED) {
#pragma unroll
        for (int rt = 0; rt < RT; rt++)
#pragma unroll
            for (int t = 0; t < NT; t++) acc[rt][t] = acc[rt][t] * (1.0f / 256.0f);
    }
#endif
    if constexpr (FUSED) {
        // h = SiLU(gate) * up (go/quant.go:629-631, go/model.go:604-606) for this wavefront's 16 rows x 64 tokens,
        // transposed through LDS (the fragment buffers are idle after the loop's last barrier) into the fp16 hi/lo
        // fragments the down projection reads: the f32 gate / up matrices never exist in memory.
        constexpr int HS = WAVES * TR + 4;           // floats per token row of the staging tile (16-byte aligned)
        static_assert(64 * HS * 4 <= (int)sizeof(xfrag), "staging tile fits the fragment buffers");
        float *const hb = reinterpret_cast<float *>(&xfrag[0][0]);
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float gv = acc[0][t][j], uv = acc[1][t][j];
                const float ex = exp_f64_as_f32(-gv);
                hb[(t * 16 + lw * 4 + j) * HS + wave * TR + li] = (gv / (1.0f + ex)) * uv;
            }
        __syncthreads();
        constexpr int NBLK = WAVES * TR / 32;        // 32-row blocks of the workgroup's rows
        const int blk0 = (int)blockIdx.x * NBLK;
        for (int u = tid; u < 64 * NBLK * 4; u += WAVES * 64) {
            const int token = u / (NBLK * 4), bl = (u >> 2) % NBLK, w = u & 3, n = tok0 + token;
            if (n >= P.n_tokens || (blk0 + bl) * 32 >= P.rows) continue;
            int offa, offb;
            slot_offsets(P.out_q4, w, offa, offb);
            const float4 a = *reinterpret_cast<const float4 *>(hb + token * HS + bl * 32 + offa);
            const float4 b = *reinterpret_cast<const float4 *>(hb + token * HS + bl * 32 + offb);
            float v[8];
            slots_from(P.out_q4, a, b, v);
            store_frag(P.xf_out, P.nt16, n, blk0 + bl, w, v);
        }
        return;
    }
    if constexpr (EPI == QG_EPI_ROPE) {
        // RoPE (go/model.go:449-477) + attention biases (:525-527) + KV store (:552-554) on the accumulators, as the
        // decode GEMV's QKV epilogue does it: tile rows 0-7 hold element i, rows 8-15 element i + hd/2 of one head,
        // so a value's rotation partner is in lane ^ 8 (same token).  The f32 Q|K|V matrix never exists in memory
        // and brope_kv_kernel is not launched.
        const QGemmParams::Rope &R = P.rope;
        if (tile0 >= P.ntiles) return;
        const int hd = R.head_dim, half = hd >> 1, tph = hd / 16;
        const int head = tile0 / tph, i = (tile0 % tph) * 8 + (li & 7), e = i + (li >> 3) * half;
        const bool is_q = head < R.n_q_heads, is_k = !is_q && head < R.n_q_heads + R.n_kv_heads;
        const int kvh = head - R.n_q_heads - (is_k ? 0 : R.n_kv_heads);
        float bv = 0.f;
        if (R.bias_q) bv = is_q ? R.bias_q[head * hd + e] : is_k ? R.bias_k[kvh * hd + e] : R.bias_v[kvh * hd + e];
        int pos[NT][4], strm[NT][4];
        float cs[NT][4], sn[NT][4];
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int n = min(tok0 + t * 16 + lw * 4 + j, P.n_tokens - 1);
                pos[t][j] = R.pos[n];
                strm[t][j] = R.stream[n];
            }
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                cs[t][j] = R.cos[pos[t][j] * half + i];
                sn[t][j] = R.sin[pos[t][j] * half + i];
            }
        const int nq = R.n_q_heads * hd;
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int n = tok0 + t * 16 + lw * 4 + j;
                const float v = acc[0][t][j] + bv;
                const float partner = __shfl_xor(v, 8);
                float outv = v;
                if (is_q || is_k) {
                    const float c = cs[t][j], s1 = sn[t][j];
                    const float x0 = li < 8 ? v : partner, x1 = li < 8 ? partner : v;
                    if (!R.conj) outv = li < 8 ? (x0 * c - x1 * s1) : (x0 * s1 + x1 * c);
                    else outv = li < 8 ? (x0 * c + x1 * s1) : (-x0 * s1 + x1 * c);
                }
                if (n >= P.n_tokens) continue;
                if (is_q) R.q[(long long)n * nq + head * hd + e] = outv;
                else {
                    float *cache = is_k ? R.kcache : R.vcache;
                    cache[(long long)strm[t][j] * R.kv_stream_stride + ((long long)kvh * R.seq_len + pos[t][j]) * hd + e] = outv;
                }
            }
        return;
    }
    // D[token = (lane>>4)*4 + j][weight row = lane & 15].  The optional bias / residual operands are loaded for
    // the whole tile first (clamped addresses, no per-element branch -> one memory latency), then added.  A
    // workgroup whose 64 tokens all exist (every one but the last of a prompt) stores without per-element tests.
    const bool split = P.ksplit > 1, whole = tok0 + NT * 16 <= P.n_tokens;
    float *const dst = split ? partp + (long long)blockIdx.z * P.n_tokens * P.ldo : outp;
    const float *const resid = split ? nullptr : P.resid, *const bias = split ? nullptr : P.bias;
#pragma unroll
    for (int rt = 0; rt < RT; rt++) {
        const int row = (tile0 + rt) * TR + li;
        if (tile0 + rt >= P.ntiles || row >= P.rows) continue;
        const float bv = bias ? bias[row] : 0.f;
        unsigned off[NT][4];
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int j = 0; j < 4; j++)
                off[t][j] = (unsigned)min(tok0 + t * 16 + lw * 4 + j, P.n_tokens - 1) * (unsigned)P.ldo + (unsigned)row;
        float rv[NT][4];
        if (resid) {
#pragma unroll
            for (int t = 0; t < NT; t++)
#pragma unroll
                for (int j = 0; j < 4; j++) rv[t][j] = resid[off[t][j]];
        }
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float v = acc[rt][t][j];
                if (bias) v += bv;
                if (resid) v += rv[t][j];
                acc[rt][t][j] = v;
            }
        if (whole) {
#pragma unroll
            for (int t = 0; t < NT; t++)
#pragma unroll
                for (int j = 0; j < 4; j++) dst[off[t][j]] = acc[rt][t][j];
        } else {
#pragma unroll
            for (int t = 0; t < NT; t++)
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (tok0 + t * 16 + lw * 4 + j < P.n_tokens) dst[off[t][j]] = acc[rt][t][j];
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
