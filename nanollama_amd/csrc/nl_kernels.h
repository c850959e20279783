// nl_kernels.h -- gfx950 device kernels of the nanollama decode path.
//
// Everything here is HBM/L2-streaming VALU work (decode GEMV over block-quantised
// weights, single-token GQA attention); there is deliberately no MFMA.
//
// Weight layout ("row-interleaved tiles", built once at upload by repack_kernel):
//   a tile = TR(16) consecutive output rows; a pair = two 32-element quant
//   blocks (64 columns).  For tile t, pair p, 16-byte chunk c, row-in-tile r:
//       quants : ((t*npairs + p)*CPP + c)*TR + r   (x 16 bytes)
//       scales : (t*npairs + p)*TR + r             (x 4 bytes: two fp16 d)
//   CPP = chunks per pair = 4 (Q8_0), 2 (Q4_0), 8 (F16), 16 (F32).
//   A wavefront owns one tile; lane = (kl, r) with r = lane & 15 the row and
//   kl = lane >> 4 one of 4 "k-lanes" walking the pairs.  Every global load
//   instruction therefore reads four 256-byte runs, each lane accumulates whole
//   blocks of its own row (dot over 32 in-register, then * d, as
//   go/quant.go:149-165 / :74-94 do per block), and the only cross-lane work
//   is two shuffles per tile.  The fp16 scale bits are kept verbatim.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

namespace nl {

constexpr int TR = 16;           // rows per tile
constexpr int KL = 4;            // k-lanes per wavefront (64 / TR)
constexpr int PAIR = 64;         // columns per pair
constexpr int ATT_CH = 128;      // positions per attention split
constexpr int ATT_THREADS = 256;
constexpr int CTL_TOKEN = 0, CTL_POS = 1, CTL_CHAIN = 2, CTL_STEP = 3, CTL_STREAM = 4, CTL_WORDS = 8;

enum { WT_F32 = 0, WT_F16 = 1, WT_Q4_0 = 2, WT_Q8_0 = 8 };
enum { PRO_PLAIN = 0, PRO_NORM = 1, PRO_ATTN = 2 };
enum { EPI_STORE = 0, EPI_RESID = 1, EPI_SWIGLU = 2, EPI_QKV = 3 };
enum { ROWMAP_IDENT = 0, ROWMAP_HEADPERM = 1 };

template <int WT> struct WTraits;
template <> struct WTraits<WT_Q8_0> { static constexpr int CPP = 4; static constexpr bool SCALED = true; };
template <> struct WTraits<WT_Q4_0> { static constexpr int CPP = 2; static constexpr bool SCALED = true; };
template <> struct WTraits<WT_F16> { static constexpr int CPP = 8; static constexpr bool SCALED = false; };
template <> struct WTraits<WT_F32> { static constexpr int CPP = 16; static constexpr bool SCALED = false; };

__device__ __forceinline__ float h2f_bits(uint32_t h) {
    // exact IEEE binary16 -> binary32 (subnormals kept), == go/gguf.go:603-636
    return __half2float(__ushort_as_half((unsigned short)h));
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

__global__ void repack_kernel(RepackParams P) {
    const int cpp = P.wtype == WT_Q8_0 ? 4 : P.wtype == WT_Q4_0 ? 2 : P.wtype == WT_F16 ? 8 : 16;
    const long long nchunks = (long long)P.ntiles * P.npairs * cpp * TR;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < nchunks;
         idx += (long long)gridDim.x * blockDim.x) {
        int r = (int)(idx % TR);
        long long t1 = idx / TR;
        int c = (int)(t1 % cpp);
        long long t2 = t1 / cpp;
        int p = (int)(t2 % P.npairs);
        int tile = (int)(t2 / P.npairs);
        int row = map_row(P.rowmap, P.head_dim, tile, r);
        uint4 v = make_uint4(0, 0, 0, 0);
        uint8_t *vb = reinterpret_cast<uint8_t *>(&v);
        if (row < P.nrows) {
            long long srow = (long long)(P.row0 + row);
            if (P.wtype == WT_Q8_0) {
                int b = 2 * p + (c >> 1);
                if (b * 32 < P.ncols) {
                    const uint8_t *blk = P.src + (srow * (P.src_cols / 32) + (P.col0 / 32 + b)) * 34;
                    for (int k = 0; k < 16; k++) vb[k] = blk[2 + (c & 1) * 16 + k];
                }
            } else if (P.wtype == WT_Q4_0) {
                int b = 2 * p + c;
                if (b * 32 < P.ncols) {
                    const uint8_t *blk = P.src + (srow * (P.src_cols / 32) + (P.col0 / 32 + b)) * 18;
                    for (int k = 0; k < 16; k++) vb[k] = blk[2 + k];
                }
            } else {
                int esz = P.wtype == WT_F16 ? 2 : 4;
                int per = 16 / esz;
                int e0 = p * PAIR + c * per;
                const uint8_t *sp = P.src + (srow * P.src_cols + P.col0 + e0) * esz;
                for (int k = 0; k < per; k++)
                    if (e0 + k < P.ncols)
                        for (int bb = 0; bb < esz; bb++) vb[k * esz + bb] = sp[k * esz + bb];
            }
        }
        reinterpret_cast<uint4 *>(P.q)[idx] = v;
        if (P.s && c == 0) {
            uint32_t sc = 0;
            if (row < P.nrows) {
                long long srow = (long long)(P.row0 + row);
                int bsz = P.wtype == WT_Q8_0 ? 34 : 18;
                for (int hb = 0; hb < 2; hb++) {
                    int b = 2 * p + hb;
                    if (b * 32 < P.ncols) {
                        const uint8_t *blk = P.src + (srow * (P.src_cols / 32) + (P.col0 / 32 + b)) * bsz;
                        sc |= ((uint32_t)blk[0] | ((uint32_t)blk[1] << 8)) << (16 * hb);
                    }
                }
            }
            P.s[((long long)tile * P.npairs + p) * TR + r] = sc;
        }
    }
}

// ------------------------------------------------------------- pair dots ---

__device__ __forceinline__ float dot4_i8(uint32_t w, float4 x, float acc) {
    acc = fmaf((float)(int)(int8_t)(w & 0xff), x.x, acc);
    acc = fmaf((float)(int)(int8_t)((w >> 8) & 0xff), x.y, acc);
    acc = fmaf((float)(int)(int8_t)((w >> 16) & 0xff), x.z, acc);
    acc = fmaf((float)(int)(int8_t)(w >> 24), x.w, acc);
    return acc;
}

template <int WT> struct PairDot;

template <> struct PairDot<WT_Q8_0> {
    // out = sum_b d_b * sum_j q_bj x_j  (go/quant.go:149-165)
    static __device__ __forceinline__ float run(const uint4 *c, uint32_t sc, const float *xp, float acc) {
        const float4 *x4 = reinterpret_cast<const float4 *>(xp);
#pragma unroll
        for (int b = 0; b < 2; b++) {
            float dot = 0.f;
            uint4 lo = c[2 * b], hi = c[2 * b + 1];
            dot = dot4_i8(lo.x, x4[8 * b + 0], dot);
            dot = dot4_i8(lo.y, x4[8 * b + 1], dot);
            dot = dot4_i8(lo.z, x4[8 * b + 2], dot);
            dot = dot4_i8(lo.w, x4[8 * b + 3], dot);
            dot = dot4_i8(hi.x, x4[8 * b + 4], dot);
            dot = dot4_i8(hi.y, x4[8 * b + 5], dot);
            dot = dot4_i8(hi.z, x4[8 * b + 6], dot);
            dot = dot4_i8(hi.w, x4[8 * b + 7], dot);
            acc = fmaf(dot, h2f_bits((sc >> (16 * b)) & 0xffff), acc);
        }
        return acc;
    }
};

__device__ __forceinline__ float dot_q4_word(uint32_t w, float4 xl, float4 xh, float acc) {
    // byte k of w: low nibble = element k, high nibble = element k+16 (go/quant.go:84-88)
    uint32_t lo = w & 0x0F0F0F0Fu, hi = (w >> 4) & 0x0F0F0F0Fu;
    acc = fmaf((float)(int)(lo & 0xff) - 8.f, xl.x, acc);
    acc = fmaf((float)(int)(hi & 0xff) - 8.f, xh.x, acc);
    acc = fmaf((float)(int)((lo >> 8) & 0xff) - 8.f, xl.y, acc);
    acc = fmaf((float)(int)((hi >> 8) & 0xff) - 8.f, xh.y, acc);
    acc = fmaf((float)(int)((lo >> 16) & 0xff) - 8.f, xl.z, acc);
    acc = fmaf((float)(int)((hi >> 16) & 0xff) - 8.f, xh.z, acc);
    acc = fmaf((float)(int)(lo >> 24) - 8.f, xl.w, acc);
    acc = fmaf((float)(int)(hi >> 24) - 8.f, xh.w, acc);
    return acc;
}

template <> struct PairDot<WT_Q4_0> {
    static __device__ __forceinline__ float run(const uint4 *c, uint32_t sc, const float *xp, float acc) {
        const float4 *x4 = reinterpret_cast<const float4 *>(xp);
#pragma unroll
        for (int b = 0; b < 2; b++) {
            float dot = 0.f;
            uint4 q = c[b];
            dot = dot_q4_word(q.x, x4[8 * b + 0], x4[8 * b + 4], dot);
            dot = dot_q4_word(q.y, x4[8 * b + 1], x4[8 * b + 5], dot);
            dot = dot_q4_word(q.z, x4[8 * b + 2], x4[8 * b + 6], dot);
            dot = dot_q4_word(q.w, x4[8 * b + 3], x4[8 * b + 7], dot);
            acc = fmaf(dot, h2f_bits((sc >> (16 * b)) & 0xffff), acc);
        }
        return acc;
    }
};

template <> struct PairDot<WT_F16> {
    static __device__ __forceinline__ float run(const uint4 *c, uint32_t, const float *xp, float acc) {
        const float4 *x4 = reinterpret_cast<const float4 *>(xp);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            uint4 q = c[k];
            float4 a = x4[2 * k], b = x4[2 * k + 1];
            acc = fmaf(h2f_bits(q.x & 0xffff), a.x, acc);
            acc = fmaf(h2f_bits(q.x >> 16), a.y, acc);
            acc = fmaf(h2f_bits(q.y & 0xffff), a.z, acc);
            acc = fmaf(h2f_bits(q.y >> 16), a.w, acc);
            acc = fmaf(h2f_bits(q.z & 0xffff), b.x, acc);
            acc = fmaf(h2f_bits(q.z >> 16), b.y, acc);
            acc = fmaf(h2f_bits(q.w & 0xffff), b.z, acc);
            acc = fmaf(h2f_bits(q.w >> 16), b.w, acc);
        }
        return acc;
    }
};

template <> struct PairDot<WT_F32> {
    static __device__ __forceinline__ float run(const uint4 *c, uint32_t, const float *xp, float acc) {
        const float4 *x4 = reinterpret_cast<const float4 *>(xp);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            uint4 q = c[k];
            float4 a = x4[k];
            acc = fmaf(__uint_as_float(q.x), a.x, acc);
            acc = fmaf(__uint_as_float(q.y), a.y, acc);
            acc = fmaf(__uint_as_float(q.z), a.z, acc);
            acc = fmaf(__uint_as_float(q.w), a.w, acc);
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
    // prologue: input vector
    const float *x;      // PRO_PLAIN / PRO_NORM input [cols]
    const float *add;    // optional addend (tensor-parallel: all-reduced partial of the previous block)
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
};

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <int PRO>
__device__ __forceinline__ void stage_x(const GemvParams &P, float *xs, double *dred, int padded) {
    const int tid = threadIdx.x, nt = blockDim.x;
    if (PRO == PRO_PLAIN) {
        for (int i = tid * 4; i < P.cols; i += nt * 4) {
            float4 v = *reinterpret_cast<const float4 *>(P.x + i);
            if (P.add) {
                float4 a = *reinterpret_cast<const float4 *>(P.add + i);
                v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
            }
            *reinterpret_cast<float4 *>(xs + i) = v;
        }
    } else if (PRO == PRO_NORM) {
        // RMSNormInto go/quant.go:597-607: float64 sum of squares, inv cast to f32, (x*inv)*w
        double ss = 0.0;
        for (int i = tid * 4; i < P.cols; i += nt * 4) {
            float4 v = *reinterpret_cast<const float4 *>(P.x + i);
            if (P.add) {
                float4 a = *reinterpret_cast<const float4 *>(P.add + i);
                v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
            }
            if (P.x_out && blockIdx.x == 0) *reinterpret_cast<float4 *>(P.x_out + i) = v;
            *reinterpret_cast<float4 *>(xs + i) = v;
            ss += (double)v.x * (double)v.x;
            ss += (double)v.y * (double)v.y;
            ss += (double)v.z * (double)v.z;
            ss += (double)v.w * (double)v.w;
        }
        ss = wave_sum_f64(ss);
        if ((tid & 63) == 0) dred[tid >> 6] = ss;
        __syncthreads();
        double tot = 0.0;
        for (int w = 0; w < (nt >> 6); w++) tot += dred[w];
        float inv = (float)(1.0 / sqrt(tot / (double)P.cols + (double)P.eps));
        for (int i = tid * 4; i < P.cols; i += nt * 4) {
            float4 v = *reinterpret_cast<float4 *>(xs + i);
            float4 w = *reinterpret_cast<const float4 *>(P.normw + i);
            v.x = (v.x * inv) * w.x;
            v.y = (v.y * inv) * w.y;
            v.z = (v.z * inv) * w.z;
            v.w = (v.w * inv) * w.w;
            *reinterpret_cast<float4 *>(xs + i) = v;
        }
    } else {
        // combine the position-split attention partials (online-softmax merge)
        const int pos = P.ctl[CTL_POS];
        const int ns = pos / ATT_CH + 1;
        const int hd = P.head_dim;
        for (int i = tid; i < P.cols; i += nt) {
            int h = i / hd, d = i - h * hd;
            const float *ml = P.part_ml + (long long)h * P.nsplit_max * 2;
            float M = ml[0];
            for (int c = 1; c < ns; c++) M = fmaxf(M, ml[2 * c]);
            float L = 0.f, o = 0.f;
            for (int c = 0; c < ns; c++) {
                float w = (float)exp((double)(ml[2 * c] - M));
                L += w * ml[2 * c + 1];
                o += w * P.part_o[((long long)h * P.nsplit_max + c) * hd + d];
            }
            xs[i] = o * (1.0f / L);
        }
    }
    for (int i = P.cols + tid; i < padded; i += nt) xs[i] = 0.f;
    __syncthreads();
}

template <int WT>
__device__ __forceinline__ void load_pair(const uint8_t *q, const uint32_t *s, long long pair_index, int r,
                                          uint4 *c, uint32_t &sc) {
    constexpr int CPP = WTraits<WT>::CPP;
    const uint4 *qp = reinterpret_cast<const uint4 *>(q) + pair_index * (CPP * TR) + r;
#pragma unroll
    for (int k = 0; k < CPP; k++) c[k] = qp[k * TR];
    sc = WTraits<WT>::SCALED ? s[pair_index * TR + r] : 0u;
}

// One wavefront = one 16-row tile x a 1/kw share of the columns; a workgroup
// holds tw tiles x kw column shares.  See the layout comment at the top.
template <int WT, int PRO, int EPI>
__global__ void __launch_bounds__(1024) gemv_kernel(GemvParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int CPP = WTraits<WT>::CPP;
    constexpr int NM = EPI == EPI_SWIGLU ? 2 : 1;
    const int padded = P.npairs * PAIR;
    float *xs = reinterpret_cast<float *>(smem);
    float *red = xs + padded;                         // [NM][waves][TR]
    const int nwaves = blockDim.x >> 6;
    double *dred = reinterpret_cast<double *>(red + NM * nwaves * TR);  // [waves]

    stage_x<PRO>(P, xs, dred, padded);

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & (TR - 1), kl = lane >> 4;
    const int tile = blockIdx.x * P.tw + wave / P.kw;
    const int kw = wave % P.kw;
    float acc0 = 0.f, acc1 = 0.f;
    if (tile < P.ntiles) {
        const long long tbase = (long long)tile * P.npairs;
        const int stride = P.kw * KL;
        int p = kw * KL + kl;
        // two pairs in flight per lane
        for (; p + stride < P.npairs; p += 2 * stride) {
            uint4 ca[CPP], cb[CPP], ga[NM > 1 ? CPP : 1], gb[NM > 1 ? CPP : 1];
            uint32_t sa, sb, ta = 0, tb = 0;
            load_pair<WT>(P.q0, P.s0, tbase + p, r, ca, sa);
            load_pair<WT>(P.q0, P.s0, tbase + p + stride, r, cb, sb);
            if (NM > 1) {
                load_pair<WT>(P.q1, P.s1, tbase + p, r, ga, ta);
                load_pair<WT>(P.q1, P.s1, tbase + p + stride, r, gb, tb);
            }
            acc0 = PairDot<WT>::run(ca, sa, xs + p * PAIR, acc0);
            if (NM > 1) acc1 = PairDot<WT>::run(ga, ta, xs + p * PAIR, acc1);
            acc0 = PairDot<WT>::run(cb, sb, xs + (p + stride) * PAIR, acc0);
            if (NM > 1) acc1 = PairDot<WT>::run(gb, tb, xs + (p + stride) * PAIR, acc1);
        }
        if (p < P.npairs) {
            uint4 ca[CPP], ga[NM > 1 ? CPP : 1];
            uint32_t sa, ta = 0;
            load_pair<WT>(P.q0, P.s0, tbase + p, r, ca, sa);
            if (NM > 1) load_pair<WT>(P.q1, P.s1, tbase + p, r, ga, ta);
            acc0 = PairDot<WT>::run(ca, sa, xs + p * PAIR, acc0);
            if (NM > 1) acc1 = PairDot<WT>::run(ga, ta, xs + p * PAIR, acc1);
        }
    }
    // k-lanes -> row sums (lanes 0..15 of each wave)
    acc0 += __shfl_xor(acc0, 16);
    acc0 += __shfl_xor(acc0, 32);
    if (NM > 1) {
        acc1 += __shfl_xor(acc1, 16);
        acc1 += __shfl_xor(acc1, 32);
    }
    if (lane < TR) {
        red[wave * TR + lane] = acc0;
        if (NM > 1) red[(nwaves + wave) * TR + lane] = acc1;
    }
    __syncthreads();

    const int t = threadIdx.x;
    if (t >= P.tw * TR) return;
    const int tin = t / TR, rr = t % TR;
    const int otile = blockIdx.x * P.tw + tin;
    if (otile >= P.ntiles) return;
    float v = 0.f, v1 = 0.f;
    for (int k = 0; k < P.kw; k++) {  // fixed order: deterministic
        v += red[(tin * P.kw + k) * TR + rr];
        if (NM > 1) v1 += red[(nwaves + tin * P.kw + k) * TR + rr];
    }
    if (EPI == EPI_QKV) {
        // RoPE (go/model.go:449-477), KV store (:552-554).  Tile rows 0-7 hold
        // element i, rows 8-15 element i + hd/2 of the same head (ROWMAP_HEADPERM).
        const int hd = P.head_dim, half = hd >> 1, tph = hd / 16;
        const int head = otile / tph, j = otile % tph;
        const int i = j * 8 + (rr & 7);
        const int e = i + (rr >> 3) * half;
        const int pos = P.ctl[CTL_POS];
        float partner = __shfl_xor(v, 8);
        float outv = v;
        if (head < P.n_q_heads + P.n_kv_heads) {
            float c = P.rope_cos[pos * half + i], s = P.rope_sin[pos * half + i];
            float x0 = (rr < 8) ? v : partner, x1 = (rr < 8) ? partner : v;
            if (!P.rope_conj) outv = (rr < 8) ? (x0 * c - x1 * s) : (x0 * s + x1 * c);
            else outv = (rr < 8) ? (x0 * c + x1 * s) : (-x0 * s + x1 * c);
        }
        if (head < P.n_q_heads) {
            P.qbuf[head * hd + e] = outv;
        } else {
            const long long soff = (long long)P.ctl[CTL_STREAM] * P.kv_stream_stride;
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
    if (row >= P.rows) return;
    if (EPI == EPI_STORE) {
        P.out[row] = v;
    } else if (EPI == EPI_RESID) {
        P.out[row] = P.resid[row] + v;
    } else if (EPI == EPI_SWIGLU) {
        // SiLU go/quant.go:629-631: x / (1 + f32(exp(f64(-x)))), then * up (go/model.go:604-606)
        float ex = (float)exp((double)(-v));
        P.out[row] = (v / (1.0f + ex)) * v1;
    }
}

// ------------------------------------------------------------- embedding ---

struct EmbedParams {
    const uint8_t *table;  // raw GGUF rows
    int wtype, dim;
    const int *ctl;
    float *x;
};

// embedLookupInto go/model.go:389-446 (row `token` of token_embd, dequantised)
__global__ void embed_kernel(EmbedParams P) {
    const int token = P.ctl[CTL_TOKEN];
    for (int i = threadIdx.x; i < P.dim; i += blockDim.x) {
        float v;
        if (P.wtype == WT_Q8_0) {
            const uint8_t *blk = P.table + ((long long)token * (P.dim / 32) + i / 32) * 34;
            float d = h2f_bits((uint32_t)blk[0] | ((uint32_t)blk[1] << 8));
            v = (float)(int)(int8_t)blk[2 + (i & 31)] * d;
        } else if (P.wtype == WT_Q4_0) {
            const uint8_t *blk = P.table + ((long long)token * (P.dim / 32) + i / 32) * 18;
            float d = h2f_bits((uint32_t)blk[0] | ((uint32_t)blk[1] << 8));
            int j = i & 31;
            int nib = j < 16 ? (blk[2 + j] & 0x0F) : (blk[2 + j - 16] >> 4);
            v = (float)(nib - 8) * d;
        } else if (P.wtype == WT_F16) {
            const uint8_t *p = P.table + ((long long)token * P.dim + i) * 2;
            v = h2f_bits((uint32_t)p[0] | ((uint32_t)p[1] << 8));
        } else {
            v = reinterpret_cast<const float *>(P.table)[(long long)token * P.dim + i];
        }
        P.x[i] = v;
    }
}

// ------------------------------------------------------------- attention ---

struct AttnParams {
    const float *qbuf;             // [n_q_heads][hd], RoPE applied
    const float *kcache, *vcache;  // this layer, stream 0: [kv][seq][hd]
    long long kv_stream_stride;
    float *part_o;                 // [n_q_heads][nsplit_max][hd]
    float *part_ml;                // [n_q_heads][nsplit_max][2]
    const int *ctl;
    int n_kv_heads, seq_len, nsplit_max;
    float scale;
};

// GQA decode attention for one token (go/model.go:557-587): one workgroup per
// (kv head, 128-position split); the G query heads of the group share every K
// and V element read.  Emits un-normalised partials (max, sum, sum p*v) that
// the WO GEMV prologue merges.
template <int HD, int G>
__global__ void __launch_bounds__(ATT_THREADS) attn_kernel(AttnParams P) {
    const int pos = P.ctl[CTL_POS];
    const int split = blockIdx.y, t0 = split * ATT_CH;
    if (t0 > pos) return;
    const int n = min(ATT_CH, pos + 1 - t0);
    const int kvh = blockIdx.x, tid = threadIdx.x;
    const long long soff = (long long)P.ctl[CTL_STREAM] * P.kv_stream_stride;
    const float *K = P.kcache + soff + ((long long)kvh * P.seq_len + t0) * HD;
    const float *V = P.vcache + soff + ((long long)kvh * P.seq_len + t0) * HD;

    constexpr int KS = HD + 1;  // padded row stride: conflict-free column walks
    __shared__ float Kt[ATT_CH * KS];
    __shared__ float qs[G * HD];
    __shared__ float sc[G * ATT_CH];
    __shared__ float ored[(ATT_THREADS / HD) * G * HD];
    __shared__ float ml[G * 2];

    for (int i = tid; i < G * HD; i += ATT_THREADS) qs[i] = P.qbuf[kvh * G * HD + i];
    for (int i = tid; i < n * (HD / 4); i += ATT_THREADS) {
        int row = i / (HD / 4), c4 = i % (HD / 4);
        float4 v = *reinterpret_cast<const float4 *>(K + row * HD + c4 * 4);
        float *dst = Kt + row * KS + c4 * 4;
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    }
    __syncthreads();

    // scores: thread (t, g) -> dot over d in the reference's order
    for (int i = tid; i < n * G; i += ATT_THREADS) {
        int t = i % n, g = i / n;
        const float *kr = Kt + t * KS, *qr = qs + g * HD;
        float dot = 0.f;
#pragma unroll 8
        for (int d = 0; d < HD; d++) dot += qr[d] * kr[d];
        sc[g * ATT_CH + t] = dot * P.scale;
    }
    __syncthreads();

    // softmax pieces per head: wave w handles heads w, w+4, ...
    const int wave = tid >> 6, lane = tid & 63;
    for (int g = wave; g < G; g += ATT_THREADS / 64) {
        float m = -INFINITY;
        for (int t = lane; t < n; t += 64) m = fmaxf(m, sc[g * ATT_CH + t]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        float l = 0.f;
        for (int t = lane; t < n; t += 64) {
            float p = (float)exp((double)(sc[g * ATT_CH + t] - m));  // go/quant.go:619
            sc[g * ATT_CH + t] = p;
            l += p;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) l += __shfl_xor(l, o);
        if (lane == 0) { ml[2 * g] = m; ml[2 * g + 1] = l; }
    }
    __syncthreads();

    // P*V: thread = (position group, d); V rows are read coalesced from global
    constexpr int NG = ATT_THREADS / HD;
    const int d = tid % HD, tg = tid / HD;
    float o[G];
#pragma unroll
    for (int g = 0; g < G; g++) o[g] = 0.f;
    for (int t = tg; t < n; t += NG) {
        float v = V[t * HD + d];
#pragma unroll
        for (int g = 0; g < G; g++) o[g] = fmaf(sc[g * ATT_CH + t], v, o[g]);
    }
#pragma unroll
    for (int g = 0; g < G; g++) ored[(tg * G + g) * HD + d] = o[g];
    __syncthreads();
    for (int i = tid; i < G * HD; i += ATT_THREADS) {
        int g = i / HD, dd = i % HD;
        float s = 0.f;
        for (int k = 0; k < NG; k++) s += ored[(k * G + g) * HD + dd];
        int h = kvh * G + g;
        P.part_o[((long long)h * P.nsplit_max + split) * HD + dd] = s;
        if (dd == 0) {
            P.part_ml[((long long)h * P.nsplit_max + split) * 2] = ml[2 * g];
            P.part_ml[((long long)h * P.nsplit_max + split) * 2 + 1] = ml[2 * g + 1];
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
    const float *logits;
    int n;
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
    for (int i = tid; i < P.n; i += blockDim.x) {
        float v = P.logits[i];
        if (v > best || idx == 0x7fffffff) { best = v; idx = i; }
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

// sum of tensor-parallel partial vectors living on ONE device (single-process
// emulation of the all-reduce, used by tests): dst[r][i] = sum_k src[k][i]
__global__ void local_allreduce_kernel(float *const *bufs, int nranks, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int k = 0; k < nranks; k++) s += bufs[k][i];
    for (int k = 0; k < nranks; k++) bufs[k][i] = s;
}

}  // namespace nl
