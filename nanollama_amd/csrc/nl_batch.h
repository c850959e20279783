// nl_batch.h -- element-wise kernels of the multi-token step (batched decode streams / prompt prefill).
// The heavy lifting is qgemm_kernel (nl_qgemm.h) and the per-token attention kernel (nl_kernels.h, batched
// over blockIdx.z); everything here is one workgroup per token.
#pragma once
#include "nl_qgemm.h"

namespace nl {

// (struct GemmOut: nl_kernels.h, beside AttnParams)

__device__ __forceinline__ float gemm_out_at(const GemmOut &g, long long idx, int col) {
    if (g.ks <= 1) return g.val[idx];
    // slabs are fetched four at a time (clamped index, no per-load branch) so the adds wait for ks/4 memory
    // round trips, not ks; slabs past ks contribute +0.0f
    float v = 0.f;
    for (int z0 = 0; z0 < g.ks; z0 += 4) {
        float p[4];
#pragma unroll
        for (int k = 0; k < 4; k++) p[k] = g.part[(long long)min(z0 + k, g.ks - 1) * g.zstride + idx];
#pragma unroll
        for (int k = 0; k < 4; k++) v += z0 + k < g.ks ? p[k] : 0.f;
    }
    if (g.bias) v += g.bias[col];
    return v;
}

// two elements at once (RoPE: a value and its rotation partner), slabs fetched EIGHT at a time for both, so a
// 16-way split costs two memory round trips instead of eight; per element the same additions in the same order
__device__ __forceinline__ void gemm_out_at2(const GemmOut &g, long long ia, int ca, long long ib, int cb, float &a, float &b) {
    if (g.ks <= 1) { a = g.val[ia]; b = g.val[ib]; return; }
    a = 0.f; b = 0.f;
    for (int z0 = 0; z0 < g.ks; z0 += 8) {
        float pa[8], pb[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const long long zo = (long long)min(z0 + k, g.ks - 1) * g.zstride;
            pa[k] = g.part[zo + ia]; pb[k] = g.part[zo + ib];
        }
#pragma unroll
        for (int k = 0; k < 8; k++) { a += z0 + k < g.ks ? pa[k] : 0.f; b += z0 + k < g.ks ? pb[k] : 0.f; }
    }
    if (g.bias) { a += g.bias[ca]; b += g.bias[cb]; }
}

// RoPE prologue of attn_kernel<HD, G, FIN> (decode batches whose Q|K|V GEMM ran split-K): the rows of this workgroup's
// kv group -- G query heads, one K and one V head of one token -- are summed from the GEMM's slabs, biased, rotated
// (go/model.go:449-477, :525-527) and handed to the attention through LDS; the K / V rows also go to the cache
// (go/model.go:552-554).  What brope_kv_kernel does per token, done per (token, kv head) by the consumer: one launch less
// per layer.  Element e of a head sits at packed row (head * hd/16 + (e % half) / 8) * 16 + (e % half) % 8 + 8 * (e / half)
// (ROWMAP_HEADPERM), its rotation partner at that row ^ 8.
template <int HD, int G>
__device__ void attn_rope_prologue(const AttnParams &P, int kvh, int item, int pos, long long soff, float *qs, float *krow, float *vcur, bool kv_part) {
    constexpr int half = HD / 2, tph = HD / 16;
    constexpr int NIT = ((G + 2) * HD + ATT_THREADS - 1) / ATT_THREADS;   // elements per thread
    const AttnParams::Rope &R = P.rp;
    const long long src0 = (long long)item * R.R;
    const int count = (kv_part ? G + 2 : G) * HD;
    // Every element's operands are requested before any is used: the stores below may alias the loads of a later element as
    // far as hipcc can tell, so written as one loop the second element's round trip started after the first one's stores
    // (stamps: 7-8k of this launch's 15k cycles were entry -> prologue done).  Same arithmetic in the same order.
    float v[NIT], partner[NIT], rc[NIT], rs[NIT];
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        if (it * ATT_THREADS >= count) { v[it] = partner[it] = rc[it] = rs[it] = 0.f; continue; }   // (uniform: a q-only split has fewer elements)
        const int i = min((int)threadIdx.x + it * ATT_THREADS, count - 1);
        const int hs = i / HD, e = i % HD, ih = e % half, hi = e / half;
        const int head = hs < G ? kvh * G + hs : hs == G ? R.n_q_heads + kvh : R.n_q_heads + P.n_kv_heads + kvh;
        const int rho = (head * tph + ih / 8) * 16 + (ih % 8) + 8 * hi;
        rc[it] = R.cos[pos * half + ih]; rs[it] = R.sin[pos * half + ih];
        gemm_out_at2(R.qkv, src0 + rho, rho, src0 + (rho ^ 8), rho ^ 8, v[it], partner[it]);
        if (R.bias_q) {
            const float *bp = hs < G ? R.bias_q + (kvh * G + hs) * HD : hs == G ? R.bias_k + kvh * HD : R.bias_v + kvh * HD;
            v[it] += bp[e];
            partner[it] += bp[e ^ half];
        }
    }
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        const int i = (int)threadIdx.x + it * ATT_THREADS;
        if (i >= count) continue;
        const int hs = i / HD, e = i % HD, hi = e / half;
        float outv = v[it];
        if (hs <= G) {
            const float x0 = hi == 0 ? v[it] : partner[it], x1 = hi == 0 ? partner[it] : v[it];
            if (!R.conj) outv = hi == 0 ? (x0 * rc[it] - x1 * rs[it]) : (x0 * rs[it] + x1 * rc[it]);
            else outv = hi == 0 ? (x0 * rc[it] + x1 * rs[it]) : (-x0 * rs[it] + x1 * rc[it]);
        }
        if (hs < G) qs[hs * HD + e] = outv;
        else {
            (hs == G ? krow : vcur)[e] = outv;
            (hs == G ? R.kcache_w : R.vcache_w)[soff + ((long long)kvh * P.seq_len + pos) * HD + e] = outv;
        }
    }
}

// two float4 groups at once (the a / b halves of an 8-slot unit), slabs fetched eight at a time for both
__device__ __forceinline__ void gemm_out_at4x2(const GemmOut &g, long long ia, int ca, long long ib, int cb, float4 &a, float4 &b) {
    if (g.ks <= 1) { a = *reinterpret_cast<const float4 *>(g.val + ia); b = *reinterpret_cast<const float4 *>(g.val + ib); return; }
    a = make_float4(0.f, 0.f, 0.f, 0.f); b = a;
    for (int z0 = 0; z0 < g.ks; z0 += 8) {
        float4 pa[8], pb[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const long long zo = (long long)min(z0 + k, g.ks - 1) * g.zstride;
            pa[k] = *reinterpret_cast<const float4 *>(g.part + zo + ia);
            pb[k] = *reinterpret_cast<const float4 *>(g.part + zo + ib);
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const bool on = z0 + k < g.ks;
            a.x += on ? pa[k].x : 0.f; a.y += on ? pa[k].y : 0.f; a.z += on ? pa[k].z : 0.f; a.w += on ? pa[k].w : 0.f;
            b.x += on ? pb[k].x : 0.f; b.y += on ? pb[k].y : 0.f; b.z += on ? pb[k].z : 0.f; b.w += on ? pb[k].w : 0.f;
        }
    }
    if (g.bias) {
        const float4 ba = *reinterpret_cast<const float4 *>(g.bias + ca), bb = *reinterpret_cast<const float4 *>(g.bias + cb);
        a.x += ba.x; a.y += ba.y; a.z += ba.z; a.w += ba.w;
        b.x += bb.x; b.y += bb.y; b.z += bb.z; b.w += bb.w;
    }
}

// gate and up of the same two float4 groups: the slabs of BOTH matrices are requested before either is summed (four
// slabs of each per round trip), same additions in the same order as gemm_out_at4x2 per matrix
__device__ __forceinline__ void gemm_out_pair4x2(const GemmOut &g, const GemmOut &u, long long ia, long long ib,
                                                 float4 &ga, float4 &gb, float4 &ua, float4 &ub) {
    ga = make_float4(0.f, 0.f, 0.f, 0.f); gb = ga; ua = ga; ub = ga;
    for (int z0 = 0; z0 < g.ks; z0 += 4) {
        float4 pga[4], pgb[4], pua[4], pub[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const long long zo = (long long)min(z0 + k, g.ks - 1) * g.zstride;
            pga[k] = *reinterpret_cast<const float4 *>(g.part + zo + ia);
            pgb[k] = *reinterpret_cast<const float4 *>(g.part + zo + ib);
            pua[k] = *reinterpret_cast<const float4 *>(u.part + zo + ia);
            pub[k] = *reinterpret_cast<const float4 *>(u.part + zo + ib);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const bool on = z0 + k < g.ks;
            ga.x += on ? pga[k].x : 0.f; ga.y += on ? pga[k].y : 0.f; ga.z += on ? pga[k].z : 0.f; ga.w += on ? pga[k].w : 0.f;
            gb.x += on ? pgb[k].x : 0.f; gb.y += on ? pgb[k].y : 0.f; gb.z += on ? pgb[k].z : 0.f; gb.w += on ? pgb[k].w : 0.f;
            ua.x += on ? pua[k].x : 0.f; ua.y += on ? pua[k].y : 0.f; ua.z += on ? pua[k].z : 0.f; ua.w += on ? pua[k].w : 0.f;
            ub.x += on ? pub[k].x : 0.f; ub.y += on ? pub[k].y : 0.f; ub.z += on ? pub[k].z : 0.f; ub.w += on ? pub[k].w : 0.f;
        }
    }
}

// four consecutive columns (idx, col multiples of 4): same arithmetic per element as gemm_out_at
__device__ __forceinline__ float4 gemm_out_at4(const GemmOut &g, long long idx, int col) {
    if (g.ks <= 1) return *reinterpret_cast<const float4 *>(g.val + idx);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z0 = 0; z0 < g.ks; z0 += 4) {
        float4 p[4];
#pragma unroll
        for (int k = 0; k < 4; k++) p[k] = *reinterpret_cast<const float4 *>(g.part + (long long)min(z0 + k, g.ks - 1) * g.zstride + idx);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const bool on = z0 + k < g.ks;
            v.x += on ? p[k].x : 0.f; v.y += on ? p[k].y : 0.f; v.z += on ? p[k].z : 0.f; v.w += on ? p[k].w : 0.f;
        }
    }
    if (g.bias) {
        const float4 b = *reinterpret_cast<const float4 *>(g.bias + col);
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    }
    return v;
}

struct BEmbedParams {
    const uint8_t *table;
    int wtype, dim;
    const int *tokens;
    float *x;  // [N][dim]
    const int *gamma_row;
    const float *gamma_val;
    // optional (decode batches / short prompts on dgemm_kernel): the token's RoPE rows and cache row offset, QGemmParams::Rope::tcos ...
    const int *pos, *stream;
    const float *rope_cos, *rope_sin;
    float *tcos, *tsin;
    long long *tkv;
    int half, hd;
    long long kv_stream_stride;
};

__global__ void bembed_kernel(BEmbedParams P) {
    const int token = sload_i32(P.tokens + blockIdx.x);   // (per-workgroup words of the step's metadata: scalar cache, no vector wait)
    float *x = P.x + (long long)blockIdx.x * P.dim;
    if (P.tcos) {
        const int pos = sload_i32(P.pos + blockIdx.x), strm = sload_i32(P.stream + blockIdx.x);
        if ((int)threadIdx.x < P.half) {
            P.tcos[(long long)blockIdx.x * P.half + threadIdx.x] = P.rope_cos[pos * P.half + threadIdx.x];
            P.tsin[(long long)blockIdx.x * P.half + threadIdx.x] = P.rope_sin[pos * P.half + threadIdx.x];
        }
        if (threadIdx.x == 0) P.tkv[blockIdx.x] = (long long)strm * P.kv_stream_stride + (long long)pos * P.hd;
    }
    const int gr = P.gamma_row ? P.gamma_row[token] : -1;
    for (int i = threadIdx.x; i < P.dim; i += blockDim.x) {
        float v = embed_value(P.table, P.wtype, P.dim, token, i);
        if (gr >= 0) v += P.gamma_val[(long long)gr * P.dim + i];
        x[i] = v;
    }
}

// RMSNormInto go/quant.go:597-607, one workgroup per token, fused on both sides:
//   in : the residual stream x, or (pend.ks > 1) x + the pending split-K slabs of the GEMM that feeds it
//        (attention output / down projection) -- the sum is written back to x, which stays the residual
//   out: the normalised row as fp16 hi/lo MFMA fragments for the next GEMM (no f32 copy, no split pass)
struct BNormParams {
    float *x;            // [N][dim] residual stream (updated in place when pend.ks > 1)
    GemmOut pend;        // pending GEMM output to fold in (val unused: with ks <= 1 the GEMM already wrote x)
    const float *w;
    float eps;
    int dim, item0;      // token of workgroup b = item0 + b; its fragment column is b
    uint4 *xf;
    int nt16, q4;
    float *scale_out;    // [N] or nullptr: norm_prescale(inv) of the token, the pre-scale of the first folded-norm producer
};

// UPT = 8-slot units per thread (host: dim / 8 <= UPT * blockDim.x).  A thread's units stay in registers
// from the sum of squares to the fragment store: one trip to memory per row, 16-byte loads.
template <int UPT>
__global__ void __launch_bounds__(256) bnorm_kernel(BNormParams P) {
    NL_KARGS8(P.x, P.w, P.xf, P.pend.part, P.pend.bias, P.dim, P.item0, P.pend.ks);   // one batch of s_load (nl_kernels.h)
    NL_KARGS4(P.eps, P.nt16, P.q4, P.pend.zstride);
    __shared__ double dred[4];
    const int item = P.item0 + blockIdx.x, n = P.dim, nu = n / 8;
    float *xr = P.x + (long long)item * n;
    const bool fold = P.pend.ks > 1;
    float4 xa[UPT], xb[UPT];
    int ca[UPT], cb[UPT];
#pragma unroll
    for (int k = 0; k < UPT; k++) {
        const int u = min((int)threadIdx.x + k * (int)blockDim.x, nu - 1);
        int offa, offb;
        slot_offsets(P.q4, u & 3, offa, offb);
        ca[k] = (u >> 2) * 32 + offa; cb[k] = (u >> 2) * 32 + offb;
        xa[k] = *reinterpret_cast<const float4 *>(xr + ca[k]);
        xb[k] = *reinterpret_cast<const float4 *>(xr + cb[k]);
    }
    float4 wa[UPT], wb[UPT];     // (requested with x: not behind the slab round trip of the fold)
#pragma unroll
    for (int k = 0; k < UPT; k++) {
        wa[k] = *reinterpret_cast<const float4 *>(P.w + ca[k]);
        wb[k] = *reinterpret_cast<const float4 *>(P.w + cb[k]);
    }
    if (fold) {
#pragma unroll
        for (int k = 0; k < UPT; k++) {
            float4 pa, pb;
            gemm_out_at4x2(P.pend, (long long)item * n + ca[k], ca[k], (long long)item * n + cb[k], cb[k], pa, pb);
            xa[k] = make_float4(pa.x + xa[k].x, pa.y + xa[k].y, pa.z + xa[k].z, pa.w + xa[k].w);
            xb[k] = make_float4(pb.x + xb[k].x, pb.y + xb[k].y, pb.z + xb[k].z, pb.w + xb[k].w);
            if ((int)threadIdx.x + k * (int)blockDim.x < nu) {
                *reinterpret_cast<float4 *>(xr + ca[k]) = xa[k];
                *reinterpret_cast<float4 *>(xr + cb[k]) = xb[k];
            }
        }
    }
    double ss = 0.0;
#pragma unroll
    for (int k = 0; k < UPT; k++) {
        double t = 0.0;
        t = fma((double)xa[k].x, (double)xa[k].x, t); t = fma((double)xa[k].y, (double)xa[k].y, t);
        t = fma((double)xa[k].z, (double)xa[k].z, t); t = fma((double)xa[k].w, (double)xa[k].w, t);
        t = fma((double)xb[k].x, (double)xb[k].x, t); t = fma((double)xb[k].y, (double)xb[k].y, t);
        t = fma((double)xb[k].z, (double)xb[k].z, t); t = fma((double)xb[k].w, (double)xb[k].w, t);
        ss += (int)threadIdx.x + k * (int)blockDim.x < nu ? t : 0.0;
    }
    ss = wave_sum_f64(ss);
    if ((threadIdx.x & 63) == 0) dred[threadIdx.x >> 6] = ss;
    __syncthreads();
    double tot = 0.0;
    for (int k = 0; k < (int)(blockDim.x >> 6); k++) tot += dred[k];
    const float inv = (float)(1.0 / sqrt(tot / (double)n + (double)P.eps));
    if (P.scale_out && threadIdx.x == 0) P.scale_out[item] = norm_prescale(inv);
#pragma unroll
    for (int k = 0; k < UPT; k++) {
        const int u = (int)threadIdx.x + k * (int)blockDim.x;
        if (u >= nu) continue;
        float xv[8], wv[8], v[8];
        slots_from(P.q4, xa[k], xb[k], xv);
        slots_from(P.q4, wa[k], wb[k], wv);
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = (xv[j] * inv) * wv[j];
        store_frag(P.xf, P.nt16, blockIdx.x, u >> 2, u & 3, v);
    }
}

// any dim: strided passes over the row (the row is re-read after the reduction)
__global__ void __launch_bounds__(256) bnorm_generic_kernel(BNormParams P) {
    __shared__ double dred[4];
    const int item = P.item0 + blockIdx.x, n = P.dim;
    float *xr = P.x + (long long)item * n;
    double ss = 0.0;
    if (P.pend.ks > 1) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const float v = gemm_out_at(P.pend, (long long)item * n + i, i) + xr[i];
            xr[i] = v;
            ss += (double)v * (double)v;
        }
    } else {
        for (int i = threadIdx.x; i < n; i += blockDim.x) ss += (double)xr[i] * (double)xr[i];
    }
    ss = wave_sum_f64(ss);
    if ((threadIdx.x & 63) == 0) dred[threadIdx.x >> 6] = ss;
    __syncthreads();   // also orders the x write-back above before the re-read below (same workgroup)
    double tot = 0.0;
    for (int k = 0; k < (int)(blockDim.x >> 6); k++) tot += dred[k];
    const float inv = (float)(1.0 / sqrt(tot / (double)n + (double)P.eps));
    if (P.scale_out && threadIdx.x == 0) P.scale_out[item] = norm_prescale(inv);
    for (int u = threadIdx.x; u < n / 8; u += blockDim.x) {
        const int blk = u >> 2, w = u & 3;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int i = blk * 32 + slot_elem(P.q4, w, j);
            v[j] = (xr[i] * inv) * P.w[i];
        }
        store_frag(P.xf, P.nt16, blockIdx.x, blk, w, v);
    }
}

struct BRopeParams {
    GemmOut qkv;            // [N][R], rows in the packed (ROWMAP_HEADPERM) tile order of the QKV matrix
    int R, head_dim, n_q_heads, n_kv_heads, seq_len, rope_conj, qk_norm;
    float eps;
    const int *pos, *stream;
    const float *rope_cos, *rope_sin;
    float *q;               // [N][n_q_heads*hd] natural order
    float *kcache, *vcache; // this layer, stream 0
    long long kv_stream_stride;
    const float *bias_q, *bias_k, *bias_v;  // optional, natural (head, element) order
};

// RoPE (go/model.go:449-477) + optional QK-norm (:542-549) + KV store (:552-554) for one token per workgroup
__global__ void brope_kv_kernel(BRopeParams P) {
    NL_KARGS8(P.qkv.val, P.qkv.part, P.qkv.bias, P.pos, P.stream, P.rope_cos, P.rope_sin, P.q);   // one batch of s_load
    NL_KARGS8(P.kcache, P.vcache, P.bias_q, P.kv_stream_stride, P.R, P.head_dim, P.n_q_heads, P.n_kv_heads);
    NL_KARGS8(P.seq_len, P.rope_conj, P.qk_norm, P.eps, P.qkv.ks, P.qkv.zstride, P.bias_k, P.bias_v);
    extern __shared__ float vals[];  // [R] in natural (head, element) order after RoPE
    const int item = blockIdx.x, hd = P.head_dim, half = hd >> 1;
    const int hsh = hd == 64 ? 6 : 5, tsh = hsh - 4;     // head_dim is 32 or 64 (nl_create): shifts, not runtime divisions
    const int pos = sload_i32(P.pos + item);
    const long long src0 = (long long)item * P.R;
    for (int rho = threadIdx.x; rho < P.R; rho += blockDim.x) {
        const int tile = rho / TR, r = rho % TR;
        const int head = tile >> tsh, j = tile & ((1 << tsh) - 1);
        const int i = j * 8 + (r & 7), e = i + (r >> 3) * half;
        // the rotation's cos / sin leave with the slab requests (V rows load them too and ignore them): one round trip
        const float rc = P.rope_cos[pos * half + i], rs = P.rope_sin[pos * half + i];
        float v, partner;
        gemm_out_at2(P.qkv, src0 + rho, rho, src0 + (rho ^ 8), rho ^ 8, v, partner);
        float outv = v;
        if (P.bias_q) {
            const int ep = i + ((r ^ 8) >> 3) * half;   // the partner's element index
            if (head < P.n_q_heads) { v += P.bias_q[head * hd + e]; partner += P.bias_q[head * hd + ep]; }
            else if (head < P.n_q_heads + P.n_kv_heads) {
                v += P.bias_k[(head - P.n_q_heads) * hd + e]; partner += P.bias_k[(head - P.n_q_heads) * hd + ep];
            } else v += P.bias_v[(head - P.n_q_heads - P.n_kv_heads) * hd + e];
            outv = v;
        }
        if (head < P.n_q_heads + P.n_kv_heads) {
            const float c = rc, s = rs;
            float x0 = (r < 8) ? v : partner, x1 = (r < 8) ? partner : v;
            if (!P.rope_conj) outv = (r < 8) ? (x0 * c - x1 * s) : (x0 * s + x1 * c);
            else outv = (r < 8) ? (x0 * c + x1 * s) : (-x0 * s + x1 * c);
        }
        vals[head * hd + e] = outv;
    }
    __syncthreads();
    if (P.qk_norm) {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
        for (int head = wave; head < P.n_q_heads + P.n_kv_heads; head += nw) {
            float *vec = vals + head * hd;
            double ss = 0.0;
            for (int i = lane; i < hd; i += 64) ss += (double)vec[i] * (double)vec[i];
            ss = wave_sum_f64(ss);
            float inv = (float)(1.0 / sqrt(ss / (double)hd + (double)P.eps));
            for (int i = lane; i < hd; i += 64) vec[i] = vec[i] * inv;
        }
        __syncthreads();
    }
    const long long soff = (long long)sload_i32(P.stream + item) * P.kv_stream_stride;
    const int nq = P.n_q_heads * hd, nk = P.n_kv_heads * hd;
    for (int i = threadIdx.x; i < P.R; i += blockDim.x) {
        float v = vals[i];
        if (i < nq) {
            P.q[(long long)item * nq + i] = v;
        } else if (i < nq + nk) {
            int kvh = (i - nq) >> hsh, e = (i - nq) & (hd - 1);
            P.kcache[soff + ((long long)kvh * P.seq_len + pos) * hd + e] = v;
        } else {
            int kvh = (i - nq - nk) >> hsh, e = (i - nq - nk) & (hd - 1);
            P.vcache[soff + ((long long)kvh * P.seq_len + pos) * hd + e] = v;
        }
    }
}

// FIN epilogue of attn_kernel (nl_kernels.h): the only split of every row is complete, so x = o / l -- the ns == 1
// arithmetic of battn_merge_kernel -- goes straight into the WO GEMM's fragments.  One thread per 8 k-slots.
template <int HD, int G>
__device__ void attn_finalize(const AttnParams &P, const float *ored, const float *ml, int kvh, int item) {
    constexpr int NG = ATT_THREADS / (HD / 4);
    const int tid = threadIdx.x;
    if (tid >= G * HD / 8) return;
    const int g = tid / (HD / 8), u = tid % (HD / 8), bl = u >> 2, w = u & 3;
    const float scale = 1.0f / ml[2 * g + 1];
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int d = bl * 32 + slot_elem(P.fin_q4, w, j);
        float s = 0.f;
#pragma unroll 8
        for (int k = 0; k < NG; k++) s += ored[(k * G + g) * HD + d];
        v[j] = s * scale;
    }
    store_frag(P.fin_xf, P.fin_nt16, item, ((kvh * G + g) * HD) / 32 + bl, w, v);
}

struct BMergeParams {
    const float *part_o, *part_ml;  // [N][heads][nsplit_max][hd] / [..][2]
    const int *pos;
    const int *nparts;   // partials per item when attn_tile16_kernel folded runs of chunks (null: one per 128 positions)
    int heads, nsplit_max, head_dim;
    uint4 *xf;   // attention output [N][heads*hd] as MFMA fragments for the WO GEMM
    int nt16, q4;
    int fast_exp;   // split weights on v_exp_f32 (prompts: their partials come from attn_tile16_kernel's v_exp_f32 already)
};

// online-softmax merge of the position splits (same arithmetic and summation order as the decode GEMV's
// PRO_ATTN prologue); one thread per 8 k-slots of the output row.  The (max, sum) pairs of all splits and the
// partial rows of four splits at a time are fetched with clamped indices before anything consumes them
// (a runtime-bounded "for c < ns: load" loop is ns dependent round trips); splits past ns get weight 0.
template <int MAXS>   // most partials an item can have: 16 (seq_len <= 2048, go/model.go:145-148, one per 128 keys), or 4 when the
                      // prompt attention kernel folded runs of chunks (12 of the 16 (max, sum) loads and weights were masked work)
__global__ void battn_merge_kernel(BMergeParams P, int n_items) {
    NL_KARGS8(P.part_o, P.part_ml, P.pos, P.xf, P.heads, P.nsplit_max, P.head_dim, P.nt16);   // one batch of s_load
    const int hd = P.head_dim, upi = P.heads * hd / 8;
    const long long total = (long long)n_items * upi;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int item = (int)(idx / upi), u = (int)(idx - (long long)item * upi);
        const int ns = min(P.nparts ? P.nparts[item] : P.pos[item] / ATT_CH + 1, MAXS);
        const long long pbase = (long long)item * P.heads * P.nsplit_max;
        const int blk = u >> 2, w = u & 3;
        const int h = blk * 32 / hd;                 // a 32-column block never straddles heads (hd = 32 or 64)
        const float2 *ml = reinterpret_cast<const float2 *>(P.part_ml) + (pbase + (long long)h * P.nsplit_max);
        float2 mlv[MAXS];
#pragma unroll
        for (int c = 0; c < MAXS; c++) mlv[c] = ml[min(c, ns - 1)];
        float M = mlv[0].x;
#pragma unroll
        for (int c = 1; c < MAXS; c++) M = c < ns ? fmaxf(M, mlv[c].x) : M;
        float wt[MAXS], L = 0.f;
#pragma unroll
        for (int c = 0; c < MAXS; c++) {
            wt[c] = 0.f;
            // (the float64 exp of go/quant.go:619 is ~150 VALU instructions, redone by the 8 threads of a head: 12 of this
            //  kernel's 16 us on a 2047-token prompt)
            if (c < ns) wt[c] = P.fast_exp ? __builtin_amdgcn_exp2f((mlv[c].x - M) * 1.44269504088896340736f) : exp_f64_as_f32(mlv[c].x - M);
            L += wt[c] * mlv[c].y;
        }
        const float scale = 1.0f / L;
        const float *po = P.part_o + (pbase + (long long)h * P.nsplit_max) * hd + (blk * 32 - h * hd);
        int offa, offb;
        slot_offsets(P.q4, w, offa, offb);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = 0.f;
        for (int c0 = 0; c0 < ns; c0 += 4) {
            float pv[4][8];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                // slots (w, j) of the block, as load_x_slots: only the two middle lanes of each float4 differ by type
                const float *src = po + (long long)min(c0 + k, ns - 1) * hd;
                const float4 a = *reinterpret_cast<const float4 *>(src + offa), b = *reinterpret_cast<const float4 *>(src + offb);
                slots_from(P.q4, a, b, pv[k]);
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                float wk = 0.f;
#pragma unroll
                for (int c = 0; c < MAXS; c++) wk = (c == c0 + k) ? wt[c] : wk;
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] += wk * pv[k][j];
            }
        }
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] *= scale;
        store_frag(P.xf, P.nt16, item, blk, w, v);
    }
}

// SiLU(gate) * up, go/quant.go:629-631 + go/model.go:604-606; one thread per (token, block, slot group);
// gate / up may still be split-K slabs; h leaves as MFMA fragments for the down projection
struct BSwigluParams {
    GemmOut g, u;        // [N][interm]
    int interm, n_tokens;
    uint4 *xf;
    int nt16, q4;
};

__global__ void bswiglu_kernel(BSwigluParams P) {
    NL_KARGS8(P.g.val, P.g.part, P.u.val, P.u.part, P.xf, P.interm, P.n_tokens, P.g.ks);   // one batch of s_load
    NL_KARGS4(P.g.zstride, P.u.zstride, P.nt16, P.q4);
    const int upt = P.interm / 8;                     // units per token
    const long long total = (long long)P.n_tokens * upt;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i / upt), u = (int)(i - (long long)n * upt), blk = u >> 2, w = u & 3;
        int offa, offb;
        slot_offsets(P.q4, w, offa, offb);
        const int ca = blk * 32 + offa, cb = blk * 32 + offb;
        const long long row = (long long)n * P.interm;
        float4 ga, gb, ua, ub;
        if (P.g.ks <= 1 && P.u.ks <= 1) {   // one branch around all four loads: they are in flight together
            ga = *reinterpret_cast<const float4 *>(P.g.val + row + ca); gb = *reinterpret_cast<const float4 *>(P.g.val + row + cb);
            ua = *reinterpret_cast<const float4 *>(P.u.val + row + ca); ub = *reinterpret_cast<const float4 *>(P.u.val + row + cb);
        } else if (P.g.ks == P.u.ks && P.g.ks > 1 && !P.g.bias && !P.u.bias && P.g.zstride == P.u.zstride) {
            gemm_out_pair4x2(P.g, P.u, row + ca, row + cb, ga, gb, ua, ub);   // (the gate || up launch: same split for both)
        } else {
            gemm_out_at4x2(P.g, row + ca, ca, row + cb, cb, ga, gb);
            gemm_out_at4x2(P.u, row + ca, ca, row + cb, cb, ua, ub);
        }
        float gv[8], uv[8], v[8];
        slots_from(P.q4, ga, gb, gv);
        slots_from(P.q4, ua, ub, uv);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float ex = exp_f64_as_f32(-gv[j]);
            v[j] = (gv[j] / (1.0f + ex)) * uv[j];
        }
        store_frag(P.xf, P.nt16, n, blk, w, v);
    }
}

// argmax per token (go/main.go:400-408), one workgroup per token
__global__ void __launch_bounds__(1024) bargmax_kernel(const float *logits, int n, int *ids) {
    __shared__ float bv[16];
    __shared__ int bi[16];
    const float *lg = logits + (long long)blockIdx.x * n;
    const int tid = threadIdx.x;
    float best = -INFINITY;
    int idx = 0x7fffffff;
    // eight logits per thread and memory round trip (clamped index, masked at the compare; same ascending order per thread):
    // the one-load-per-iteration form was 47 dependent round trips per workgroup (20 us for 64 x 48000 logits)
    for (int i0 = tid; i0 < n; i0 += 8 * (int)blockDim.x) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = lg[min(i0 + k * (int)blockDim.x, n - 1)];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int i = i0 + k * (int)blockDim.x;
            if (i < n && (v[k] > best || idx == 0x7fffffff)) { best = v[k]; idx = i; }
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
        ids[blockIdx.x] = idx == 0x7fffffff ? 0 : idx;
    }
}

}  // namespace nl

namespace nl {

// Prefill attention: the tokens of a multi-token step that belong to ONE stream at consecutive positions.
// One workgroup per (kv head, 128-position split, tile of QT query tokens).  The QT x G (token, query head)
// pairs of the tile are the rows of two fp32 GEMMs on the matrix cores (v_mfma_f32_16x16x4_f32, full fp32
// products and accumulation -- no reduced-precision inputs):
//     S[row][key] = q[row] . K[key]          (rows x 128 keys, reduction over head_dim)
//     O[row][d]   = sum_key P[row][key] V[key][d]
// Each wavefront owns 16 rows.  The split's K and V rows are staged in LDS once per workgroup and shared by
// all rows; causality is each row's own position (nv[row] = number of keys of this split it may see).
// MFMA operand layout (A[i][k]: lane = 16k+i, B[k][j]: lane = 16k+j, D[i][j]: lane = 16(i/4)+j, vgpr i%4)
// leaves the order of the reduction index free, so lane group k owns a CONTIGUOUS slice of head_dim (QK) or
// 4 consecutive keys (PV) and every LDS read is a 16-byte ds_read_b128; the output tile's column index is
// permuted (d = NTO*j + nt) so each lane stores NTO consecutive floats.
// The first product is computed transposed (S^T = K q^T) so that its D layout IS the A layout P needs in the
// second: P never leaves registers and there is one barrier.  The partial (max, sum, sum p*v) layout
// is the decode kernel's, so battn_merge_kernel is unchanged.  exp is float32 here (the decode kernel follows
// go/quant.go:619 with a float64 exp; the difference is ~1e-7 relative and this path has ~N^2/2 of them).
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_f32<DPP_QUAD_XOR1>(v));
    v = fmaxf(v, dpp_f32<DPP_QUAD_XOR2>(v));
    v = fmaxf(v, dpp_f32<DPP_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_f32<DPP_ROW_MIRROR>(v));
    return v;
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f32<DPP_QUAD_XOR1>(v);
    v += dpp_f32<DPP_QUAD_XOR2>(v);
    v += dpp_f32<DPP_HALF_MIRROR>(v);
    v += dpp_f32<DPP_ROW_MIRROR>(v);
    return v;
}

template <int HD, int G, int QT>
__global__ void __launch_bounds__(QT * G * 4, 2) attn_tile_kernel(AttnParams P, int n_items) {
    constexpr int VH = QT * G;               // rows (token, query head) per workgroup
    static_assert(VH % 16 == 0 && VH <= 64, "16 rows per wavefront");
    constexpr int NTH = VH * 4;              // threads: one wavefront per 16 rows
    constexpr int KS = HD + 4;               // K row stride in LDS (floats): b128 reads of 8 rows hit 32 banks
    constexpr int DK = HD / 4;               // head_dim slice of one lane group in the QK reduction
    constexpr int NTO = HD / 16;             // output column tiles
    constexpr int R4 = HD / 4;
    __shared__ __attribute__((aligned(16))) float Ks[ATT_CH * KS];
    __shared__ __attribute__((aligned(16))) float Vs[ATT_CH * HD];
    __shared__ int nv[VH];

    const int kvh = blockIdx.x, split = blockIdx.y, i0 = blockIdx.z * QT, tid = threadIdx.x;
    const int t0 = split * ATT_CH;
    int maxpos = -1;
    for (int i = 0; i < QT; i++)
        if (i0 + i < n_items) maxpos = max(maxpos, P.bpos[i0 + i]);
    if (t0 > maxpos) return;
    const int nrows = min(ATT_CH, maxpos + 1 - t0);
    const int lane = tid & 63, w = tid >> 6, j = lane & 15, kq = lane >> 4;

    // A operand of QK: row w*16+j, head_dim slice [DK*kq, DK*kq+DK)
    float qa[DK];
    {
        const int vh = w * 16 + j, item = i0 + vh / G, g = vh % G;
        if (item < n_items) {
            const float4 *q4 = reinterpret_cast<const float4 *>(
                P.qbuf + (long long)item * P.q_item_stride + (kvh * G + g) * HD + DK * kq);
#pragma unroll
            for (int s = 0; s < DK / 4; s++) {
                float4 t = q4[s];
                qa[4 * s] = t.x; qa[4 * s + 1] = t.y; qa[4 * s + 2] = t.z; qa[4 * s + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int s = 0; s < DK; s++) qa[s] = 0.f;
        }
    }
    const long long soff = (long long)P.bstream[i0] * P.kv_stream_stride;
    const float4 *K4 = reinterpret_cast<const float4 *>(P.kcache + soff + ((long long)kvh * P.seq_len + t0) * HD);
    const float4 *V4 = reinterpret_cast<const float4 *>(P.vcache + soff + ((long long)kvh * P.seq_len + t0) * HD);
    // every global load of the staging pass is issued before the first LDS store (one memory latency, not NIT);
    // all 8 key tiles are always computed (uniform, branch-free MFMA stream); rows beyond nrows are zero-filled
    // (masked to p = 0 by the softmax, and 0 * garbage could be NaN in P V otherwise)
    constexpr int NIT = (ATT_CH * R4 + NTH - 1) / NTH;
    float4 kreg[NIT], vreg[NIT];
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        const int i = tid + it * NTH;
        kreg[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        vreg[it] = kreg[it];
        if (i / R4 < nrows) { kreg[it] = K4[i]; vreg[it] = V4[i]; }
    }
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        const int i = tid + it * NTH, row = i / R4, c4 = i % R4;
        if (i < ATT_CH * R4) {
            *reinterpret_cast<float4 *>(Ks + row * KS + c4 * 4) = kreg[it];
            *reinterpret_cast<float4 *>(Vs + row * HD + c4 * 4) = vreg[it];
        }
    }
    if (tid < VH) {
        const int item = i0 + tid / G;
        nv[tid] = item < n_items ? min(ATT_CH, max(0, P.bpos[item] + 1 - t0)) : 0;
    }
    __syncthreads();

    // ---- S^T = K q^T: 8 key tiles (M) x 16 rows (N); lane (j, kq) ends up holding row j's scores for keys
    //      16*mt + 4*kq + r -- exactly the A-operand layout of P in P V, so P never leaves registers ----
    v4f acc[8];
#pragma unroll
    for (int mt = 0; mt < 8; mt++) acc[mt] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s4 = 0; s4 < DK / 4; s4++) {
        float4 kb[8];
#pragma unroll
        for (int mt = 0; mt < 8; mt++)
            kb[mt] = *reinterpret_cast<const float4 *>(Ks + (mt * 16 + j) * KS + DK * kq + 4 * s4);
#pragma unroll
        for (int mt = 0; mt < 8; mt++)
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kb[mt].x, qa[4 * s4], acc[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < 8; mt++)
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kb[mt].y, qa[4 * s4 + 1], acc[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < 8; mt++)
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kb[mt].z, qa[4 * s4 + 2], acc[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < 8; mt++)
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kb[mt].w, qa[4 * s4 + 3], acc[mt], 0, 0, 0);
    }

    // ---- softmax pieces of row j: 32 keys in this lane, the rest in lanes j+16, j+32, j+48 ----
    const int nvj = nv[w * 16 + j];
    float m = -INFINITY;
#pragma unroll
    for (int mt = 0; mt < 8; mt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float sv = mt * 16 + 4 * kq + r < nvj ? acc[mt][r] * P.scale : -INFINITY;
            acc[mt][r] = sv;
            m = fmaxf(m, sv);
        }
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    float l = 0.f;
#pragma unroll
    for (int mt = 0; mt < 8; mt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float pv = mt * 16 + 4 * kq + r < nvj ? expf(acc[mt][r] - m) : 0.f;
            acc[mt][r] = pv;
            l += pv;
        }
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);

    // ---- O = P V: A = P[row j][key 16s+4kq+i] = acc[s][i], B = V rows from LDS ----
    v4f o[NTO];
#pragma unroll
    for (int nt = 0; nt < NTO; nt++) o[nt] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 8; s++) {
        {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float *vr = Vs + (16 * s + 4 * kq + i) * HD + NTO * j;
                if constexpr (NTO == 4) {
                    const float4 vb = *reinterpret_cast<const float4 *>(vr);
                    o[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(acc[s][i], vb.x, o[0], 0, 0, 0);
                    o[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(acc[s][i], vb.y, o[1], 0, 0, 0);
                    o[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(acc[s][i], vb.z, o[2], 0, 0, 0);
                    o[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(acc[s][i], vb.w, o[3], 0, 0, 0);
                } else {
                    const float2 vb = *reinterpret_cast<const float2 *>(vr);
                    o[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(acc[s][i], vb.x, o[0], 0, 0, 0);
                    o[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(acc[s][i], vb.y, o[1], 0, 0, 0);
                }
            }
        }
    }

    // O tile: this lane holds rows 4*kq+r, columns d = NTO*j .. NTO*j+NTO-1
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int vh = w * 16 + 4 * kq + r;
        if (nv[vh] == 0) continue;
        const int item = i0 + vh / G, h = kvh * G + vh % G;
        const long long slot = (long long)item * P.part_item_stride + (long long)h * P.nsplit_max + split;
        float *po = P.part_o + slot * HD + NTO * j;
        if constexpr (NTO == 4) *reinterpret_cast<float4 *>(po) = make_float4(o[0][r], o[1][r], o[2][r], o[3][r]);
        else *reinterpret_cast<float2 *>(po) = make_float2(o[0][r], o[1][r]);
    }
    if (kq == 0 && nvj > 0) {
        const int vh = w * 16 + j, item = i0 + vh / G, h = kvh * G + vh % G;
        const long long slot = (long long)item * P.part_item_stride + (long long)h * P.nsplit_max + split;
        P.part_ml[slot * 2] = m;
        P.part_ml[slot * 2 + 1] = l;
    }
}

// ---- the same tile on the fp16 matrix cores -------------------------------------------------------------
// v_mfma_f32_16x16x4_f32 runs at the f32 VECTOR rate (1/16 of the fp16 MFMA rate); the f32 kernel above spends
// 256 of them (8192 SIMD cycles) per wavefront.  Here every f32 operand is split x = hi + lo into two fp16 values
// (split_hi_lo, ~2^-22 relative; the lo x lo term is dropped) and each product becomes three
// v_mfma_f32_16x16x32_f16 (hi*hi + lo*hi + hi*lo, f32 accumulation): 96 MFMAs of 16 cycles per wavefront.
//   * K is split into Kh / Kl[key][hd] fp16 (A operand of S^T = K q^T: a lane reads 8 consecutive head_dim
//     elements of its key with one ds_read_b128).
//   * V is split AND transposed: VTh / VTl[d][key position] -- the 16-bit MFMA wants a lane to hold 8 reduction
//     indices (keys) of one head_dim element d.  The key -> position map inside each 32-key group puts the 8 keys a
//     lane owns side by side (key = 32m + 16h + 4kg + e  ->  position 32m + 8kg + 4h + e), which is exactly how the
//     S^T accumulators leave P in registers (lane (row, kg) holds keys 16*mt + 4kg + e of key tile mt): P is converted
//     in place, never moved, and V^T is read with one ds_read_b128 per operand.
//   * Both are XOR-swizzled 16-byte segments of unpadded rows (Kv16Image): conflict-free under ds_read_b128's lane groups.
//   * A workgroup folds a RUN of chunks with the online softmax in registers: log2 domain, two elements per instruction;
//     p * 2^10 = hi (a mask) + lo, the running output as O^T = V^T P^T so that a query row stays in one lane column.
//   * From 256 tokens on (SHADOW) the images are built once per layer by kv16_build_kernel and a workgroup takes 256 rows.
// Partials (max, sum, sum p*v) leave in the decode kernel's layout, one per run, so battn_merge_kernel is shared.
// Measured (mini, 2047 tokens, per layer): 52 us attention + 15 us merge at the start of round 3 (one chunk per
// workgroup, every workgroup converting its chunk) -> 31 + 6 (images) + 7 us.  What is left: the matrix, vector and LDS
// pipes of a SIMD take turns -- all the wavefronts between two chunk barriers are in the same phase -- at ~38 % MFMA
// busy (profiles/r03_prompt_attention.txt has the stamps, the census and the SQ counters of every step).
#ifdef NL_ATT_STAMPS
// developer build (tools/att_stamps.sh): phase stamps of one workgroup + a census of every workgroup of the last launch
// (entry / exit on the 100 MHz wall clock, HW_ID and XCC_ID registers)
__device__ long long g_att_stamps[64];
__device__ long long g_att_census[4 * 8192];
__device__ long long g_att_timeline[512 * 64];   // the same stamps of every workgroup (linear id < 512), wall clock
#define ATT_STAMP(i) do { \
    if (kvh == 1 && c_first == 0 && i0 == NL_ATT_STAMPS * QT && threadIdx.x == 64 && (i) < 64) g_att_stamps[(i)] = clock64(); \
    if (threadIdx.x == 64 && (i) < 64 && blockIdx.y == 0 && blockIdx.z == 0 && blockIdx.x < 512) g_att_timeline[blockIdx.x * 64 + (i)] = clock64(); \
    if (((i) == 0 || (i) == 8) && threadIdx.x == 0) { \
        const int wg_ = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x; \
        if (wg_ < 8192) { \
            g_att_census[4 * wg_ + ((i) ? 1 : 0)] = wall_clock64(); \
            if ((i) == 0) { g_att_census[4 * wg_ + 2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); \
                            g_att_census[4 * wg_ + 3] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) | ((long long)__builtin_amdgcn_s_getreg(6 | (0 << 6) | (31 << 11)) << 8); } \
        } } } while (0)
#else
#define ATT_STAMP(i) do { } while (0)
#endif
// LDS image of one 128-key chunk of one kv head, in halves: Kh | Kl [key][HD], then VTh | VTl [d][128].  Rows are not
// padded; the 16-byte segments of a row are XOR-swizzled so that the kernel's ds_read_b128 are conflict-free.  A wave64
// ds_read_b128 is serviced in four groups of 16 lanes -- {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32
// (MI355X_MICROARCH.md, LDS) -- i.e. with lane = j + 16*kq every group holds each j once, half of them with segment s and
// half with s ^ 1; the padded layouts of round 2 (rows of HD + 8 / 128 + 8 halves) put 7 of those 16 lanes on a busy
// bank and every read took 8 LDS cycles instead of 4 (SQ_LDS_BANK_CONFLICT = 4 per ds_read_b128, measured).
template <int HD> struct Kv16Image {
    static constexpr int KS = HD;                   // K row: HD halves = HD/8 segments
    static constexpr int VS = ATT_CH;               // V^T row: 128 halves = 16 segments = one 256-byte bank row
    static constexpr int K_HALVES = 2 * ATT_CH * KS, V_HALVES = 2 * HD * VS;
    static constexpr int K_BYTES = K_HALVES * 2, V_BYTES = V_HALVES * 2, BYTES = K_BYTES + V_BYTES;
    static_assert(K_BYTES % 1024 == 0 && V_BYTES % 1024 == 0, "whole 1 KB LDS-DMA instructions");
    // offset (halves) of segment `seg` (8 halves) of K row `row`: HD = 64: two rows per bank row, rows r and r + 2 differ
    // by one segment; HD = 32: four rows per bank row
    static __device__ __forceinline__ int k_off(int row, int seg) {
        const int sw = HD == 64 ? ((row >> 1) & 7) : ((0 - (row >> 2)) & 3);
        return row * KS + ((seg ^ sw) << 3);
    }
    // offset (halves) of segment `seg` (8 key positions) of V^T row d
    static __device__ __forceinline__ int v_off(int d, int seg) { return d * VS + ((seg ^ (d & 15)) << 3); }
};

// f32 cache rows of one chunk -> the image (split_hi_lo, V transposed with the key -> position map of the kernel below).
// Thread (key pair rp, float4 column c4) owns keys 2rp, 2rp+1 (clamped loads, zero beyond nrows so that masked
// probabilities meet finite values); all loads are issued before the first LDS store.
template <int HD, int NTH>
__device__ __forceinline__ void kv16_stage(const float4 *K4, const float4 *V4, int nrows, int tid,
                                           _Float16 *Kh, _Float16 *Kl, _Float16 *VTh, _Float16 *VTl, bool sync_before_store) {
    typedef Kv16Image<HD> Img;
    constexpr int R4 = HD / 4;
    constexpr int NIT = (ATT_CH / 2 * R4 + NTH - 1) / NTH;
    float4 kreg[NIT][2], vreg[NIT][2];
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        const int i = min(tid + it * NTH, ATT_CH / 2 * R4 - 1), rp = i / R4, c4 = i % R4;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int row = min(2 * rp + h, nrows - 1);
            kreg[it][h] = K4[row * R4 + c4];
            vreg[it][h] = V4[row * R4 + c4];
        }
    }
    // (hipcc otherwise sinks the second half of these loads below the first half's LDS stores: two dependent memory
    // round trips per chunk instead of one)
    __builtin_amdgcn_sched_barrier(0);
    if (sync_before_store) __syncthreads();            // every wavefront is done with the previous chunk's K / V^T
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        const int i = tid + it * NTH, rp = i / R4, c4 = i % R4;
        if (i >= ATT_CH / 2 * R4) continue;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const bool live = 2 * rp + h < nrows;
            const float kv[4] = {kreg[it][h].x, kreg[it][h].y, kreg[it][h].z, kreg[it][h].w};
            _Float16 hh[4], ll[4];
#pragma unroll
            for (int e = 0; e < 4; e++) split_hi_lo(live ? kv[e] : 0.f, hh[e], ll[e]);
            typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
            const int ko = Img::k_off(2 * rp + h, c4 >> 1) + (c4 & 1) * 4;
            *reinterpret_cast<h4_t *>(Kh + ko) = h4_t{hh[0], hh[1], hh[2], hh[3]};
            *reinterpret_cast<h4_t *>(Kl + ko) = h4_t{ll[0], ll[1], ll[2], ll[3]};
        }
        // V^T: keys 2rp, 2rp+1 sit side by side at position 32m + 8kg + 4h + e (e even)
        const int r = 2 * rp, pos = (r & ~31) + 8 * ((r >> 2) & 3) + 4 * ((r >> 4) & 1) + (r & 3);
        const float v0[4] = {vreg[it][0].x, vreg[it][0].y, vreg[it][0].z, vreg[it][0].w};
        const float v1[4] = {vreg[it][1].x, vreg[it][1].y, vreg[it][1].z, vreg[it][1].w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
            _Float16 h0, l0, h1, l1;
            split_hi_lo(r < nrows ? v0[e] : 0.f, h0, l0);
            split_hi_lo(r + 1 < nrows ? v1[e] : 0.f, h1, l1);
            const int vo = Img::v_off(c4 * 4 + e, pos >> 3) + (pos & 7);
            *reinterpret_cast<h2_t *>(VTh + vo) = h2_t{h0, h1};
            *reinterpret_cast<h2_t *>(VTl + vo) = h2_t{l0, l1};
        }
    }
}

// prompts of many query tiles: every chunk is converted ONCE per layer (a tile-kernel workgroup converting its own chunks
// repeats the work of every other tile that reads them -- ~34x at 2047 tokens -- and the conversion was ~half of a
// workgroup's life).  grid (kv head, chunk); the image leaves LDS with coalesced 16-byte stores.
struct Kv16BuildParams {
    const float *kcache, *vcache;   // this layer, the prompt's stream: [kv][seq][hd]
    uint4 *kv16;                    // [kv][nsplit_max] images
    int seq_len, nsplit_max, n_keys;
};
template <int HD>
__global__ void __launch_bounds__(512) kv16_build_kernel(Kv16BuildParams P) {
    typedef Kv16Image<HD> Img;
    __shared__ __attribute__((aligned(16))) _Float16 img[Img::K_HALVES + Img::V_HALVES];
    const int kvh = blockIdx.x, chunk = blockIdx.y, tid = threadIdx.x, t0 = chunk * ATT_CH;
    const int nrows = min(ATT_CH, P.n_keys - t0);
    const float4 *K4 = reinterpret_cast<const float4 *>(P.kcache + ((long long)kvh * P.seq_len + t0) * HD);
    const float4 *V4 = reinterpret_cast<const float4 *>(P.vcache + ((long long)kvh * P.seq_len + t0) * HD);
    kv16_stage<HD, 512>(K4, V4, nrows, tid, img, img + ATT_CH * Img::KS, img + Img::K_HALVES, img + Img::K_HALVES + HD * Img::VS, false);
    __syncthreads();
    uint4 *dst = P.kv16 + ((long long)kvh * P.nsplit_max + chunk) * (Img::BYTES / 16);
    const uint4 *src = reinterpret_cast<const uint4 *>(img);
    for (int i = tid; i < Img::BYTES / 16; i += 512) dst[i] = src[i];
}

// max / sum over the four lanes j, j + 16, j + 32, j + 48 (one per row of 16), result in all four: gfx950's
// v_permlane16_swap (odd rows of the first operand <-> even rows of the second) and v_permlane32_swap (upper half <->
// lower half) are vector-pipe instructions -- __shfl_xor(v, 16 / 32) is a ds_bpermute round trip each, and a wavefront
// of the kernel below had eight of them in a row per chunk.
__device__ __forceinline__ float rows4_max(float v) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float rows4_sum(float v) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

template <int HD, int G, int QT, bool SHADOW>
__global__ void __launch_bounds__(QT * G * 4, QT * G >= 128 ? 4 : 2) attn_tile16_kernel(AttnParams P, int n_items) {
    constexpr int VH = QT * G;               // rows (token, query head) per workgroup
    static_assert(VH % 16 == 0 && VH <= 256, "16 rows per wavefront");
    constexpr int NTH = VH * 4, NW = NTH / 64;
    typedef Kv16Image<HD> Img;
    constexpr int KS = Img::KS, VS = Img::VS;
    constexpr int NKS = HD / 32;             // 32-wide reduction steps of q.k
    constexpr int NTO = HD / 16;             // output column tiles
    static_assert(ATT_CH == 128, "key position map below assumes 4 groups of 32 keys");
    constexpr int NBUF = SHADOW ? 2 : 1;     // SHADOW: the next chunk's images travel while this chunk is multiplied
    __shared__ __attribute__((aligned(16))) _Float16 Kimg[NBUF][Img::K_HALVES], Vimg[NBUF][Img::V_HALVES];
    __shared__ int npos[VH];                 // keys visible to each row: its position + 1 (0: no such item)

    // a workgroup = one query tile x a RUN of consecutive 128-key chunks, folded with the online softmax in registers;
    // its (max, sum, sum p*v) leaves as partial `slot` of the tile's items.  Prompts come with the host's list of runs
    // (AttnParams::live_map, one entry per blockIdx.x: kv head << 24 | tile << 16 | slot << 12 | first chunk << 6 | chunks);
    // without it the grid is (kv head, chunk, tile) with one chunk per workgroup.
    const int tid = threadIdx.x;
    const int code = P.live_map ? sload_i32(P.live_map + blockIdx.x) : 0;
    const int kvh = P.live_map ? (int)((unsigned)code >> 24) : (int)blockIdx.x;
    const int i0 = (P.live_map ? ((code >> 16) & 255) : (int)blockIdx.z) * QT;
    const int split = P.live_map ? ((code >> 12) & 15) : (int)blockIdx.y;           // partial slot
    const int c_first = P.live_map ? ((code >> 6) & 63) : (int)blockIdx.y;
    int c_count = P.live_map ? (code & 63) : 1;
    int maxpos = -1, minpos = 0x7fffffff;
    if (P.pos_base_valid) {   // a prompt: consecutive positions, nothing to read
        minpos = P.pos_base + i0;
        maxpos = P.pos_base + min(i0 + QT, n_items) - 1;
    } else {
        for (int i = 0; i < QT; i++) {
            const int pp = i0 + i < n_items ? P.bpos[i0 + i] : -1;
            maxpos = max(maxpos, pp);
            minpos = min(minpos, pp);
        }
    }
    if (c_first * ATT_CH > maxpos) return;
    c_count = min(c_count, maxpos / ATT_CH + 1 - c_first);
    const int lane = tid & 63, w = tid >> 6, j = lane & 15, kq = lane >> 4;
    ATT_STAMP(0);
    // SHADOW: a chunk's images, built by kv16_build_kernel, go straight into LDS (global_load_lds_dwordx4: 1 KB per
    // instruction, 64 per chunk) one chunk ahead of the arithmetic.  What bounds this kernel is the rate at which a CU
    // pulls those images out of the L2 -- 11-13 B/clk (MI355X_MICROARCH.md, the prologue-burst row; measured here as
    // ~5k cycles per 64 KB chunk however the requests were placed) against ~3k cycles of MFMA work per 128 rows -- so a
    // workgroup takes 256 rows (16 wavefronts, one workgroup per CU) per staged chunk and double-buffers the images.
    // The DMA is issued from inline asm (cdna_hip_programming.md 5.7): hipcc counts a builtin LDS-DMA and drains it
    // (vmcnt(0)) in front of the next LDS read it cannot prove disjoint, i.e. right away; these it does not see, and the
    // one wait per chunk is written out.  Source = scalar base + one VGPR of lane offsets: a per-lane 64-bit address kept
    // across the loop is spilled by hipcc and reloaded from scratch behind every barrier.
    [[maybe_unused]] auto dma_chunk = [&](int chunk, int buf) {
        const char *sbase = reinterpret_cast<const char *>(P.kv16) + ((long long)kvh * P.nsplit_max + chunk) * Img::BYTES;
        const int wv = __builtin_amdgcn_readfirstlane(w);
        const unsigned kdst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)&Kimg[buf][0]);
        const unsigned vdst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)&Vimg[buf][0]);
        const unsigned voff = (unsigned)lane * 16u;
        constexpr int NPK = Img::K_BYTES / 1024, NPV = Img::V_BYTES / 1024;
        for (int i = wv; i < NPK + NPV; i += NW) {
            unsigned keep;
            const unsigned dst = i < NPK ? kdst + i * 1024 : vdst + (i - NPK) * 1024;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff), "s"(dst), "s"(sbase + i * 1024) : "memory");
        }
    };
    [[maybe_unused]] auto dma_landed = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
    if constexpr (SHADOW) dma_chunk(c_first, 0);

    // B operand of S^T: row w*16+j, head_dim elements 32*ks + 8*kq .. +7, as hi / lo halves
    half8_t qh[NKS], ql[NKS];
    {
        const int vh = w * 16 + j, item = min(i0 + vh / G, n_items - 1), g = vh % G;
        const float *qp = P.qbuf + (long long)item * P.q_item_stride + (kvh * G + g) * HD + 8 * kq;
#pragma unroll
        for (int ks = 0; ks < NKS; ks++) {
            const float4 a = *reinterpret_cast<const float4 *>(qp + 32 * ks), b = *reinterpret_cast<const float4 *>(qp + 32 * ks + 4);
            const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
            for (int e = 0; e < 8; e++) {
                _Float16 h, l;
                split_hi_lo(v[e], h, l);
                qh[ks][e] = h; ql[ks][e] = l;
            }
        }
    }
    if (tid < VH) {
        const int item = i0 + tid / G;
        npos[tid] = item < n_items ? P.bpos[item] + 1 : 0;
    }
    const long long soff = P.single_stream ? 0 : (long long)sload_i32(P.bstream + i0) * P.kv_stream_stride;
    constexpr float LOG2E = 1.44269504088896340736f, LN2 = 0.69314718055994530942f;
    typedef float v2f __attribute__((ext_vector_type(2)));
    const float c2 = P.scale * LOG2E;
    float m = -INFINITY, l = 0.f;            // running max (log2 domain) / sum * 2^10 of row j (the same value in its four kq lanes)
    v4f o[NTO];                              // running sum p*v * 2^10 of row j: head_dim elements nt*16 + 4*kq + r
#pragma unroll
    for (int nt = 0; nt < NTO; nt++) o[nt] = (v4f){0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
    for (int c = 0; c < c_count; c++) {
        const int t0 = (c_first + c) * ATT_CH;
        const int nrows = min(ATT_CH, maxpos + 1 - t0);
        const bool full = minpos >= t0 + ATT_CH - 1;   // every row of the tile sees every key of this chunk
        _Float16 *const Kh = Kimg[c & (NBUF - 1)], *const Kl = Kh + ATT_CH * KS, *const VTh = Vimg[c & (NBUF - 1)], *const VTl = VTh + HD * VS;
        if constexpr (SHADOW) {
            dma_landed();                              // this wavefront's pieces of the chunk (issued a chunk ago)
            ATT_STAMP(10 * c + 1);
        } else {
            const float4 *K4 = reinterpret_cast<const float4 *>(P.kcache + soff + ((long long)kvh * P.seq_len + t0) * HD);
            const float4 *V4 = reinterpret_cast<const float4 *>(P.vcache + soff + ((long long)kvh * P.seq_len + t0) * HD);
            kv16_stage<HD, NTH>(K4, V4, nrows, tid, Kh, Kl, VTh, VTl, c > 0);
        }
        ATT_STAMP(10 * c + 3);
        __syncthreads();                               // the chunk is in LDS; every wavefront is past the previous chunk
        if constexpr (SHADOW) {
            if (c + 1 < c_count) dma_chunk(c_first + c + 1, (c + 1) & 1);
        }
        ATT_STAMP(10 * c + 4);

        // ---- S^T = K q^T: 8 key tiles x 16 rows; lane (j, kq) ends up with row j's scores for keys 16*mt + 4*kq + r
        v4f acc[8];
#pragma unroll
        for (int mt = 0; mt < 8; mt++) acc[mt] = (v4f){0.f, 0.f, 0.f, 0.f};
        // (two instruction streams, chosen by a uniform test: three products per term, or the fp16x1 precision mode's one)
        auto scores = [&](auto x1_tag) {
            constexpr bool X1 = decltype(x1_tag)::value;
#pragma unroll
            for (int ks = 0; ks < NKS; ks++) {
#pragma unroll
                for (int mh = 0; mh < 8; mh += 4) {        // four key tiles at a time: 32 operand registers live, not 64
                    half8_t kh[4], kl[4];
#pragma unroll
                    for (int mt = 0; mt < 4; mt++) {
                        const int ko = Img::k_off((mh + mt) * 16 + j, 4 * ks + kq);
                        kh[mt] = *reinterpret_cast<const half8_t *>(Kh + ko);
                        if (!X1) kl[mt] = *reinterpret_cast<const half8_t *>(Kl + ko);
                    }
                    if (!X1) {
#pragma unroll
                        for (int mt = 0; mt < 4; mt++) acc[mh + mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl[mt], qh[ks], acc[mh + mt], 0, 0, 0);
#pragma unroll
                        for (int mt = 0; mt < 4; mt++) acc[mh + mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[mt], ql[ks], acc[mh + mt], 0, 0, 0);
                    }
#pragma unroll
                    for (int mt = 0; mt < 4; mt++) acc[mh + mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[mt], qh[ks], acc[mh + mt], 0, 0, 0);
                }
            }
        };
        if (P.x1) scores(std::true_type{});
        else scores(std::false_type{});

        ATT_STAMP(10 * c + 5);
        // ---- online softmax of row j: 32 keys of this chunk in this lane, the rest in lanes j+16, j+32, j+48.  Chunks
        //      below the tile's diagonal see all 128 keys in every row (uniform test): no per-element masks there.
        //      The vector pipe is this kernel's busiest (SQ_INSTS_VALU x 4 cycles ~ 40 % of the launch, all of it on the
        //      critical path of the longest runs), so everything is kept in the log2 domain and two elements per
        //      instruction (v_pk_mul_f32 / v_pk_add_f32): t = s * (scale * log2 e); p * 2^10 = exp2(t + (10 - max)) on
        //      v_exp_f32 (~1 ulp; the two roundings of the argument add < 1.5e-6 relative for |t| < 32); max and sum
        //      leave in natural units (max * ln 2, sum * 2^-10) ----
        //      (two separate instruction streams: with the masked variant merely branched around, hipcc hoists its 32
        //      compares into the common path)
        float alpha;
        auto softmax = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            [[maybe_unused]] const int nvj = FULL ? ATT_CH : min(ATT_CH, max(0, npos[w * 16 + j] - t0));
            float mc = -INFINITY;
#pragma unroll
            for (int mt = 0; mt < 8; mt++) {
                if constexpr (FULL) {
                    acc[mt] = acc[mt] * c2;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; r++) acc[mt][r] = mt * 16 + 4 * kq + r < nvj ? acc[mt][r] * c2 : -INFINITY;
                }
                mc = fmaxf(fmaxf(mc, fmaxf(acc[mt][0], acc[mt][1])), fmaxf(acc[mt][2], acc[mt][3]));
            }
            mc = rows4_max(mc);
            const float mn = fmaxf(m, mc);
            // weight of what has been accumulated so far (nothing yet, or a row with no visible key so far: 0)
            alpha = m == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m - mn);
            const float nb = 10.0f - mn;
            v2f ls = {0.f, 0.f};
#pragma unroll
            for (int mt = 0; mt < 8; mt++) {
                if constexpr (FULL) {
                    const v4f t = acc[mt] + nb;
                    acc[mt] = (v4f){__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1]), __builtin_amdgcn_exp2f(t[2]), __builtin_amdgcn_exp2f(t[3])};
                } else {
#pragma unroll
                    for (int r = 0; r < 4; r++) acc[mt][r] = mt * 16 + 4 * kq + r < nvj ? __builtin_amdgcn_exp2f(acc[mt][r] + nb) : 0.f;
                }
                ls += acc[mt].lo;
                ls += acc[mt].hi;
            }
            const float lc = rows4_sum(ls[0] + ls[1]);
            l = l * alpha + lc;
            m = mn;
        };
        if (full) softmax(std::true_type{});
        else softmax(std::false_type{});
        if (c > 0) {   // (uniform)
#pragma unroll
            for (int nt = 0; nt < NTO; nt++) o[nt] = o[nt] * alpha;
        }

        ATT_STAMP(10 * c + 6);
        // ---- O^T += V^T P^T: A = V^T rows from LDS, B = P^T (this lane's own accumulators, key tiles 2m and 2m+1).  The
        //      transposed product keeps query row j in lane column j, where S^T left its max and sum: the running output is
        //      rescaled in place, and a lane ends up with four consecutive head_dim elements of its row ----
        auto pv = [&](auto x1_tag) {
            constexpr bool X1 = decltype(x1_tag)::value;
#pragma unroll
            for (int mm = 0; mm < 4; mm++) {
                // p * 2^10 = hi + lo: hi = the leading 11 bits (a mask, exact in fp16 down to its subnormals), lo = the rest
                // (fp16x1: p rounded to fp16, no lo)
                half8_t ph, pl;
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const v4f &a4 = acc[2 * mm + (e >> 2)];
                    const float p0 = a4[e & 2], p1 = a4[(e & 2) + 1];
                    if (X1) { ph[e] = (_Float16)p0; ph[e + 1] = (_Float16)p1; }
                    else {
                        const float h0 = __uint_as_float(__float_as_uint(p0) & 0xFFFFE000u), h1 = __uint_as_float(__float_as_uint(p1) & 0xFFFFE000u);
                        const v2f lf = (v2f){p0, p1} - (v2f){h0, h1};
                        ph[e] = (_Float16)h0; ph[e + 1] = (_Float16)h1;
                        pl[e] = (_Float16)lf[0]; pl[e + 1] = (_Float16)lf[1];
                    }
                }
#pragma unroll
                for (int nt = 0; nt < NTO; nt++) {
                    // B rows nt*16 + j, key segment 4*mm + kq (Kv16Image: swizzled, conflict-free)
                    const int vo = Img::v_off(nt * 16 + j, 4 * mm + kq);
                    const half8_t vh = *reinterpret_cast<const half8_t *>(VTh + vo);
                    if (!X1) {
                        const half8_t vl = *reinterpret_cast<const half8_t *>(VTl + vo);
                        o[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, pl, o[nt], 0, 0, 0);
                        o[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl, ph, o[nt], 0, 0, 0);
                    }
                    o[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, ph, o[nt], 0, 0, 0);
                }
            }
        };
        if (P.x1) pv(std::true_type{});
        else pv(std::false_type{});
        ATT_STAMP(10 * c + 7);
    }

    // O^T tile: this lane holds row j's head_dim elements nt*16 + 4*kq .. +3 (the four lanes of a row write 64 contiguous
    // bytes per nt); rows that see no key of this run leave no partial (the merge does not count this slot for them)
    constexpr float unscale = 1.0f / 1024.0f;
    const int t_first = c_first * ATT_CH;
    if (npos[w * 16 + j] > t_first) {
        const int vh = w * 16 + j, item = i0 + vh / G, h = kvh * G + vh % G;
        const long long slot = (long long)item * P.part_item_stride + (long long)h * P.nsplit_max + split;
        float *po = P.part_o + slot * HD + 4 * kq;
#pragma unroll
        for (int nt = 0; nt < NTO; nt++) {
            const v4f ov = o[nt] * unscale;
            *reinterpret_cast<float4 *>(po + nt * 16) = make_float4(ov[0], ov[1], ov[2], ov[3]);
        }
    }
    if (kq == 0 && npos[w * 16 + j] > t_first) {
        const int vh = w * 16 + j, item = i0 + vh / G, h = kvh * G + vh % G;
        const long long slot = (long long)item * P.part_item_stride + (long long)h * P.nsplit_max + split;
        P.part_ml[slot * 2] = m * LN2;
        P.part_ml[slot * 2 + 1] = l * unscale;
    }
    ATT_STAMP(8);
}

// query tokens per workgroup of the fp16 tile kernel: 128 (token, head) rows = 8 wavefronts share one staged K/V
// split (measured at 2047 tokens, mini: 64 rows 67 us, 128 rows 52 us, 256 rows 67 us per layer)
#ifndef NL_ATT16_MUL
#define NL_ATT16_MUL 2
#endif
template <int G> struct AttnTile16QT { static constexpr int value = NL_ATT16_MUL * (G == 1 ? 64 : G == 2 ? 32 : G == 8 ? 8 : 16); };
// ... and 256 rows = 16 wavefronts, one workgroup per CU, when the chunks come as prebuilt images (SHADOW, see the kernel)
template <int G> struct AttnTile16ShadowQT { static constexpr int value = 2 * AttnTile16QT<G>::value; };
template <int G> struct AttnTileQT { static constexpr int value = G == 1 ? 64 : G == 2 ? 32 : G == 8 ? 8 : 16; };

}  // namespace nl
