// nl_group.h -- larger models (GQA, any width): Q/K/V projection + RoPE + KV store + attention in ONE launch.
//
// go/model.go:517-587.  A head's weights are far too many bytes for a few compute units here (7.9B tier: 885 KB per
// kv group), so the projection keeps its chip-wide spread -- one or two 16-row tiles per workgroup, 16 wavefronts
// splitting the columns -- and only the tiny attention step is pulled in: the workgroups that hold the tiles of one kv
// group form a cluster (blocks with equal index mod 8: one XCD as observed -- speed only; a poll that gives up makes the host redo the step on the general plan), publish their rows as 8-byte {tag, value} granules
// (cdna_hip_programming.md G16 form R2), and G of them -- one per query head of the group -- gather q, k, v and run the
// softmax attention, writing the same (max, sum, sum p*v) partials the stand-alone attention launch writes; the WO
// GEMV's prologue consumes them unchanged.  This removes one of the five dependent launches per layer (~6.5 us of pure
// latency at short contexts for the 7.9B tier) for the price of the exchange + attention inside the projection launch.
// The small tiers use the deeper fusion of nl_block.h instead.  Short contexts only (the engine switches plans).
#pragma once
#include "nl_kernels.h"

namespace nl {

constexpr int GRP_THREADS = 1024;
constexpr int GRP_KV_THREADS = 512;

struct GroupParams {
    const uint8_t *qkv_q;
    const uint32_t *qkv_s;
    int D, npairs, n_q_heads, n_kv_heads, seq_len, rope_conj, qk_norm, single_stream;
    int tpm, members;            // 16-row tiles per workgroup; workgroups per kv group = (G + 2) * 4 / tpm
    unsigned gqa, wpt, wpt_inv, m8_inv, m_inv;   // G, wavefronts per tile, udiv_inv of wpt / 8 * members / members (host: no divisions at entry)
    const float *x, *normw;
    float eps, scale;
    const float *rope_cos, *rope_sin;
    float *kcache, *vcache;
    long long kv_stream_stride;
    const int *ctl;
    const float *bias_q, *bias_k, *bias_v;
    float *part_o, *part_ml;     // [heads][nsplit_max][64] / [heads][nsplit_max][2], as attn_kernel writes them
    int nsplit_max;
    unsigned long long *xchg;    // [kv groups][(G + 2) * 64] granules
    const unsigned *tick;
    unsigned layer_tag;
    unsigned *status, *host_status;
    int spin_limit;              // polls before a wavefront gives up (see BlockParams, nl_block.h)
};

__host__ __device__ constexpr size_t grp_lds_bytes() {
    return sizeof(float) * (size_t)(16 * XS_WAVE + 16 * TR + 3 * 64 + 8 + ATT_CH + 32 * 64) + 16 * sizeof(double);
}
// block b: cluster (b / (8 * members)) * 8 + b % 8, member (b / 8) % members  (cluster members share b % 8: an XCD as observed, not a contract)
inline int grp_grid(int clusters, int members) { return ((clusters + 7) / 8) * members * 8; }

template <int WT, int NF>
__global__ void __launch_bounds__(GRP_THREADS) qkv_attn_kernel(GroupParams P) {
    constexpr int HD = 64, CPP = WTraits<WT>::CPP, R4 = HD / 4, NGR = GRP_KV_THREADS / R4, NV = ATT_CH / NGR, NW = GRP_THREADS / 64;
    NL_KARGS8(P.qkv_q, P.qkv_s, P.x, P.normw, P.rope_cos, P.rope_sin, P.kcache, P.vcache);   // one batch of s_load (nl_kernels.h)
    NL_KARGS8(P.ctl, P.bias_q, P.part_o, P.part_ml, P.xchg, P.tick, P.status, P.host_status);
    NL_KARGS8(P.D, P.npairs, P.n_q_heads, P.n_kv_heads, P.seq_len, P.single_stream, P.tpm, P.members);
    NL_KARGS8(P.eps, P.scale, P.kv_stream_stride, P.nsplit_max, P.layer_tag, P.rope_conj, P.qk_norm, P.bias_k);
    NL_KARGS2(P.spin_limit, P.gqa);
    const int M = P.members;
    const int b8 = (int)(blockIdx.x >> 3);
    const int cl = (int)udiv_by(blockIdx.x, 8u * (unsigned)M, P.m8_inv) * 8 + (blockIdx.x & 7), mem = b8 - (int)udiv_by((unsigned)b8, (unsigned)M, P.m_inv) * M;
    if (cl >= P.n_kv_heads) return;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *dred = reinterpret_cast<double *>(smem);                 // [16]
    float *xs = reinterpret_cast<float *>(dred + 16);                // [16][XS_WAVE]
    float *red = xs + NW * XS_WAVE;                                  // [16][16]
    float *qs = red + NW * TR;                                       // [64]
    float *kcur = qs + 64, *vcur = kcur + 64;
    float *ml = vcur + 64;                                           // [8]
    float *sc = ml + 8;                                              // [128]
    float *ored = sc + ATT_CH;                                       // [32][64]

    const int G = (int)P.gqa, D = P.D;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int r = lane >> 2, k = lane & 3;
    const int wpt = (int)P.wpt;                    // wavefronts per tile (NW / tpm)
    const int slot = (int)udiv_by((unsigned)wave, P.wpt, P.wpt_inv), cs = wave - slot * wpt;  // tile of this workgroup, column share
    // tile u of the group: q heads first (4 tiles each), then k, then v
    auto tile_of = [&](int u, int &sect, int &hq, int &j) {
        j = u & 3;
        hq = u >> 2;
        sect = hq < G ? 0 : hq == G ? 1 : 2;
        return (sect == 0 ? cl * G + hq : sect == 1 ? P.n_q_heads + cl : P.n_q_heads + P.n_kv_heads + cl) * 4 + j;
    };
    int w_sect, w_hq, w_j;
    const int tile = tile_of(mem * P.tpm + slot, w_sect, w_hq, w_j);

    // ---- loads that depend on nothing ----
    const int pos = sload_i32(P.ctl + CTL_POS);       // (scalar cache: no vector wait in front of the weight requests)
    const long long soff = P.single_stream ? 0 : (long long)sload_i32(P.ctl + CTL_STREAM) * P.kv_stream_stride;
    const unsigned tag = (__hip_atomic_load(P.tick, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << 8) | P.layer_tag;
    const long long tp0 = (long long)tile * P.npairs;
    const int ngroups = (P.npairs + KL - 1) / KL;
    float4 xv[NF], gv[NF];
    uint4 cw[NF][CPP];
    uint2 sw[NF];
    bool lv[NF], xin[NF];
#pragma unroll
    for (int f = 0; f < NF; f++) {
        const int g = cs + f * wpt, gg = min(g, ngroups - 1);
        const int gs = min(KL, P.npairs - gg * KL);
        lv[f] = g < ngroups && k < gs;
        const int xcol = gg * (KL * PAIR) + lane * 4;
        xin[f] = xcol < D;
        xv[f] = ld_off<float4>(P.x, (unsigned)(xin[f] ? xcol : 0) * 4u);
        gv[f] = ld_off<float4>(P.normw, (unsigned)(xin[f] ? xcol : 0) * 4u);
        load_pair<WT>(P.qkv_q, P.qkv_s, tp0, gg, gs, r, min(k, gs - 1), cw[f], sw[f]);
    }
    const bool attn_member = mem < G;    // this workgroup runs the attention of query head cl * G + mem
    const int c4 = tid % R4, tg = tid / R4;
    float4 kreg[NV], vreg[NV], kregn[NV];
    const float4 *K4 = reinterpret_cast<const float4 *>(P.kcache + soff + (long long)cl * P.seq_len * HD);
    const float4 *V4 = reinterpret_cast<const float4 *>(P.vcache + soff + (long long)cl * P.seq_len * HD);
    // epilogue inputs (threads 0 .. tpm * 16 - 1: one projection row each)
    const int e_slot = tid >> 4, e_rr = tid & 15;
    int e_sect = 0, e_hq = 0, e_j = 0;
    float e_cos = 0.f, e_sin = 0.f, e_b = 0.f, e_bp = 0.f;
    const bool e_act = tid < P.tpm * TR;
    if (e_act) tile_of(mem * P.tpm + e_slot, e_sect, e_hq, e_j);
    const int e_i = e_j * 8 + (e_rr & 7), e_e = e_i + (e_rr >> 3) * (HD / 2);
    if (e_act) {
        e_cos = P.rope_cos[pos * (HD / 2) + e_i];
        e_sin = P.rope_sin[pos * (HD / 2) + e_i];
        if (P.bias_q) {   // addBias before RoPE, go/model.go:525-527
            const float *b = e_sect == 0 ? P.bias_q + (cl * G + e_hq) * HD : e_sect == 1 ? P.bias_k + cl * HD : P.bias_v + cl * HD;
            e_b = b[e_e];
            e_bp = b[e_e ^ (HD / 2)];
        }
    }

    // ---- RMSNorm scaling (go/quant.go:597-607) into wave-private LDS, dot products of this wavefront's column groups ----
    float *xw = xs + wave * XS_WAVE;
    double ss = 0.0;
    float acc = 0.f;
#pragma unroll
    for (int f = 0; f < NF; f++) {
        float4 xa = xin[f] ? xv[f] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (slot == 0 && cs + f * wpt < ngroups) {   // the wavefronts of tile 0 see every column exactly once
            ss = fma((double)xa.x, (double)xa.x, ss); ss = fma((double)xa.y, (double)xa.y, ss);
            ss = fma((double)xa.z, (double)xa.z, ss); ss = fma((double)xa.w, (double)xa.w, ss);
        }
        xa.x *= gv[f].x; xa.y *= gv[f].y; xa.z *= gv[f].z; xa.w *= gv[f].w;
        *reinterpret_cast<float4 *>(xw + (lane >> 4) * XS_PAIR + (lane & 15) * 4) = xa;
        __builtin_amdgcn_wave_barrier();
        const float a1 = PairDot<WT>::run(cw[f], sw[f], xw + k * XS_PAIR, acc);
        acc = lv[f] ? a1 : acc;
        __builtin_amdgcn_wave_barrier();
    }
    // The head's cache rows are requested only now, after the projection dots (round 4): a wavefront issues in order and a
    // compute unit ingests ~11 B/clk, so requested at entry the 64 KB of rows delayed the attention blocks' projection tiles
    // -- and with them the exchange every block of the cluster waits for.  Rows beyond pos repeat row pos (the same cache
    // lines; they are masked below), the row AT pos comes from LDS.
    if (attn_member && tid < GRP_KV_THREADS) {
        const int lim = min(min(ATT_CH, P.seq_len), pos + 1);
#pragma unroll
        for (int kk = 0; kk < NV; kk++) {
            const unsigned ro = (unsigned)(min(tg + kk * NGR, lim - 1) * R4 + c4) * 16u;
            kreg[kk] = ld_off<float4>(K4, ro);
            vreg[kk] = ld_off<float4>(V4, ro);
        }
    }
    acc = quad_sum(acc);
    if (k == 0) red[wave * TR + r] = acc;
    ss = wave_sum_f64(ss);
    if (slot == 0 && lane == 0) dred[cs] = ss;
    __syncthreads();

    // ---- scale, bias, RoPE (go/model.go:449-477); publish this workgroup's rows to the cluster ----
    const int gvec = (G + 2) * HD;   // q heads | k | v of the group
    if (e_act) {
        double tot = 0.0;
        for (int w = 0; w < wpt; w++) tot += dred[w];
        const float inv = (float)(1.0 / sqrt(tot / (double)D + (double)P.eps));
        const float *rt = red + e_slot * wpt * TR;
        float dotv = 0.f, dotp = 0.f;
        for (int w = 0; w < wpt; w++) { dotv += rt[w * TR + e_rr]; dotp += rt[w * TR + (e_rr ^ 8)]; }   // fixed order
        const float v = dotv * inv + e_b, partner = dotp * inv + e_bp;
        float outv = v;
        if (e_sect < 2) {
            const float x0 = (e_rr < 8) ? v : partner, x1 = (e_rr < 8) ? partner : v;
            if (!P.rope_conj) outv = (e_rr < 8) ? (x0 * e_cos - x1 * e_sin) : (x0 * e_sin + x1 * e_cos);
            else outv = (e_rr < 8) ? (x0 * e_cos + x1 * e_sin) : (-x0 * e_sin + x1 * e_cos);
        }
        __hip_atomic_store(P.xchg + (size_t)cl * gvec + e_hq * HD + e_e, ((unsigned long long)tag << 32) | __float_as_uint(outv),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!attn_member) return;

    // ---- the attention of query head cl * G + mem (go/model.go:557-587): gather q | k | v ----
    const int h = cl * G + mem;
    if (tid < 192) {
        const bool dead = __hip_atomic_load(P.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        const int src = tid < 64 ? mem * HD + tid : G * HD + (tid - 64);
        unsigned long long gq;
        for (int spins = 0;; spins++) {
            gq = __hip_atomic_load(P.xchg + (size_t)cl * gvec + src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all((unsigned)(gq >> 32) == tag)) break;
            if (dead || spins >= P.spin_limit) { if (lane == 0) { atomicOr(P.status, 8u); *P.host_status = 8u; } break; }
            __builtin_amdgcn_s_sleep(1);
        }
        qs[tid] = __uint_as_float((unsigned)gq);   // qs | kcur | vcur are contiguous
    }
    __syncthreads();
    if (P.qk_norm) {   // RMSNormBare per head on q and k after RoPE, go/model.go:542-549
        if (wave < 2) {
            float *vec = wave == 0 ? qs : kcur;
            const float val = vec[lane];
            const double s2 = wave_sum_f64((double)val * (double)val);
            const float inv = (float)(1.0 / sqrt(s2 / (double)HD + (double)P.eps));
            vec[lane] = val * inv;
        }
        __syncthreads();
    }
    if (mem == 0 && tid < 128)   // KV store go/model.go:552-554, once per kv head
        (tid < 64 ? P.kcache : P.vcache)[soff + ((long long)cl * P.seq_len + pos) * HD + (tid & 63)] = kcur[tid];

    const int nch = pos / ATT_CH + 1;
    for (int ch = 0; ch < nch; ch++) {
        const int t0 = ch * ATT_CH, n = min(ATT_CH, pos + 1 - t0);
        if (tid < GRP_KV_THREADS) {
            // K of the next pass and V of this pass are requested now: V is not needed before the P*V phase (two barriers
            // away) and the next K rows arrive during this pass, so a later pass costs its arithmetic, not a memory round trip
            if (ch > 0) {
#pragma unroll
                for (int kk = 0; kk < NV; kk++) {
                    kreg[kk] = kregn[kk];
                    vreg[kk] = V4[(long long)(t0 + min(tg + kk * NGR, n - 1)) * R4 + c4];
                }
            }
            if (ch + 1 < nch) {
                const int n1 = min(ATT_CH, pos + 1 - t0 - ATT_CH);
#pragma unroll
                for (int kk = 0; kk < NV; kk++) kregn[kk] = K4[(long long)(t0 + ATT_CH + min(tg + kk * NGR, n1 - 1)) * R4 + c4];
            }
            // scores: this thread holds 4 of the 64 dims of 4 cache rows; the 16 lanes of a row sum on DPP
            const float4 q4 = *reinterpret_cast<const float4 *>(qs + c4 * 4);
            const float4 kc4 = *reinterpret_cast<const float4 *>(kcur + c4 * 4), vc4 = *reinterpret_cast<const float4 *>(vcur + c4 * 4);
#pragma unroll
            for (int kk = 0; kk < NV; kk++) {
                const int row = tg + kk * NGR;
                const bool cur = t0 + row == pos;   // the row this launch produced: taken from LDS, not from memory
                kreg[kk].x = cur ? kc4.x : kreg[kk].x; kreg[kk].y = cur ? kc4.y : kreg[kk].y;
                kreg[kk].z = cur ? kc4.z : kreg[kk].z; kreg[kk].w = cur ? kc4.w : kreg[kk].w;
                vreg[kk].x = cur ? vc4.x : vreg[kk].x; vreg[kk].y = cur ? vc4.y : vreg[kk].y;
                vreg[kk].z = cur ? vc4.z : vreg[kk].z; vreg[kk].w = cur ? vc4.w : vreg[kk].w;
                float d = fmaf(q4.w, kreg[kk].w, fmaf(q4.z, kreg[kk].z, fmaf(q4.y, kreg[kk].y, q4.x * kreg[kk].x)));
                d += dpp_f32<DPP_QUAD_XOR1>(d);
                d += dpp_f32<DPP_QUAD_XOR2>(d);
                d += dpp_f32<DPP_HALF_MIRROR>(d);
                d += dpp_f32<DPP_ROW_MIRROR>(d);
                if (c4 == 0 && row < n) sc[row] = d * P.scale;
            }
        }
        __syncthreads();
        if (wave == 0) {   // Softmax go/quant.go:610-626 over the pass, as attn_kernel does
            const float s0 = lane < n ? sc[lane] : -INFINITY;
            const float s1 = lane + 64 < n ? sc[lane + 64] : -INFINITY;
            const float m = wave_max_f32(fmaxf(s0, s1));
            const float p0 = lane < n ? exp_f64_as_f32(s0 - m) : 0.f;
            const float p1 = lane + 64 < n ? exp_f64_as_f32(s1 - m) : 0.f;
            if (lane < n) sc[lane] = p0;
            if (lane + 64 < n) sc[lane + 64] = p1;
            const float l = wave_sum_f32(p0 + p1);
            if (lane == 0) { ml[0] = m; ml[1] = l; }
        }
        __syncthreads();
        if (tid < GRP_KV_THREADS) {
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int kk = 0; kk < NV; kk++) {
                const int row = tg + kk * NGR;
                const float pw = row < n ? sc[row] : 0.f;   // (a masked row's V may be stale but is finite)
                o.x = fmaf(pw, vreg[kk].x, o.x); o.y = fmaf(pw, vreg[kk].y, o.y);
                o.z = fmaf(pw, vreg[kk].z, o.z); o.w = fmaf(pw, vreg[kk].w, o.w);
            }
            *reinterpret_cast<float4 *>(ored + tg * HD + c4 * 4) = o;
        }
        __syncthreads();
        if (tid < HD) {
            float s = 0.f;
#pragma unroll 8
            for (int kk = 0; kk < NGR; kk++) s += ored[kk * HD + tid];
            P.part_o[((long long)h * P.nsplit_max + ch) * HD + tid] = s;
            if (tid == 0) {
                P.part_ml[((long long)h * P.nsplit_max + ch) * 2] = ml[0];
                P.part_ml[((long long)h * P.nsplit_max + ch) * 2 + 1] = ml[1];
            }
        }
        __syncthreads();
    }
}

}  // namespace nl
