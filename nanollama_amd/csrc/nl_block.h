// nl_block.h -- the attention half of a decoder layer as ONE launch, for small models.
//
// go/model.go:517-594 is five dependent steps per token and layer: RMSNorm, Q/K/V projections, RoPE + KV store,
// GQA attention over the cache, WO + residual.  Only the first and the last are all-to-all over the model width;
// everything between is local to a head.  For the small tiers a decode launch costs ~3.8 us whatever it moves
// (1.5 us boundary + one cold memory round trip + reductions), so three launches for this half is ~12 us per layer.
//
// Here a CLUSTER of four workgroups per query head does the whole chain in one launch:
//     RMSNorm(x) -> each member projects one 16-row tile of q, of k and of v (the k / v rows of a GQA group are
//     recomputed by every head of the group: the work is latency, not bytes) -> bias, RoPE -> the members exchange
//     their 3 x 16 values through 8-byte {tag, value} granules (the only cross-workgroup step: 192 values per head,
//     cdna_hip_programming.md G16 form R2; cluster members are the blocks b, b+8, b+16, b+24, which the dispatcher
//     happens to place on one XCD -- observed, used for speed only, never relied on) -> QK-norm -> KV store (first
//     head of the group) -> every member runs the head's softmax
//     attention over the cache (the cache rows come from L2) -> each member multiplies ITS quarter of the head's
//     64-column WO slice, a partial [D/4] vector.
// The H partial vectors are added to the residual stream by the consuming gate/up GEMV (PRO_NORM_PARTS, fixed head
// order).  All weight, x and cache loads are issued at entry, so the launch pays one memory latency; the WO rows
// stream in while the attention runs.  A first version with ONE workgroup per head measured 10.6 us per launch on
// nano -- a head's 156 KB of weights through a single compute unit -- against 12.1 us for the three launches it
// replaces; spreading the head over four compute units is what makes the fusion pay.
// The engine selects this path only where a head's weights are small (nano, mini) and only at short contexts
// (every cache position of the head passes through each member).
#pragma once
#include "nl_kernels.h"

namespace nl {

constexpr int BLK_THREADS = 768;     // 12 wavefronts: {q, k, v tile of this member} x 4 groups of 256 columns
constexpr int BLK_KV_THREADS = 512;  // threads that hold cache rows (32 row groups x 16 float4)
constexpr int BLK_MEMBERS = 4;       // workgroups per head = 16-row tiles per 64-element head
constexpr int BLK_MAXG = 4;          // 256-column groups per row: D <= 1024
constexpr int BLK_MAX_PARTS = 12;    // heads whose partials the consumer adds (PRO_NORM_PARTS)

struct BlockParams {
    const uint8_t *qkv_q;        // packed Q|K|V of the layer (ROWMAP_HEADPERM tiles)
    const uint32_t *qkv_s;
    const uint8_t *wo_q;         // per-head WO slices: [H][D/16 tiles][1 pair]
    const uint32_t *wo_s;
    int D, npairs, n_q_heads, n_kv_heads, seq_len, rope_conj, qk_norm, single_stream;
    unsigned gqa, gqa_inv;       // query heads per kv head and udiv_inv of it (head -> kv head without a division)
    const float *x, *normw;
    float eps, scale;
    const float *rope_cos, *rope_sin;
    float *kcache, *vcache;      // this layer, stream 0: [kv][seq][64]
    long long kv_stream_stride;
    const int *ctl;
    const float *bias_q, *bias_k, *bias_v, *bias_out;
    float *parts;                // [H][D]: head h's share of WO * attention output (+ bias_out in head 0)
    const float *parts_in;       // optional [nparts_in][D]: the previous layer's feed-forward partials, added to x first
    int nparts_in;
    float *x_out;                // with parts_in: x + sum parts_in, stored by head 0 / member 0
    unsigned long long *xchg;    // [H][192] granules: the head's q | k | v of this position
    const unsigned *tick;        // forward counter (advanced by the embedding launch): tag = tick << 8 | layer + 1
    unsigned layer_tag;
    unsigned *status;            // set non-zero if an exchange poll gave up: HIP promises nothing about co-residence, so the
                                 // host then redoes the step on the general plan and retires the fused one (nl_engine.hip)
    unsigned *host_status;       // the same flag in host-visible memory (read by the host without a copy)
    int spin_limit;              // polls before a wavefront gives up (NL_FUSED_SPIN_LIMIT; a test forces 0)
    long long *dbg;              // optional phase stamps (wall_clock64, 100 MHz... s_memtime shader clock) of block 0
};

__host__ __device__ constexpr int blk_xs_floats(int D) { return (D / PAIR) * XS_PAIR; }
__host__ __device__ constexpr size_t blk_lds_bytes(int D) {
    return sizeof(float) * (size_t)(4 * XS_WAVE + 12 * TR + 4 * 64 + 8 + ATT_CH + 32 * 66 + 2 * 32 * 64) + 16 * sizeof(double);
}
inline int blk_grid(int heads) { return ((heads + 7) / 8) * 32; }   // block b: head (b/32)*8 + b%8, member (b/8)%4

template <int WT, int NPIN>
__global__ void __launch_bounds__(BLK_THREADS) attn_block_kernel(BlockParams P) {
    constexpr int HD = 64, CPP = WTraits<WT>::CPP, R4 = HD / 4, NGR = BLK_KV_THREADS / R4, NV = ATT_CH / NGR, NW = BLK_THREADS / 64;
    NL_KARGS8(P.qkv_q, P.qkv_s, P.wo_q, P.wo_s, P.x, P.normw, P.kcache, P.vcache);
    NL_KARGS8(P.ctl, P.parts, P.parts_in, P.x_out, P.xchg, P.tick, P.rope_cos, P.rope_sin);
    NL_KARGS8(P.D, P.npairs, P.n_q_heads, P.n_kv_heads, P.seq_len, P.nparts_in, P.layer_tag, P.single_stream);
    NL_KARGS8(P.dbg, P.host_status, P.status, P.bias_q, P.bias_out, P.eps, P.scale, P.kv_stream_stride);
    NL_KARGS2(P.spin_limit, P.gqa);
    const int h = (blockIdx.x >> 5) * 8 + (blockIdx.x & 7), jm = (blockIdx.x >> 3) & 3;
    if (h >= P.n_q_heads) return;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *dred = reinterpret_cast<double *>(smem);                 // [16]
    float *xs = reinterpret_cast<float *>(dred + 16);                // [4][XS_WAVE]: x * g per 256-column group
    float *red = xs + 4 * XS_WAVE;                                   // [12][16]
    float *qs = red + NW * TR;                                        // [64]
    float *kcur = qs + 64, *vcur = kcur + 64;                        // this position's K / V row
    float *on = vcur + 64;                                           // [64] normalised attention output
    float *ml = on + 64;                                             // [8]
    float *sc = ml + 8;                                              // [128]
    float *chunk = sc + ATT_CH;                                      // [32][66]: (m, l, o[64]) per 64-position half pass
    float *ored = chunk + 32 * 66;                                   // [2][32][64]

    const int G = (int)P.gqa, kvh = (int)udiv_by((unsigned)h, P.gqa, P.gqa_inv);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int r = lane >> 2, k = lane & 3;
    const int D = P.D;
#define BLK_STAMP(i) do { if (P.dbg && blockIdx.x == 0 && tid == 0) P.dbg[i] = clock64(); } while (0)
    BLK_STAMP(0);

    // ---- every load of the launch that does not depend on a result is issued here ----
    const int pos = sload_i32(P.ctl + CTL_POS);
    const long long soff = P.single_stream ? 0 : (long long)sload_i32(P.ctl + CTL_STREAM) * P.kv_stream_stride;
    const unsigned tag = (__hip_atomic_load(P.tick, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << 8) | P.layer_tag;
    const int sect = wave >> 2, grp = wave & 3;   // 0 q, 1 k, 2 v; this wavefront's 256-column group
    const int tile = (sect == 0 ? h : sect == 1 ? P.n_q_heads + kvh : P.n_q_heads + P.n_kv_heads + kvh) * 4 + jm;
    const long long tp0 = (long long)tile * P.npairs;
    const int ngroups = (P.npairs + KL - 1) / KL;
    uint4 cw[CPP];
    uint2 sw;
    const int gg = min(grp, ngroups - 1), gs = min(KL, P.npairs - gg * KL);
    const bool lv = grp < ngroups && k < gs;
    // x and the norm weights of this wavefront's own 256 columns (one float4 per lane), then its weights.  A wavefront
    // whose column group does not exist (D = 576: three groups, four wavefronts per tile) skips the loads as a whole
    // (wave-uniform branch: no lane waits on a partial load) and contributes zeros.
    const int xcol = gg * (KL * PAIR) + lane * 4;
    const bool xin = xcol < D;
    // (values of a wavefront that skips the loads below are never consumed unmasked: x / partials only by the staging
    //  wavefronts that load them, a missing column group's dot product is replaced by 0)
    float4 xv, gv;
    undef_regs(xv); undef_regs(gv);
#pragma unroll
    for (int j = 0; j < CPP; j++) undef_regs(cw[j]);
    undef_regs(sw);
    float4 pv[NPIN > 0 ? NPIN : 1];
#pragma unroll
    for (int p = 0; p < (NPIN > 0 ? NPIN : 1); p++) undef_regs(pv[p]);
    if (grp < ngroups) {
        if (sect == 0) {   // the q wavefronts stage x for the whole workgroup: one read of x (+ partials) per workgroup, not three
            const unsigned xo = (unsigned)(xin ? xcol : 0) * 4u;
            xv = ld_off<float4>(P.x, xo);
            gv = ld_off<float4>(P.normw, xo);
#pragma unroll
            for (int p = 0; p < NPIN; p++) pv[p] = ld_off<float4>(P.parts_in, (unsigned)(min(p, P.nparts_in - 1) * D) * 4u + xo);
        }
        load_pair<WT>(P.qkv_q, P.qkv_s, tp0, gg, gs, r, min(k, gs - 1), cw, sw);
    }
    // The first 128 cache rows of this kv head (rows >= pos hold stale finite data and are masked / replaced below).
    // Each of the eight cache wavefronts owns 16 consecutive positions of a pass.  K: lane (row kr = lane / 4, quarter
    // kq = lane % 4) holds 16 of the row's 64 dims -- a score is 16 in-lane FMAs and two quad adds, no LDS.  V: lane (row
    // group vg = lane / 16, float4 column vc = lane % 16) holds float4 vc of rows vg, vg + 4, vg + 8, vg + 12 -- P*V is
    // four FMAs per component and two cross-row adds.  The wavefront reduces its 16 positions to one (max, sum, sum p*v)
    // partial by itself; eight partials per pass meet in LDS behind ONE barrier (was: scores | softmax | P*V | 32-way
    // reduction, four barriers per pass).
    static_assert(NV == 4 && BLK_KV_THREADS == 512 && HD == 64, "eight cache wavefronts x 16 positions, 16 floats per lane");
    const int kr = lane >> 2, kq = lane & 3, vg = lane >> 4, vc = lane & 15;
    float4 kreg[NV], vreg[NV], kregn[NV];
    const float4 *K4 = reinterpret_cast<const float4 *>(P.kcache + soff + (long long)kvh * P.seq_len * HD);
    const float4 *V4 = reinterpret_cast<const float4 *>(P.vcache + soff + (long long)kvh * P.seq_len * HD);
    if (tid < BLK_KV_THREADS) {
        // (rows beyond pos repeat row pos -- the same cache lines: at short contexts a workgroup fetched 64 KB of cache rows
        //  it then masked, more than its weights)
        const int lim = min(min(ATT_CH, P.seq_len), pos + 1);
        const unsigned krow = (unsigned)min(wave * 16 + kr, lim - 1) * R4 + (unsigned)kq * 4u;
#pragma unroll
        for (int kk = 0; kk < NV; kk++) {
            kreg[kk] = ld_off<float4>(K4, (krow + kk) * 16u);
            vreg[kk] = ld_off<float4>(V4, (unsigned)(min(wave * 16 + vg + 4 * kk, lim - 1) * R4 + vc) * 16u);
        }
    }
    // epilogue inputs of the projection rows: 48 threads of the last wavefront (it holds no cache rows and stages nothing),
    // one row each
    constexpr int ET0 = BLK_THREADS - 64;
    const int et = tid - ET0;
    const bool e_thr = (unsigned)et < 48u;
    const int e_sect = (et >> 4) & 3, e_rr = et & 15;
    const int e_i = jm * 8 + (e_rr & 7), e_e = e_i + (e_rr >> 3) * (HD / 2);
    float e_cos = 0.f, e_sin = 0.f, e_b = 0.f, e_bp = 0.f;
    if (e_thr) {
        e_cos = P.rope_cos[pos * (HD / 2) + e_i];
        e_sin = P.rope_sin[pos * (HD / 2) + e_i];
        if (P.bias_q) {   // addBias before RoPE, go/model.go:525-527
            const float *b = e_sect == 0 ? P.bias_q + h * HD : e_sect == 1 ? P.bias_k + kvh * HD : P.bias_v + kvh * HD;
            e_b = b[e_e];
            e_bp = b[e_e ^ (HD / 2)];
        }
    }

    BLK_STAMP(15);
    // ---- RMSNorm (go/quant.go:597-607): x * g into this wavefront's own LDS slice (no workgroup barrier before the
    //      dot products); float64 sum of squares from the q wavefronts, which cover every column once ----
    float *xw = xs + grp * XS_WAVE;      // shared by the three wavefronts of a column group
    double ss = 0.0;
    if (sect == 0) {
        float4 xa = xv;
#pragma unroll
        for (int p = 0; p < NPIN; p++) {   // the previous layer's feed-forward partials, fixed order
            const bool on = p < P.nparts_in;
            xa.x += on ? pv[p].x : 0.f; xa.y += on ? pv[p].y : 0.f; xa.z += on ? pv[p].z : 0.f; xa.w += on ? pv[p].w : 0.f;
        }
        if (!xin || grp >= ngroups) xa = make_float4(0.f, 0.f, 0.f, 0.f);
        ss = fma((double)xa.x, (double)xa.x, ss); ss = fma((double)xa.y, (double)xa.y, ss);
        ss = fma((double)xa.z, (double)xa.z, ss); ss = fma((double)xa.w, (double)xa.w, ss);
        if (NPIN > 0 && h == 0 && jm == 0 && xin && grp < ngroups) *reinterpret_cast<float4 *>(P.x_out + xcol) = xa;   // the updated residual stream
        xa.x *= gv.x; xa.y *= gv.y; xa.z *= gv.z; xa.w *= gv.w;
        *reinterpret_cast<float4 *>(xw + (lane >> 4) * XS_PAIR + (lane & 15) * 4) = xa;
    }
    // sum of squares: 16-lane rows on DPP, four partials per q wavefront.  x arrives long before the weights do, so the
    // reduction, its barrier and the float64 sqrt / divide of inv happen inside the weight latency, not in the epilogue.
    ss += dpp_f64<DPP_QUAD_XOR1>(ss);
    ss += dpp_f64<DPP_QUAD_XOR2>(ss);
    ss += dpp_f64<DPP_HALF_MIRROR>(ss);
    ss += dpp_f64<DPP_ROW_MIRROR>(ss);
    if (sect == 0 && (lane & 15) == 0) dred[grp * 4 + (lane >> 4)] = ss;
    BLK_STAMP(1);
    __syncthreads();
    float inv = 0.f;
    if (e_thr) {
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < 16; w++) tot += dred[w];
        inv = (float)(1.0 / sqrt(tot / (double)D + (double)P.eps));
    }
    BLK_STAMP(2);

    // ---- this member's q, k and v tile: each wavefront one 256-column group of one tile ----
    float acc = PairDot<WT>::run(cw, sw, xw + k * XS_PAIR, 0.f);
    acc = lv ? acc : 0.f;
    BLK_STAMP(16);
    acc = quad_sum(acc);
    if (k == 0) red[wave * TR + r] = acc;
    BLK_STAMP(3);
    __syncthreads();
    BLK_STAMP(4);

    // this member's quarter of the head's WO slice streams in while the exchange and the attention run
    // (lane = (row, block of the head's pair): two lanes per row, each 32 of the 64 columns)
    const int wo_rows = D / BLK_MEMBERS, wo_tiles = D / TR;
    const int wrow = jm * wo_rows + min(tid >> 1, wo_rows - 1), wblk = tid & 1;
    uint4 wc[CPP / 2];
    uint32_t wd16;
    load_block<WT>(P.wo_q, P.wo_s, (long long)h * wo_tiles + (wrow >> 4), 0, 1, wrow & 15, 0, wblk, wc, wd16);

    // ---- scale, bias, RoPE (go/model.go:449-477); publish the 48 values to the other members ----
    if (e_thr) {
        const float *rt = red + e_sect * 4 * TR;   // the tile's four column-group partials, fixed order
        const float dotv = ((rt[e_rr] + rt[TR + e_rr]) + rt[2 * TR + e_rr]) + rt[3 * TR + e_rr];
        const float dotp = ((rt[e_rr ^ 8] + rt[TR + (e_rr ^ 8)]) + rt[2 * TR + (e_rr ^ 8)]) + rt[3 * TR + (e_rr ^ 8)];
        const float v = dotv * inv + e_b, partner = dotp * inv + e_bp;
        float outv = v;
        if (e_sect < 2) {
            const float x0 = (e_rr < 8) ? v : partner, x1 = (e_rr < 8) ? partner : v;
            if (!P.rope_conj) outv = (e_rr < 8) ? (x0 * e_cos - x1 * e_sin) : (x0 * e_sin + x1 * e_cos);
            else outv = (e_rr < 8) ? (x0 * e_cos + x1 * e_sin) : (-x0 * e_sin + x1 * e_cos);
        }
        __hip_atomic_store(P.xchg + (size_t)h * 192 + e_sect * 64 + e_e, ((unsigned long long)tag << 32) | __float_as_uint(outv),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    BLK_STAMP(5);
    // ---- gather the head's q | k | v (192 granules; a wavefront retries until all of its lanes see the tag) ----
    if (tid < 192) {
        const bool dead = __hip_atomic_load(P.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        unsigned long long gq;
        for (int spins = 0;; spins++) {
            gq = __hip_atomic_load(P.xchg + (size_t)h * 192 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all((unsigned)(gq >> 32) == tag)) break;
            if (dead || spins >= P.spin_limit) { if (lane == 0) { atomicOr(P.status, 4u); *P.host_status = 4u; } break; }
            __builtin_amdgcn_s_sleep(1);
        }
        qs[tid] = __uint_as_float((unsigned)gq);   // qs | kcur | vcur are contiguous
    }
    BLK_STAMP(6);
    __syncthreads();
    BLK_STAMP(7);
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
    if (h == kvh * G && jm == 0 && tid < 128)   // KV store go/model.go:552-554, once per kv head
        (tid < 64 ? P.kcache : P.vcache)[soff + ((long long)kvh * P.seq_len + pos) * HD + (tid & 63)] = kcur[tid];

    BLK_STAMP(8);
    // ---- GQA attention over positions 0..pos (go/model.go:557-587), 128 positions per pass ----
    const int nch = pos / ATT_CH + 1;
    float *wpart = ored;                     // [8][68]: a cache wavefront's (max, sum, -, -, sum p*v[64]) of the pass
    for (int ch = 0; ch < nch; ch++) {
        const int t0 = ch * ATT_CH, n = min(ATT_CH, pos + 1 - t0);
        if (tid < BLK_KV_THREADS) {
            // K of the next pass and V of this pass are requested now: the next K rows arrive during this pass, so a later
            // pass costs its arithmetic, not a memory round trip
            if (ch > 0) {
#pragma unroll
                for (int kk = 0; kk < NV; kk++) {
                    kreg[kk] = kregn[kk];
                    vreg[kk] = V4[(long long)(t0 + min(wave * 16 + vg + 4 * kk, n - 1)) * R4 + vc];
                }
            }
            if (ch + 1 < nch) {
                const int n1 = min(ATT_CH, pos + 1 - t0 - ATT_CH);
#pragma unroll
                for (int kk = 0; kk < NV; kk++) kregn[kk] = K4[(long long)(t0 + ATT_CH + min(wave * 16 + kr, n1 - 1)) * R4 + kq * 4 + kk];
            }
            // scores: 16 dims in the lane, the row's four quarters summed on DPP (valid in every lane of the quad)
            const int krow = wave * 16 + kr;
            const bool kcurrow = t0 + krow == pos;   // the row this launch produced: not in memory yet for this workgroup
            float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll
            for (int kk = 0; kk < NV; kk++) {
                const float4 q4 = *reinterpret_cast<const float4 *>(qs + kq * 16 + kk * 4);
                const float4 kc4 = *reinterpret_cast<const float4 *>(kcur + kq * 16 + kk * 4);
                const float4 k4 = kcurrow ? kc4 : kreg[kk];
                d0 = fmaf(q4.x, k4.x, d0); d1 = fmaf(q4.y, k4.y, d1); d2 = fmaf(q4.z, k4.z, d2); d3 = fmaf(q4.w, k4.w, d3);
            }
            const float sv = krow < n ? quad_sum((d0 + d1) + (d2 + d3)) * P.scale : -INFINITY;
            // this wavefront's 16 positions: Softmax pieces go/quant.go:610-626 (max-subtract, f32(exp(f64)), f32 sum)
            const float mw = wave_max_f32(sv);
            const float p = krow < n ? exp_f64_as_f32(sv - (mw == -INFINITY ? 0.f : mw)) : 0.f;
            const float lw = wave_sum_f32(kq == 0 ? p : 0.f);
            // P*V: the probabilities of rows vg + 4 kk come from the lanes that hold those rows' scores
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 vc4 = *reinterpret_cast<const float4 *>(vcur + vc * 4);
#pragma unroll
            for (int kk = 0; kk < NV; kk++) {
                const float pw = __shfl(p, (vg + 4 * kk) * 4);
                const bool vcurrow = t0 + wave * 16 + vg + 4 * kk == pos;
                const float4 v4 = vcurrow ? vc4 : vreg[kk];       // (a masked row's V may be stale but is finite, and its p is 0)
                o.x = fmaf(pw, v4.x, o.x); o.y = fmaf(pw, v4.y, o.y); o.z = fmaf(pw, v4.z, o.z); o.w = fmaf(pw, v4.w, o.w);
            }
            o.x += __shfl_xor(o.x, 16); o.y += __shfl_xor(o.y, 16); o.z += __shfl_xor(o.z, 16); o.w += __shfl_xor(o.w, 16);
            o.x += __shfl_xor(o.x, 32); o.y += __shfl_xor(o.y, 32); o.z += __shfl_xor(o.z, 32); o.w += __shfl_xor(o.w, 32);
            if (lane < 16) *reinterpret_cast<float4 *>(wpart + wave * 68 + 4 + vc * 4) = o;
            if (lane == 0) { wpart[wave * 68] = mw; wpart[wave * 68 + 1] = lw; }
        }
        __syncthreads();
        if (ch == 0) BLK_STAMP(13);
        if (wave == 0) {
            // the eight partials of the pass, merged in a fixed order.  The weights exp(m_w - M) are this engine's own
            // construct -- the reference has one softmax over all positions -- and use the f32 exponential, as the merge of
            // the passes below and the split merge of the five-launch plan do (load_x4<PRO_ATTN>).  (Round 4 tried the
            // reference's float32(exp(float64)) here, go/quant.go:619: only eight lanes compute it, but they are wavefront 0
            // on the launch's critical path -- this step went from 850 to 1200 cycles, 1 % of nano's token; the logits of the
            // fused block are held to the oracle's within 1e-4 either way, test_fused_attention_block_matches_oracle)
            const float mw = lane < 8 ? wpart[min(lane, 7) * 68] : -INFINITY;
            const float M = wave_max_f32(mw);
            const float wgt = (lane < 8 && mw != -INFINITY) ? __expf(mw - M) : 0.f;   // (an empty wavefront weighs 0)
            const float L = wave_sum_f32(lane < 8 ? wgt * wpart[min(lane, 7) * 68 + 1] : 0.f);
            float ov = 0.f;
#pragma unroll
            for (int w = 0; w < 8; w++) ov = fmaf(__shfl(wgt, w), wpart[w * 68 + 4 + lane], ov);
            chunk[ch * 66 + 2 + lane] = ov;
            if (lane == 0) { chunk[ch * 66] = M; chunk[ch * 66 + 1] = L; }
        }
        if (ch == 0) BLK_STAMP(14);
        if (ch + 1 < nch) __syncthreads();   // wpart is rewritten by the next pass
    }
    BLK_STAMP(9);
    if (tid < HD) {   // merge the passes exactly as the WO prologue of the five-launch plan does (load_x4<PRO_ATTN>)
        if (nch == 1) on[tid] = chunk[2 + tid] * (1.0f / chunk[1]);
        else {
            float M = chunk[0];
            for (int c = 1; c < nch; c++) M = fmaxf(M, chunk[c * 66]);
            float v = 0.f, L = 0.f;
            for (int c = 0; c < nch; c++) {
                const float w = exp_f64_as_f32(chunk[c * 66] - M);
                L += w * chunk[c * 66 + 1];
                v += w * chunk[c * 66 + 2 + tid];
            }
            on[tid] = v * (1.0f / L);
        }
    }
    __syncthreads();
    BLK_STAMP(10);

    // ---- this member's rows of the head's 64 columns of WO (go/model.go:590): partial [D/4], added by the consumer ----
    if (tid < wo_rows * 2) {
        float v = BlockDot<WT>::run(wc, wd16, on + wblk * 32);
        v += dpp_f32<DPP_QUAD_XOR1>(v);     // the two blocks of a row sit in adjacent lanes
        if (P.bias_out && h == 0) v += P.bias_out[wrow];
        if (wblk == 0) P.parts[(long long)h * D + wrow] = v;
    }
    BLK_STAMP(11);
#undef BLK_STAMP
}


// ====================================================================================================================
// The feed-forward half of a layer (go/model.go:597-612) as ONE launch, same idea: gate / up and down are all-to-all
// over the hidden width I only because down reads every h.  Cut I into slices of 256 columns: a cluster of eight
// workgroups (blocks b, b+8, ..., b+56: one XCD as observed, for speed only) owns a slice -- each member projects 32 gate and 32 up rows of it
// (four 16-row tiles x up to four 256-column groups = 16 wavefronts), applies SiLU(gate) * up, the members exchange
// their 32 values of h through tagged granules (256 per cluster), and each member multiplies its eighth of the rows of
// W_down restricted to the slice's 256 columns (a second packed copy of W_down, sliced by column) -- a partial [D / 8]
// vector per member, one partial [D] vector per slice.  The I / 256 partial vectors are added to the residual stream
// by the next consumer's prologue (the next layer's attention block, or the LM head), like the attention block's.
// nano: gate/up 4.3 us + down 3.3 us as two launches -> one launch.

constexpr int FFN_THREADS = 1024;    // 16 wavefronts: {gate tile 0, gate tile 1, up tile 0, up tile 1} x 4 column groups
constexpr int FFN_MEMBERS = 8;
constexpr int FFN_SLICE = 256;       // hidden columns per cluster
constexpr int FFN_MAX_PARTS = 8;     // partial vectors a consumer adds: I <= 2048

struct FfnParams {
    const uint8_t *gate_q, *up_q;    // packed gate / up [I rows x D]
    const uint32_t *gate_s, *up_s;
    const uint8_t *dn_q;             // W_down sliced by 256 columns: [I / 256][D / 16 tiles][4 pairs]
    const uint32_t *dn_s;
    int D, I, npairs;                // npairs of gate / up rows (D / 64)
    const float *x, *normw;          // residual stream and ffn_norm
    const float *parts_in;           // [nparts_in][D] partial vectors to add to x first (the attention block's)
    int nparts_in;
    float *x_out;                    // x + sum parts_in, stored by cluster 0 / member 0 (the residual the consumer adds to)
    float eps;
    float *parts_out;                // [I / 256][D]
    unsigned long long *xchg;        // [I] granules
    const unsigned *tick;
    unsigned layer_tag;
    unsigned *status, *host_status;
    int spin_limit;                  // polls before a wavefront gives up (see BlockParams)
    long long *dbg;                  // optional phase stamps of block 0 (nl_debug_stamps)
};

__host__ __device__ constexpr size_t ffn_lds_bytes() {
    return sizeof(float) * (size_t)(4 * XS_WAVE + 16 * TR + KL * XS_PAIR) + 16 * sizeof(double);
}
inline int ffn_grid(int clusters) { return ((clusters + 7) / 8) * 64; }   // block b: cluster (b/64)*8 + b%8, member (b/8)%8

template <int WT, int NPIN>
__global__ void __launch_bounds__(FFN_THREADS) ffn_block_kernel(FfnParams P) {
    constexpr int CPP = WTraits<WT>::CPP, NW = FFN_THREADS / 64;
    NL_KARGS8(P.gate_q, P.up_q, P.gate_s, P.up_s, P.dn_q, P.dn_s, P.x, P.normw);
    NL_KARGS8(P.parts_in, P.x_out, P.parts_out, P.xchg, P.tick, P.status, P.D, P.I);
    NL_KARGS4(P.npairs, P.nparts_in, P.layer_tag, P.eps);
    NL_KARGS4(P.dbg, P.host_status, P.spin_limit, P.nparts_in);
    const int cl = (blockIdx.x >> 6) * 8 + (blockIdx.x & 7), mem = (blockIdx.x >> 3) & 7;
    if (cl * FFN_SLICE >= P.I) return;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *dred = reinterpret_cast<double *>(smem);                 // [16]
    float *xs = reinterpret_cast<float *>(dred + 16);                // [4][XS_WAVE]: x * g per 256-column group
    float *red = xs + 4 * XS_WAVE;                                   // [16][16]
    float *hs = red + NW * TR;                                       // [4][XS_PAIR] h of this slice, padded pairs

    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int r = lane >> 2, k = lane & 3, D = P.D;
    const int tsel = wave >> 2, grp = wave & 3;      // 0,1: gate tiles; 2,3: up tiles of the same 32 rows
    const int ngroups = (P.npairs + KL - 1) / KL;
    const unsigned tag = (__hip_atomic_load(P.tick, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << 8) | P.layer_tag;
    const int tile = (cl * FFN_SLICE + mem * 32) / TR + (tsel & 1);
    const uint8_t *const Wq = tsel < 2 ? P.gate_q : P.up_q;
    const uint32_t *const Ws = tsel < 2 ? P.gate_s : P.up_s;
#define FFN_STAMP(i) do { if (P.dbg && blockIdx.x == 0 && tid == 0) P.dbg[i] = clock64(); } while (0)
    FFN_STAMP(0);

    // ---- loads: x + the predecessor's partial vectors + norm weights of this wavefront's 256 columns, then its weights ----
    const int gg = min(grp, ngroups - 1), gs = min(KL, P.npairs - gg * KL);
    const bool lv = grp < ngroups && k < gs;
    const int xcol = gg * (KL * PAIR) + lane * 4;
    const bool xin = xcol < D;
    float4 xv, gv, pv[NPIN];     // (see attn_block_kernel: unfilled values are never consumed unmasked)
    uint4 cw[CPP];
    uint2 sw;
    undef_regs(xv); undef_regs(gv); undef_regs(sw);
#pragma unroll
    for (int j = 0; j < CPP; j++) undef_regs(cw[j]);
#pragma unroll
    for (int p = 0; p < NPIN; p++) undef_regs(pv[p]);
    if (grp < ngroups) {
        if (tsel == 0) {   // the wavefronts of tile 0 stage x for the whole workgroup (one read of x + partials, not four)
            const unsigned xo = (unsigned)(xin ? xcol : 0) * 4u;
            xv = ld_off<float4>(P.x, xo);
            gv = ld_off<float4>(P.normw, xo);
#pragma unroll
            for (int p = 0; p < NPIN; p++) pv[p] = ld_off<float4>(P.parts_in, (unsigned)(min(p, P.nparts_in - 1) * D) * 4u + xo);
        }
        load_pair<WT>(Wq, Ws, (long long)tile * P.npairs, gg, gs, r, min(k, gs - 1), cw, sw);
    }

    FFN_STAMP(1);
    // ---- residual + RMSNorm scaling into wave-private LDS; sum of squares from the wavefronts of tile 0 ----
    float *xw = xs + grp * XS_WAVE;      // shared by the four wavefronts of a column group
    double ss = 0.0;
    if (tsel == 0) {
        float4 xa = xv;
#pragma unroll
        for (int p = 0; p < NPIN; p++) {
            const bool on = p < P.nparts_in;
            xa.x += on ? pv[p].x : 0.f; xa.y += on ? pv[p].y : 0.f; xa.z += on ? pv[p].z : 0.f; xa.w += on ? pv[p].w : 0.f;
        }
        if (!xin || grp >= ngroups) xa = make_float4(0.f, 0.f, 0.f, 0.f);
        ss = fma((double)xa.x, (double)xa.x, ss); ss = fma((double)xa.y, (double)xa.y, ss);
        ss = fma((double)xa.z, (double)xa.z, ss); ss = fma((double)xa.w, (double)xa.w, ss);
        if (cl == 0 && mem == 0 && xin && grp < ngroups) *reinterpret_cast<float4 *>(P.x_out + xcol) = xa;   // the updated residual stream
        xa.x *= gv.x; xa.y *= gv.y; xa.z *= gv.z; xa.w *= gv.w;
        *reinterpret_cast<float4 *>(xw + (lane >> 4) * XS_PAIR + (lane & 15) * 4) = xa;
    }
    ss += dpp_f64<DPP_QUAD_XOR1>(ss);
    ss += dpp_f64<DPP_QUAD_XOR2>(ss);
    ss += dpp_f64<DPP_HALF_MIRROR>(ss);
    ss += dpp_f64<DPP_ROW_MIRROR>(ss);
    if (tsel == 0 && (lane & 15) == 0) dred[grp * 4 + (lane >> 4)] = ss;
    FFN_STAMP(2);
    __syncthreads();
    FFN_STAMP(3);
    // epilogue threads: 32 lanes of the last wavefront (idle during the dot products unless D has four column groups)
    constexpr int ET0 = FFN_THREADS - 64;
    const int et = tid - ET0;
    const bool e_thr = (unsigned)et < 32u;
    float inv = 0.f;
    if (e_thr) {
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < 16; w++) tot += dred[w];
        inv = (float)(1.0 / sqrt(tot / (double)D + (double)P.eps));
    }
    float acc = PairDot<WT>::run(cw, sw, xw + k * XS_PAIR, 0.f);
    acc = lv ? acc : 0.f;
    acc = quad_sum(acc);
    if (k == 0) red[wave * TR + r] = acc;
    FFN_STAMP(4);
    __syncthreads();
    FFN_STAMP(5);

    // this member's rows of the slice's W_down columns stream in while the exchange runs: lane = (row, pair of the slice)
    // (lane = (row, pair of the slice, block of the pair): eight lanes per row, each 32 of the 256 columns)
    const int dn_rows = D / FFN_MEMBERS, dn_tiles = D / TR;
    const int drow = mem * dn_rows + min(tid >> 3, dn_rows - 1), dpair = (tid >> 1) & 3, dblk = tid & 1;
    uint4 dc[CPP / 2];
    uint32_t dd16;
    load_block<WT>(P.dn_q, P.dn_s, ((long long)cl * dn_tiles + (drow >> 4)) * KL, 0, KL, drow & 15, dpair, dblk, dc, dd16);

    // ---- h = SiLU(gate) * up (go/quant.go:629-631, go/model.go:604-606) for this member's 32 rows; publish ----
    if (e_thr) {
        const int t16 = et >> 4, rr = et & 15;
        const float *rg = red + (t16 * 4) * TR, *ru = red + ((2 + t16) * 4) * TR;
        const float g = (((rg[rr] + rg[TR + rr]) + rg[2 * TR + rr]) + rg[3 * TR + rr]) * inv;
        const float u = (((ru[rr] + ru[TR + rr]) + ru[2 * TR + rr]) + ru[3 * TR + rr]) * inv;
        const float ex = exp_f64_as_f32(-g);
        const float h = (g / (1.0f + ex)) * u;
        __hip_atomic_store(P.xchg + (size_t)cl * FFN_SLICE + mem * 32 + et, ((unsigned long long)tag << 32) | __float_as_uint(h),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    FFN_STAMP(6);
    if (tid < FFN_SLICE) {
        const bool dead = __hip_atomic_load(P.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        unsigned long long gq;
        for (int spins = 0;; spins++) {
            gq = __hip_atomic_load(P.xchg + (size_t)cl * FFN_SLICE + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all((unsigned)(gq >> 32) == tag)) break;
            if (dead || spins >= P.spin_limit) { if (lane == 0) { atomicOr(P.status, 16u); *P.host_status = 16u; } break; }
            __builtin_amdgcn_s_sleep(1);
        }
        hs[(tid >> 6) * XS_PAIR + (tid & 63)] = __uint_as_float((unsigned)gq);
    }
    FFN_STAMP(7);
    __syncthreads();
    FFN_STAMP(8);

    // ---- this member's rows of W_down over the slice's 256 columns (go/model.go:609-612): a partial [D / 8] vector ----
    if (tid < dn_rows * 8) {
        float v = BlockDot<WT>::run(dc, dd16, hs + dpair * XS_PAIR + dblk * 32);
        v = quad_sum(v);                        // lanes 8i .. 8i+7 hold a row: two quads
        v += dpp_f32<DPP_HALF_MIRROR>(v);       // (lane 7 - i of the 8: the other quad's sum)
        if ((tid & 7) == 0) P.parts_out[(size_t)cl * D + drow] = v;
    }
    FFN_STAMP(9);
#undef FFN_STAMP
}

}  // namespace nl
