// nl_tp.h -- a tensor-parallel rank's decoder layer as TWO launches (go/model.go:517-594 and :597-612).
//
// A rank of a tp 4 / tp 8 group holds 12-25 MB of a 7.9B layer: 2.4-5 us of HBM time spread over the four dependent
// launches of the wide-tier plan (projection + attention, WO, gate || up, down), each of which costs 4.5-7 us whatever it
// streams.  The shard is a small-tier problem, so it gets the small-tier answer -- two launches per layer:
//
//   tp_attn_kernel   RMSNorm -> this rank's Q | K | V tiles (chip-wide spread, as nl_group.h) -> RoPE -> cluster exchange
//                    -> GQA attention per query head -> the heads' normalised outputs published as granules -> every
//                    workgroup multiplies its rows of this rank's WO column slice (weights requested at entry: they
//                    stream in while the attention runs) -> push all-reduce finished by the row's owner lane.
//   tp_ffn_kernel    PRODUCER workgroups: RMSNorm -> one gate / up tile each (or a gate + up tile pair, SwiGLU applied)
//                    -> published as granules.  CONSUMER workgroups: request their rows of this rank's W_down column
//                    slice at entry, gather h, multiply -> push all-reduce finished by the row's owner lane.
//
// Every in-launch hand-off is cdna_hip_programming.md Guideline 16 form R2 (8-byte {tag, value} granules, relaxed
// agent-scope stores and polls, the value is its own flag); tag = (forward counter << 8) | layer, the counter advanced by
// the embedding launch, so a replayed graph needs no per-launch argument.  Producers never wait, so the launches cannot
// deadlock on their own; every poll is bounded and a give-up is reported like the other fused launches' (nl_block.h).
// The all-reduce tail is the one of the GEMV's EPI_P2P (nl_kernels.h, nl_p2p.h): granules into every rank's receive slot,
// the row's owner adds the G partials in rank order.  In an in-process shard group (NL_FLAG_LOCAL_GROUP) the tail stores
// the rank's partial instead and nl_group_forward adds the shards' partials in the same order: bitwise the same result.
#pragma once
#include "nl_kernels.h"
#include "nl_group.h"
#include "nl_p2p.h"

namespace nl {

#ifdef NL_TP_STAMPS
// developer build (tools/tp_stamps.sh): wall-clock (100 MHz) phase stamps of a few workgroups of layer NL_TP_STAMPS's launches
__device__ long long g_tp_stamps[8][16];
__device__ long long g_tp_census[2][512][2];     // [launch kind][block][entry, exit] of the stamped layer
#define TP_STAMP(slot, i) do { if ((slot) >= 0 && threadIdx.x == 0) g_tp_stamps[(slot)][(i)] = wall_clock64(); } while (0)
#define TP_CENSUS(kind, layer_tag, which) do { if ((layer_tag) == NL_TP_STAMPS && threadIdx.x == 0 && blockIdx.x < 512) g_tp_census[(kind)][blockIdx.x][(which)] = wall_clock64(); } while (0)
#else
#define TP_STAMP(slot, i) do { } while (0)
#define TP_CENSUS(kind, layer_tag, which) do { } while (0)
#endif

constexpr int TP_THREADS = 1024;
constexpr int TP_NCH_MAX = 4;        // passes a head's attention may take inside the launch on ONE block
constexpr int TP_REC_MAX = 8;        // records a runner merges: its own passes + the helpers' (with helpers a head covers up to 4 x 2 passes)
constexpr int TP_PASS = 256;         // positions per pass: ALL sixteen wavefronts of the runner hold 16 cache rows each (the first
                                     // version used eight, 128 positions per pass, and left the other eight idle: at position 470
                                     // four dependent passes of ~1.7 us per layer instead of two).  fused_max_pos <= 4 x 256

// One all-reduce seam as the tail of a launch sees it.
struct TpSeam {
    u64 *dst[8];            // rank r's receive slot for THIS rank's partial (seam parity applied)
    const u64 *slots;       // this rank's receive slots of the seam: [n][rows]
    const unsigned *epoch;  // forward counter (tag = epoch << 8 | seam)
    unsigned *status;       // set non-zero when a poll gave up (host: NL_ERR_COMM)
    long long timeout;      // wall_clock64 ticks
    int n;                  // ranks; 0 = in-process group: store the partial to `partial` and return; -1 = no tensor
                            // parallelism at all (one GPU holds the whole layer): out[row] = resid + v
    unsigned seam;
    int rows;               // D
    float *partial;         // n == 0: [rows]
};

// out[row] = resid + sum over ranks (rank order) of the partials of `row`; this lane owns the row.
__device__ __forceinline__ void tp_allreduce_row(const TpSeam &S, unsigned e_tag, int row, float v, float resid, float *out) {
    if (S.n == 0) { S.partial[row] = v; return; }
    if (S.n < 0) { out[row] = resid + v; return; }
    const u64 gran = ((u64)e_tag << 32) | __float_as_uint(v);
#pragma unroll
    for (int pr = 0; pr < 8; pr++)
        if (pr < S.n) __hip_atomic_store(S.dst[pr] + row, gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const bool dead = __hip_atomic_load(S.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    u64 g[8];
    const long long t0 = wall_clock64();
    for (;;) {
        bool ok = true;
#pragma unroll
        for (int pr = 0; pr < 8; pr++)
            g[pr] = __hip_atomic_load(S.slots + (size_t)min(pr, S.n - 1) * S.rows + row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
        for (int pr = 0; pr < 8; pr++) ok &= (unsigned)(g[pr] >> 32) == e_tag;
        if (ok) break;
        if (dead || wall_clock64() - t0 > S.timeout) { atomicOr(S.status, 1u); break; }
        __builtin_amdgcn_s_sleep(2);
    }
    float sum = 0.f;
#pragma unroll
    for (int pr = 0; pr < 8; pr++) sum += pr < S.n ? __uint_as_float((unsigned)g[pr]) : 0.f;   // fixed rank order
    out[row] = resid + sum;
}

// ---- 16-byte granules {tag, v0, v1, v2}: ONE dwordx4 write-through store each.  That a 16-byte aligned dwordx4 store is seen
// whole or not at all by a dwordx4 load is an ASSUMED hardware property (the ISA documents no single-copy atomicity beyond 8
// bytes; one lane's 16 bytes travel as one request to one cache line): soaked by gran16_soak_kernel below -- 1e8+ cross-XCD
// reads per run in tests/test_gpu_tp_fused.py, never a mixed copy -- and every consumer validates the tag of every granule.  Measured
// (tools/allgather_probe.hip, profiles/r04_allgather_probe.log): an all-gather of 2752 floats to 256 workgroups costs 1.77 us
// from the last publish with these against 2.05 us with 8-byte granules and 2.6-4.5 us with flag + payload forms.
// A TILE of 16 values (one 16-row projection tile, a quarter of a head) is six granules: granule j holds values 3j .. 3j+2.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int GPT = 6;      // granules per 16-value tile

__device__ __forceinline__ void gran16_store(u32x4 *p, unsigned tag, float a, float b, float c) {
    const u32x4 v = {tag, __float_as_uint(a), __float_as_uint(b), __float_as_uint(c)};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
}
// the calling wavefront (all 64 lanes active) publishes ntile tiles: lane l holds value l & 15 of tile l >> 4
__device__ __forceinline__ void gran16_publish(u32x4 *tile0, int ntile, unsigned tag, float v, int lane) {
    const int t = lane / GPT, j = lane - t * GPT;                 // lanes 0 .. 6 ntile - 1 store one granule each
    const int src = min(t, 3) * 16 + 3 * j;
    const float a = __shfl(v, src), b = __shfl(v, min(src + 1, 63)), c = __shfl(v, min(src + 2, 63));
    if (t < ntile) gran16_store(tile0 + t * GPT + j, tag, a, b, c);
}
__device__ __forceinline__ u32x4 gran16_load(const u32x4 *p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void gran16_load5(const u32x4 *p0, const u32x4 *p1, const u32x4 *p2, const u32x4 *p3, const u32x4 *p4,
                                             u32x4 &a, u32x4 &b, u32x4 &c, u32x4 &d, u32x4 &e) {
    asm volatile("global_load_dwordx4 %0, %5, off sc1\n\tglobal_load_dwordx4 %1, %6, off sc1\n\tglobal_load_dwordx4 %2, %7, off sc1\n\t"
                 "global_load_dwordx4 %3, %8, off sc1\n\tglobal_load_dwordx4 %4, %9, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d), "=&v"(e) : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4) : "memory");
}
__device__ __forceinline__ void gran16_load2(const u32x4 *p0, const u32x4 *p1, u32x4 &a, u32x4 &b) {
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b) : "v"(p0), "v"(p1) : "memory");
}
__device__ __forceinline__ void gran16_load4(const u32x4 *p0, const u32x4 *p1, const u32x4 *p2, const u32x4 *p3, u32x4 &a, u32x4 &b, u32x4 &c, u32x4 &d) {
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\tglobal_load_dwordx4 %2, %6, off sc1\n\t"
                 "global_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
}

// ---- warming the NEXT launch's first weight bytes (round 5; tools/l2_prefetch_probe.hip, profiles/r05_l2_prefetch_probe.log) ----
// A launch of the wide tier starts cold: its first weight round (18.9 MB of gate || up, 14.2 MB of Q | K | V) costs the 2 us of a
// first fetch plus 3 us of transfer while nothing else can run -- and the launch before it leaves HBM idle for longer than that
// (the attention chain; the h gather and the tail).  So wavefronts that are waiting anyway read ONE dword of every 128-byte
// line the SAME-numbered block of the next launch will request first: block b runs on XCD b % 8 in both launches (dispatch
// order; an assumption that costs nothing but the benefit when it fails), the lines land in that XCD's L2, and the next
// launch's first round is L2 hits (probe: 27.5 MB re-read 2.1 us faster than cold; 37 MB -- beyond the 32 MB of L2 -- 3.4 us).
// A touch is an ordinary load whose result is only consumed at the end of the kernel (pf_done).  A wavefront's loads return
// in order, so only wavefronts whose NEXT load comes after the touched bytes have landed may touch.
struct PfTiles {
    const uint8_t *q[2];             // packed quants of up to two matrices (gate, up) -- or one (Q | K | V: q[1] unused)
    const uint32_t *s[2];
    unsigned tile_qbytes, tile_sbytes;    // bytes of one 16-row tile's quants / scales (contiguous per tile)
    int ntiles;                      // tiles per matrix
    int nmat;                        // 0 = off
    int blocks;                      // grid of the launch that will read them
};
__device__ __forceinline__ unsigned pf_lines(const PfTiles &T) { return (T.tile_qbytes + 127u) / 128u + (T.tile_sbytes + 127u) / 128u; }
__device__ __forceinline__ unsigned pf_touch(const PfTiles &T, int mat, int tile, unsigned l) {
    const unsigned ql = (T.tile_qbytes + 127u) / 128u;
    const uint8_t *p = l < ql ? T.q[mat] + (size_t)tile * T.tile_qbytes + min(l * 128u, T.tile_qbytes - 4u)
                              : reinterpret_cast<const uint8_t *>(T.s[mat]) + (size_t)tile * T.tile_sbytes + min((l - ql) * 128u, T.tile_sbytes - 4u);
    return *reinterpret_cast<const unsigned *>(p);
}
__device__ __forceinline__ void pf_done(unsigned a, unsigned b) { asm volatile("" :: "v"(a), "v"(b)); }
// what a block of tp_attn_kernel requests at entry (its projection tiles), for the launch in front of it
struct PfQkv {
    PfTiles T;                       // q[0] / s[0]: the packed [q; k; v] rows
    int tpm, members, gqa, n_q_heads, n_kv_heads;
    unsigned m8_inv, m_inv;
};

// Every thread with gi < ng spins on granule gi until its tag matches (a wavefront leaves together); bounded.
__device__ __forceinline__ u32x4 gran16_wait(const u32x4 *src, int gi, int ng, unsigned tag, unsigned *status, unsigned *host_status,
                                             int spin_limit, unsigned code, bool dead) {
    u32x4 g;
    const int lane = threadIdx.x & 63;
    for (int spins = 0;; spins++) {
        g = gran16_load(src + min(gi, ng - 1));
        if (__all(g.x == tag)) break;
        if (dead || spins >= spin_limit) { if (lane == 0) { atomicOr(status, code); *host_status = code; } break; }
        __builtin_amdgcn_s_sleep(1);
    }
    return g;
}

// All-gather of `ntile` 16-value tiles (and, TWO, of a second array of as many: dst = SiLU(first) * second, go/quant.go:629-631,
// go/model.go:604-606) into LDS as padded 64-column pairs, by every thread of the workgroup.  Two stages, because a polling
// workgroup costs the streaming ones fabric bandwidth (tp_ffn_kernel with every wavefront polling: +1 us on the producers'
// weight fetch): ONE wavefront spins on the first granule of <= 16 tiles spread over the array while the others sit at the
// barrier, then every granule is fetched once, a thread's loads in flight together, every tag checked -- the sampled tiles
// being visible promises nothing about the rest, so the sweep retries.  NR: granules per thread and array.
template <int NR, bool TWO, bool PROBE>
__device__ __forceinline__ void tp_gather16(const u32x4 *src, const u32x4 *src2, int ntile, unsigned tag, float *dst, float *dst2, int nvals,
                                            unsigned *status, unsigned *host_status, int spin_limit, unsigned code) {
    const int tid = threadIdx.x, lane = tid & 63;
    const bool dead = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    if (PROBE) {
        if (tid < 64) {
            const int np = min(ntile, 16), pl = lane & 15;
            const int pt = min((pl * ntile) / np, ntile - 1);              // lanes 0-15: array one; 16-31: array two (TWO); the rest repeat
            const u32x4 *pp = ((TWO && (lane & 16)) ? src2 : src) + (size_t)pt * GPT;
            for (int spins = 0;; spins++) {
                const u32x4 g = gran16_load(pp);
                if (__all(g.x == tag)) break;
                if (dead || spins >= spin_limit) { if (lane == 0) { atomicOr(status, code); *host_status = code; } break; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
    }
    const int ng = ntile * GPT;
    static_assert(NR == 1 || NR == 2, "granules per thread");
    u32x4 ga[NR], gb[NR];
    if ((tid & ~63) < ng) {           // this wavefront holds granules
        const bool second = (tid & ~63) + TP_THREADS < ng;      // ... and a second one per lane (wave-uniform: lanes past the end re-read the last granule)
        for (int spins = 0;; spins++) {
            const u32x4 *p0 = src + min(tid, ng - 1), *p1 = src + min(tid + TP_THREADS, ng - 1);
            if (TWO) {
                const u32x4 *q0 = src2 + min(tid, ng - 1), *q1 = src2 + min(tid + TP_THREADS, ng - 1);
                if (NR == 1 || !second) gran16_load2(p0, q0, ga[0], gb[0]);
                else gran16_load4(p0, p1, q0, q1, ga[0], ga[NR - 1], gb[0], gb[NR - 1]);
            } else {
                if (NR == 1 || !second) ga[0] = gran16_load(p0);
                else gran16_load2(p0, p1, ga[0], ga[NR - 1]);
            }
            bool ok = ga[0].x == tag && (!TWO || gb[0].x == tag);
            if (NR == 2 && second) ok = ok && ga[NR - 1].x == tag && (!TWO || gb[NR - 1].x == tag);
            if (__all(ok)) break;
            if (dead || spins >= spin_limit) { if (lane == 0) { atomicOr(status, code); *host_status = code; } break; }
            __builtin_amdgcn_s_sleep(PROBE ? 1 : 3);
        }
#pragma unroll
        for (int q = 0; q < NR; q++) {
            const int gi = tid + q * TP_THREADS;
            if (gi < ng) {
                const int t = gi / GPT, j = gi - t * GPT;
                const unsigned va[3] = {ga[q].y, ga[q].z, ga[q].w}, vb[3] = {gb[q].y, gb[q].z, gb[q].w};
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const int i = t * 16 + 3 * j + c;
                    if (3 * j + c < 16 && i < nvals) {
                        dst[(i >> 6) * XS_PAIR + (i & 63)] = __uint_as_float(va[c]);
                        if (TWO) dst2[i] = __uint_as_float(vb[c]);
                    }
                }
            }
        }
    }
    if (TWO) {
        // SiLU(g) * u (go/quant.go:629-631, go/model.go:604-606) with the float64 exponentials spread over all threads
        __syncthreads();
        for (int i = tid; i < nvals; i += TP_THREADS) {
            float *d = dst + (i >> 6) * XS_PAIR + (i & 63);
            const float g = *d;
            *d = (g / (1.0f + exp_f64_as_f32(-g))) * dst2[i];
        }
    }
}

// ---- Q4_0 dot products on the matrix pipe (round 5; profiles/r05_fake_dot.log: with a quarter of the vector instructions of
// the dot products the 7.9B step drops from 1.43 to 1.23 ms -- the decode GEMV of a wide layer is NOT purely memory-bound on
// this chip: 2.1 vector operations and 4 bytes of LDS per weight leave 6.5 us of a 35 us layer exposed) ----------------------
// The arithmetic is nl_persist.h's: a 32-element block of the input is scaled by a power of two so that its largest element
// fits 31 bits, rounded to integers X (2^-30 of the block maximum per element) and written as four SIGNED base-256 digits;
// v_mfma_i32_4x4x4_16b_i8 multiplies the digit rows (A) with int8 weights (B) exactly in int32.  Sixteen independent 4x4x4
// products per instruction: a QUAD of lanes = four weight ROWS of one block, so the fused launches read a second copy of their
// matrices in which the 16-byte chunks of every (16-row tile, 256-column group) are permuted (mf_permute_kernel): lane
// l = 4 q + j, chunk slot s <- row 4 (q & 3) + j, block (q >> 2) + 4 s of the group.  A Q4_0 weight is (n - 8) d: the products use
// the nibbles n as they are and the block's -8 sum(X) is added once per unit.
typedef int mf_i32x4 __attribute__((ext_vector_type(4)));
constexpr int MF_GROUP_BYTES = 8 * 128 + 8 * 8;      // image of one 256-column group: 8 blocks x [4 digits][32 bytes], then {1/scale, offset} per block
static_assert(MF_GROUP_BYTES == XS_WAVE * 4, "a group's digit image replaces its float staging area");

__device__ __forceinline__ float mf_rows4_sum(float v) {      // lanes l, l + 16, l + 32, l + 48 (nl_batch.h rows4_sum)
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
template <int CTRL> __device__ __forceinline__ int mf_dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }

// This lane's four consecutive elements v (elements 4 e4 .. 4 e4 + 3 of a block whose eight lanes are adjacent and all active)
// -> the block's digit image blk_img[m][32] and {1 / scale, -8 sum(X) / scale}
__device__ __forceinline__ void mf_digits4(float4 v, unsigned char *blk_img, float2 *bs, int e4) {
    float a = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    a = fmaxf(a, dpp_f32<DPP_QUAD_XOR1>(a));
    a = fmaxf(a, dpp_f32<DPP_QUAD_XOR2>(a));
    a = fmaxf(a, dpp_f32<DPP_HALF_MIRROR>(a));                    // the block's 8 lanes
    unsigned ex = __float_as_uint(a) >> 23;
    ex = min(max(ex, 30u), 254u);
    const float scale = __uint_as_float((283u - ex) << 23);        // 2^(29 - (ex - 127)): |v * scale| < 2^30
    const int X0 = __float2int_rn(v.x * scale), X1 = __float2int_rn(v.y * scale), X2 = __float2int_rn(v.z * scale), X3 = __float2int_rn(v.w * scale);
    // the signed base-256 digits of X are the bytes of (X + 0x808080) ^ 0x808080 (digit + 128 never carries)
    const unsigned W0 = ((unsigned)X0 + 0x00808080u) ^ 0x00808080u, W1 = ((unsigned)X1 + 0x00808080u) ^ 0x00808080u;
    const unsigned W2 = ((unsigned)X2 + 0x00808080u) ^ 0x00808080u, W3 = ((unsigned)X3 + 0x00808080u) ^ 0x00808080u;
    // 4 x 4 byte transpose: word m = digit m of the four elements
    const unsigned l01 = __builtin_amdgcn_perm(W1, W0, 0x05010400u), h01 = __builtin_amdgcn_perm(W1, W0, 0x07030602u);
    const unsigned l23 = __builtin_amdgcn_perm(W3, W2, 0x05010400u), h23 = __builtin_amdgcn_perm(W3, W2, 0x07030602u);
    unsigned *p = reinterpret_cast<unsigned *>(blk_img) + e4;
    p[0] = __builtin_amdgcn_perm(l23, l01, 0x05040100u);
    p[8] = __builtin_amdgcn_perm(l23, l01, 0x07060302u);
    p[16] = __builtin_amdgcn_perm(h23, h01, 0x05040100u);
    p[24] = __builtin_amdgcn_perm(h23, h01, 0x07060302u);
    // sum of the block's X, exactly: 15 low bits and the rest summed apart (each below 2^21), joined in float32 with one rounding
    int sl = (X0 & 0x7fff) + (X1 & 0x7fff) + (X2 & 0x7fff) + (X3 & 0x7fff), sh = (X0 >> 15) + (X1 >> 15) + (X2 >> 15) + (X3 >> 15);
    sl += mf_dpp_i32<DPP_QUAD_XOR1>(sl); sh += mf_dpp_i32<DPP_QUAD_XOR1>(sh);
    sl += mf_dpp_i32<DPP_QUAD_XOR2>(sl); sh += mf_dpp_i32<DPP_QUAD_XOR2>(sh);
    sl += mf_dpp_i32<DPP_HALF_MIRROR>(sl); sh += mf_dpp_i32<DPP_HALF_MIRROR>(sh);
    if (e4 == 0) {
        const float inv = __uint_as_float((ex - 29u) << 23);       // 1 / scale
        *bs = make_float2(inv, -8.0f * fmaf((float)sh, 32768.0f, (float)sl) * inv);
    }
}

// One unit: the lane's 32 nibbles (row and block as mf_permute_kernel placed them) times the block's inputs -> d * sum (n - 8) x.
// img: the block's digit image; j = lane & 3 = the digit row this lane feeds the quad's products with.
__device__ __forceinline__ float mf_unit_q4(uint4 nib, unsigned d16, const unsigned char *blk_img, float2 bs, int j) {
    const uint4 *ap = reinterpret_cast<const uint4 *>(blk_img + j * 32);
    const uint4 a0 = ap[0], a1 = ap[1];
    const unsigned m = 0x0f0f0f0fu;
    mf_i32x4 acc = {0, 0, 0, 0};
#define MF_P(A, W) acc = __builtin_amdgcn_mfma_i32_4x4x4i8((int)(A), (int)(W), acc, 0, 0, 0)
    MF_P(a0.x, nib.x & m); MF_P(a0.y, nib.y & m); MF_P(a0.z, nib.z & m); MF_P(a0.w, nib.w & m);                          // elements 0 .. 15: the low nibbles
    MF_P(a1.x, (nib.x >> 4) & m); MF_P(a1.y, (nib.y >> 4) & m); MF_P(a1.z, (nib.z >> 4) & m); MF_P(a1.w, (nib.w >> 4) & m);  // 16 .. 31: the high ones
#undef MF_P
    const float f = fmaf(fmaf(fmaf((float)acc[3], 256.f, (float)acc[2]), 256.f, (float)acc[1]), 256.f, (float)acc[0]);
    return fmaf(f, bs.x, bs.y) * h2f_bits(d16);
}

// Two units of a lane (its two chunk slots: blocks b and b + 4 of a group) with their chains of eight dependent products
// interleaved: a 4x4x4 product has a few cycles of latency its successor in the chain would otherwise wait for.
__device__ __forceinline__ float mf_unit2_q4(const uint4 *nib, unsigned d16x2, const unsigned char *img0 /* block b's image */, const float2 *bs0, int j) {
    const uint4 *ap = reinterpret_cast<const uint4 *>(img0 + j * 32), *bp = reinterpret_cast<const uint4 *>(img0 + 4 * 128 + j * 32);
    const uint4 a0 = ap[0], a1 = ap[1], b0 = bp[0], b1 = bp[1];
    const float2 s0 = bs0[0], s1 = bs0[4];
    const unsigned m = 0x0f0f0f0fu;
    mf_i32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#define MF_P(ACC, A, W) ACC = __builtin_amdgcn_mfma_i32_4x4x4i8((int)(A), (int)(W), ACC, 0, 0, 0)
    MF_P(acc0, a0.x, nib[0].x & m); MF_P(acc1, b0.x, nib[1].x & m);
    MF_P(acc0, a0.y, nib[0].y & m); MF_P(acc1, b0.y, nib[1].y & m);
    MF_P(acc0, a0.z, nib[0].z & m); MF_P(acc1, b0.z, nib[1].z & m);
    MF_P(acc0, a0.w, nib[0].w & m); MF_P(acc1, b0.w, nib[1].w & m);
    MF_P(acc0, a1.x, (nib[0].x >> 4) & m); MF_P(acc1, b1.x, (nib[1].x >> 4) & m);
    MF_P(acc0, a1.y, (nib[0].y >> 4) & m); MF_P(acc1, b1.y, (nib[1].y >> 4) & m);
    MF_P(acc0, a1.z, (nib[0].z >> 4) & m); MF_P(acc1, b1.z, (nib[1].z >> 4) & m);
    MF_P(acc0, a1.w, (nib[0].w >> 4) & m); MF_P(acc1, b1.w, (nib[1].w >> 4) & m);
#undef MF_P
    // digits joined two by two in int32 (each pair below 2^25), then one float32 multiply-add
    const float f0 = fmaf((float)(acc0[2] + (acc0[3] << 8)), 65536.f, (float)(acc0[0] + (acc0[1] << 8)));
    const float f1 = fmaf((float)(acc1[2] + (acc1[3] << 8)), 65536.f, (float)(acc1[0] + (acc1[1] << 8)));
    return fmaf(f0, s0.x, s0.y) * h2f_bits(d16x2 & 0xffffu) + fmaf(f1, s1.x, s1.y) * h2f_bits(d16x2 >> 16);
}

// The permuted copy (see above): one thread per (tile, group, lane, slot) of a Q4_0 matrix whose rows are whole groups.
__global__ void mf_permute_kernel(const uint4 *q, const uint32_t *s, uint4 *q2, uint32_t *s2, long long ngroups_total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ngroups_total * 64) return;
    const long long grp = i >> 6;
    const int lane = (int)(i & 63), qd = lane >> 2, j = lane & 3;
    const int row = 4 * (qd & 3) + j;
    uint32_t sw = 0;
#pragma unroll
    for (int slot = 0; slot < 2; slot++) {
        const int blk = (qd >> 2) + 4 * slot, k = blk >> 1, c = blk & 1;      // source: pair k of the row, chunk c (nl_kernels.h load_pair)
        q2[grp * 128 + slot * 64 + lane] = q[grp * 128 + c * 64 + row * 4 + k];
        sw |= ((s[grp * 64 + row * 4 + k] >> (16 * c)) & 0xffffu) << (16 * slot);
    }
    s2[grp * 64 + lane] = sw;
}

// ------------------------------------------------------------------------------------------------ attention half ---

struct TpAttnParams {
    GroupParams G;               // projection + attention geometry, as qkv_attn_kernel (xchg / part_o / part_ml unused)
    const uint8_t *wo_q;         // this rank's WO column slice, packed [D / 16 tiles][Hs * 64 columns]
    const uint32_t *wo_s;
    int wo_npairs, wo_ntiles;
    int wo_gshift;               // log2 of the 256-column groups of a WO row (1, 2, 4, 8 or 16 groups; host-checked)
    int wo_tpw;                  // WO tiles per workgroup: block b owns tiles [b * tpw, b * tpw + tpw)
    int n_heads_local;           // Hs
    u32x4 *xq;                   // [kv groups][(G + 2) * 4 tiles][6] granules: q | k | v rows of this position, tile order
    u32x4 *xo;                   // [Hs][4 tiles][6] granules: the heads' normalised attention outputs
    const float *bias_out;
    float *x;                    // residual stream, read at entry (RMSNorm input, owner rows) and rewritten by the owners
    TpSeam seam;
    u32x4 *xp;                   // [Hs][3][22] granules: a helper block's (max, sum, sum p v[64]) record of ONE attention pass, for the head's runner
    int helpers;                 // 1: from the second 256-position pass on, the passes below the last run on three blocks that are not runners
    int live_grid;               // blocks [0, live_grid) belong to kv groups (grp_grid); the rest only multiply WO rows -- and help
    PfTiles pf;                  // round 0 of the feed-forward launch behind this one (gate, up tiles b of block b); nmat 0 = off
    PfTiles pf2;                 // the NEXT layer's projection tiles, for the memory-side Infinity Cache to keep (NL_PREFETCH bit 16; measured
                                 // and left off: 1.389 against 1.336 ms per token -- the extra lines push round 0 out of the L2s)
    int pf_early;                // 1: touched as soon as the block's own projection tiles are out (else: just before the WO gather)
};

__host__ __device__ constexpr size_t tp_attn_lds_bytes(int wo_npairs) {
    return 16 * sizeof(double) + sizeof(float) * (size_t)(16 * XS_WAVE + 16 * TR + 3 * 64 + 16 * 68 + TP_REC_MAX * 66 + wo_npairs * XS_PAIR + 16 * TR) +
           0;
}

// One role of the launch as straight-line code: every load below is unconditional (clamped addresses, masked uses), so
// hipcc's counted s_waitcnt in front of the first use of x is exact -- with loads behind role branches it assumes the
// shortest path and makes the RMSNorm wait for the weights, the dots for the cache rows and the WO tiles.
// LIVE: the block holds projection tiles; RUNNER: ... and runs the attention of one query head.
template <int WT, int NF, bool LIVE, bool RUNNER, bool MF, bool HELP>
__device__ __forceinline__ void tp_attn_body(const TpAttnParams &Q, char *smem, int cl, int mem) {
    const GroupParams &P = Q.G;
    constexpr int HD = 64, CPP = WTraits<WT>::CPP, R4 = HD / 4, NV = 4, NW = TP_THREADS / 64;
    double *dred = reinterpret_cast<double *>(smem);                 // [16]
    float *xs = reinterpret_cast<float *>(dred + 16);                // [16][XS_WAVE]
    float *red = xs + NW * XS_WAVE;                                  // [16][16]
    float *qs = red + NW * TR;                                       // [64]
    float *kcur = qs + 64, *vcur = kcur + 64;
    float *wpart = vcur + 64;                                        // [16][68]: a cache wavefront's (max, sum, -, -, sum p*v[64]) of the pass
    float *chunk = wpart + 16 * 68;                                  // [TP_REC_MAX][66]: (M, L, o[64]) per pass (own and helpers')
    float *ao = chunk + TP_REC_MAX * 66;                             // [wo_npairs][XS_PAIR]: every local head's output
    float *red2 = ao + Q.wo_npairs * XS_PAIR;                        // [16][16]
    static_assert(!MF || WT == WT_Q4_0, "matrix-pipe dot products: Q4_0");

    const int G = (int)P.gqa, D = P.D;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int r = lane >> 2, k = lane & 3;
    const int wpt = (int)P.wpt;
    const int slot = (int)udiv_by((unsigned)wave, P.wpt, P.wpt_inv), cs = wave - slot * wpt;
    auto tile_of = [&](int u, int &sect, int &hq, int &j) {
        j = u & 3;
        hq = u >> 2;
        sect = hq < G ? 0 : hq == G ? 1 : 2;
        return (sect == 0 ? cl * G + hq : sect == 1 ? P.n_q_heads + cl : P.n_q_heads + P.n_kv_heads + cl) * 4 + j;
    };
#ifdef NL_TP_STAMPS
    const int sslot = P.layer_tag != NL_TP_STAMPS ? -1 : (LIVE && cl == 0 && mem == 0) ? 0 : (LIVE && cl == 0 && mem == (int)P.gqa) ? 1
                      : (LIVE && cl == P.n_kv_heads - 1 && mem == P.members - 1) ? 2 : blockIdx.x == 7 ? 3 : -1;
#endif
    TP_STAMP(sslot, 0);

    // ---- every load of the role.  The forward counter and the position come through the scalar cache (written by earlier
    //      launches): a vector load of either would put a memory round trip in front of the weight requests ----
    const int pos = sload_i32(P.ctl + CTL_POS);
    const long long soff = P.single_stream ? 0 : (long long)sload_i32(P.ctl + CTL_STREAM) * P.kv_stream_stride;
    const unsigned tag = ((unsigned)sload_i32(reinterpret_cast<const int *>(P.tick)) << 8) | P.layer_tag;
    const unsigned e_tag = Q.seam.n > 0 ? ((unsigned)sload_i32(reinterpret_cast<const int *>(Q.seam.epoch)) << 8) | Q.seam.seam : 0u;
    const int ngroups = (P.npairs + KL - 1) / KL;
    float4 xv[NF], gv[NF];
    uint4 cw[NF][CPP];
    uint2 sw[NF];
    bool lv[NF], xin[NF];
    if (LIVE) {
        int w_sect, w_hq, w_j;
        const long long tp0 = (long long)tile_of(mem * P.tpm + slot, w_sect, w_hq, w_j) * P.npairs;
#pragma unroll
        for (int f = 0; f < NF; f++) {
            const int g = cs + f * wpt, gg = min(g, ngroups - 1);
            const int gs = min(KL, P.npairs - gg * KL);
            lv[f] = g < ngroups && k < gs;
            const int xcol = gg * (KL * PAIR) + lane * 4;
            xin[f] = xcol < D;
            xv[f] = ld_off<float4>(P.x, (unsigned)(xin[f] ? xcol : 0) * 4u);
            gv[f] = ld_off<float4>(P.normw, (unsigned)(xin[f] ? xcol : 0) * 4u);
        }
#pragma unroll
        for (int f = 0; f < NF; f++) {
            const int g = cs + f * wpt, gg = min(g, ngroups - 1);
            const int gs = min(KL, P.npairs - gg * KL);
            load_pair<WT>(P.qkv_q, P.qkv_s, tp0, gg, gs, r, min(k, gs - 1), cw[f], sw[f]);
        }
    }
    // epilogue inputs of the projection rows (threads 0 .. tpm * 16 - 1 use them; every thread requests a clamped copy)
    const int e_slot = min(tid >> 4, P.tpm - 1), e_rr = tid & 15;
    int e_sect = 0, e_hq = 0, e_j = 0;
    const bool e_act = LIVE && tid < P.tpm * TR;
    if (LIVE) tile_of(mem * P.tpm + e_slot, e_sect, e_hq, e_j);
    const int e_i = e_j * 8 + (e_rr & 7), e_e = e_i + (e_rr >> 3) * (HD / 2);
    float e_cos = 0.f, e_sin = 0.f, e_b = 0.f, e_bp = 0.f, e_bo = 0.f;
    if (LIVE) {
        e_cos = P.rope_cos[pos * (HD / 2) + e_i];
        e_sin = P.rope_sin[pos * (HD / 2) + e_i];
    }
    if (LIVE && P.bias_q) {   // (optional tensors, last: a pointer test in front of a load costs the loads after it their exact wait)
        const float *b = e_sect == 0 ? P.bias_q + (cl * G + e_hq) * HD : e_sect == 1 ? P.bias_k + cl * HD : P.bias_v + cl * HD;
        e_b = b[e_e];
        e_bp = b[e_e ^ (HD / 2)];
    }
    // ---- what the SECOND half of the launch reads -- this wavefront's share of WO: (tile, 256-column group) of the rows the
    //      BLOCK owns (clamped: a wavefront or block without a share repeats a neighbour's request and masks the result), the
    //      residual rows, the head's cache rows -- is requested by a LIVE block only after its projection dots: a wavefront
    //      issues in order and the compute unit ingests ~11 B/clk, so requested at entry these bytes delay the projection
    //      tile (the runner block's barrier came 1 us after its neighbours') ----
    const int ngw = 1 << Q.wo_gshift, tpw = Q.wo_tpw;
    const int wslot = wave >> Q.wo_gshift, wgrp = wave & (ngw - 1);
    const int wtile = (int)blockIdx.x * tpw + wslot;
    const bool wg_has_wo = (int)blockIdx.x * tpw < Q.wo_ntiles;
    const int wo_ngroups = (Q.wo_npairs + KL - 1) / KL;
    const int wgg = min(wgrp, wo_ngroups - 1), wgs = min(KL, Q.wo_npairs - wgg * KL);
    const bool wlv = wslot < tpw && wtile < Q.wo_ntiles && wgrp < wo_ngroups && k < wgs;
    const int w_rowt = tid >> 4, w_rr = tid & 15;            // tail: thread = (tile slot, row) of the block's WO rows
    const int w_row = ((int)blockIdx.x * tpw + w_rowt) * TR + w_rr;
    const bool w_act = tid < tpw * TR && (int)blockIdx.x * tpw + w_rowt < Q.wo_ntiles && w_row < D;
    uint4 wc[CPP];
    uint2 wsc;
    float e_resid = 0.f;
    const int kr = lane >> 2, kq = lane & 3, vg = lane >> 4, vc = lane & 15;
    float4 kreg[NV], vreg[NV], kregn[NV];
    // Attention passes are shared once there is more than one (go/model.go:557-587 over 256 positions per pass): the runner of a
    // head keeps the LAST pass (the one that holds this position); pass nch - 1 - p goes to helper p = 1 .. 3 -- the two blocks of
    // the head's own kv group that hold k / v tiles or none (mem >= G: same XCD as the runner) and one of the blocks past the kv
    // groups.  A helper gathers the head's q itself and hands ONE record (max, sum, sum p v[64]) to the runner.
    const int nch = min(pos / TP_PASS + 1, (HELP && Q.helpers) ? 2 * TP_NCH_MAX : TP_NCH_MAX);
    const bool shared = HELP && Q.helpers && nch > 1;       // (HELP: the launch plan of positions from the second pass on -- the plan
                                                            //  of the first pass is compiled without any of this)
    int hhead = -1, hpart = 0;            // this block as a helper
    if (!RUNNER && shared) {
        if (LIVE) { const int j = mem - G; hpart = 1 + j / G; hhead = hpart <= 2 ? cl * G + (j - (hpart - 1) * G) : -1; }
        else { const int j = (int)blockIdx.x - Q.live_grid; hpart = 3; hhead = (j >= 0 && j < Q.n_heads_local) ? j : -1; }
    }
    // part p (0 = the runner) takes the passes [nch - (p + 1) per, nch - p per), per = ceil(nch / 4): contiguous, the last ones the runner's
    const int per = shared ? (nch + 3) >> 2 : nch;
    const int mypart = RUNNER ? 0 : hpart;
    const int chend = nch - mypart * per, ch0 = max(chend - per, 0);     // this block's passes [ch0, chend)
    const bool helper = !RUNNER && shared && hhead >= 0 && chend > 0;
    const int kvh = RUNNER ? cl : helper ? hhead / G : 0;
    const float4 *K4 = reinterpret_cast<const float4 *>(P.kcache + soff + (long long)kvh * P.seq_len * HD);
    const float4 *V4 = reinterpret_cast<const float4 *>(P.vcache + soff + (long long)kvh * P.seq_len * HD);
    // ---- the blocks that only wait for the heads warm the next launch's first round (see PfTiles): wavefronts 1 .. 15
    //      (wavefront 0 polls), one line per thread, split over the non-runner blocks of this block's XCD ----
    unsigned pf0 = 0, pf1 = 0, pf2v = 0;
    auto warm = [&]() {
        if (!RUNNER && Q.pf.nmat > 0 && wave > 0) {
            const int x = (int)(blockIdx.x & 7), b8 = (int)(blockIdx.x >> 3), M = P.members;
            const int lg = max(0, (P.n_kv_heads - x + 7) >> 3);                 // cluster groups of this XCD lane that hold projection tiles
            const int nb8 = ((int)gridDim.x - x + 7) >> 3;
            const int j = LIVE ? (b8 / M) * (M - G) + (mem - G) : lg * (M - G) + (b8 - lg * M);     // rank among the lane's non-runners
            const int np = lg * (M - G) + max(nb8 - lg * M, 0);
            const unsigned lpt = pf_lines(Q.pf), per_block = (unsigned)Q.pf.nmat * lpt;
            const unsigned total = (unsigned)((Q.pf.blocks - x + 7) >> 3) * per_block, stride = (unsigned)np * (TP_THREADS - 64);
            const unsigned p0 = (unsigned)j * (TP_THREADS - 64) + (unsigned)(tid - 64), p1 = p0 + stride;
            if (p0 < total) {
                const unsigned tb = p0 / per_block, rem = p0 - tb * per_block, mat = rem / lpt;
                const int tile = (int)tb * 8 + x;
                if (tile < Q.pf.ntiles) pf0 = pf_touch(Q.pf, (int)mat, tile, rem - mat * lpt);
            }
            if (p1 < total) {
                const unsigned tb = p1 / per_block, rem = p1 - tb * per_block, mat = rem / lpt;
                const int tile = (int)tb * 8 + x;
                if (tile < Q.pf.ntiles) pf1 = pf_touch(Q.pf, (int)mat, tile, rem - mat * lpt);
            }
            if (Q.pf2.nmat > 0) {
                const unsigned lpt2 = pf_lines(Q.pf2);
                const unsigned total2 = (unsigned)((Q.pf2.ntiles - x + 7) >> 3) * lpt2;
                if (p0 < total2) {
                    const unsigned tb = p0 / lpt2;
                    pf2v = pf_touch(Q.pf2, 0, (int)tb * 8 + x, p0 - tb * lpt2);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto second_half_loads = [&]() {
        load_pair<WT>(Q.wo_q, Q.wo_s, (long long)min(wtile, Q.wo_ntiles - 1) * Q.wo_npairs, wgg, wgs, r, min(k, wgs - 1), wc, wsc);
        e_resid = Q.x[min(w_row, D - 1)];
        // the cache rows of this kv head, positions 0 .. min(pos, 127) (rows beyond pos repeat row pos: the same cache lines;
        // the row AT pos is produced by this launch and replaced from LDS below).  Eight cache wavefronts x 16 consecutive
        // positions of a pass: K as 16 of the row's 64 dims per lane (a score = 16 in-lane FMAs + two quad adds), V as float4
        // columns of four rows per lane -- nl_block.h's attention phase.
        if (RUNNER || helper) {       // (the block's first pass: a runner that shares starts on the last one)
            const int t0 = ch0 * TP_PASS, lim = min(min(TP_PASS, P.seq_len - t0), pos + 1 - t0);
            const unsigned krow = (unsigned)(t0 + min(wave * 16 + kr, lim - 1)) * R4 + (unsigned)kq;
#pragma unroll
            for (int kk = 0; kk < NV; kk++) {
                kreg[kk] = ld_off<float4>(K4, (krow + 4 * kk) * 16u);       // float4 kq + 4 kk of the row: a quad reads 64 contiguous bytes per instruction
                vreg[kk] = ld_off<float4>(V4, (unsigned)((t0 + min(wave * 16 + vg + 4 * kk, lim - 1)) * R4 + vc) * 16u);
            }
        }
        if (Q.bias_out) e_bo = Q.bias_out[min(w_row, D - 1)];
    };
    if (!LIVE) second_half_loads();
    if (!LIVE && Q.pf_early) warm();
    __builtin_amdgcn_sched_barrier(0);      // (every request above leaves before anything below consumes one)
    TP_STAMP(sslot, 1);

    if (LIVE) {
        // ---- RMSNorm scaling (go/quant.go:597-607) into wave-private LDS, dot products of this wavefront's column groups.  (An
        //      early barrier for the sum of squares was measured and removed: in a fresh launch x takes as long to arrive as
        //      the weights -- ~2 us for the first fetch of every compute unit at once -- so it only serialised.) ----
        float *xw = xs + wave * XS_WAVE;
        double ss = 0.0;
        float acc = 0.f;
#pragma unroll
        for (int f = 0; f < NF; f++) {
            float4 xa = xin[f] ? xv[f] : make_float4(0.f, 0.f, 0.f, 0.f);
            if (slot == 0 && cs + f * wpt < ngroups) {   // the wavefronts of tile slot 0 see every column exactly once
                ss = fma((double)xa.x, (double)xa.x, ss); ss = fma((double)xa.y, (double)xa.y, ss);
                ss = fma((double)xa.z, (double)xa.z, ss); ss = fma((double)xa.w, (double)xa.w, ss);
            }
            xa.x *= gv[f].x; xa.y *= gv[f].y; xa.z *= gv[f].z; xa.w *= gv[f].w;
            if (!MF) {
                *reinterpret_cast<float4 *>(xw + (lane >> 4) * XS_PAIR + (lane & 15) * 4) = xa;
                __builtin_amdgcn_wave_barrier();
                const float a1 = PairDot<WT>::run(cw[f], sw[f], xw + k * XS_PAIR, acc);
                acc = lv[f] ? a1 : acc;
                __builtin_amdgcn_wave_barrier();
            } else if (f == slot) {
                // Matrix pipe: column group g's digit image is built ONCE per workgroup, by wavefront g (which holds the group as its
                // slot-th one: g = cs + slot * wpt), eight blocks of eight adjacent lanes; every wavefront then multiplies from the
                // shared images
                unsigned char *img = reinterpret_cast<unsigned char *>(xw);
                mf_digits4(xa, img + (lane >> 3) * 128, reinterpret_cast<float2 *>(img + 1024) + (lane >> 3), lane & 7);
            }
        }
        if (MF) {
            __syncthreads();
#pragma unroll
            for (int f = 0; f < NF; f++) {
                const unsigned char *img = reinterpret_cast<const unsigned char *>(xs + min(cs + f * wpt, ngroups - 1) * XS_WAVE);
                const float2 *bs = reinterpret_cast<const float2 *>(img + 1024);
                const int b0 = lane >> 4;
                const float a1 = acc + mf_unit2_q4(cw[f], sw[f].x, img + b0 * 128, bs + b0, lane & 3);
                acc = lv[f] ? a1 : acc;
            }
        }
        second_half_loads();
        __builtin_amdgcn_sched_barrier(0);
        if (!MF) {
            acc = quad_sum(acc);
            if (k == 0) red[wave * TR + r] = acc;
        } else {
            acc = mf_rows4_sum(acc);               // lanes 0 .. 15: row = lane
            if (lane < TR) red[wave * TR + lane] = acc;
        }
        ss = wave_sum_f64(ss);
        if (slot == 0 && lane == 0) dred[cs] = ss;
        TP_STAMP(sslot, 2);
        __syncthreads();
        TP_STAMP(sslot, 3);
        float inv = 0.f;
        if (e_act) {
            double tot = 0.0;
            for (int w = 0; w < wpt; w++) tot += dred[w];
            inv = (float)(1.0 / sqrt(tot / (double)D + (double)P.eps));
        }

        // ---- scale, bias, RoPE (go/model.go:449-477); publish this workgroup's tiles to the cluster, tile order ----
        if (wave == 0) {
            float outv = 0.f;
            if (e_act) {
                const float *rt = red + e_slot * wpt * TR;
                float dotv = 0.f, dotp = 0.f;
                for (int w = 0; w < wpt; w++) { dotv += rt[w * TR + e_rr]; dotp += rt[w * TR + (e_rr ^ 8)]; }   // fixed order
                const float v = dotv * inv + e_b, partner = dotp * inv + e_bp;
                outv = v;
                if (e_sect < 2) {
                    const float x0 = (e_rr < 8) ? v : partner, x1 = (e_rr < 8) ? partner : v;
                    if (!P.rope_conj) outv = (e_rr < 8) ? (x0 * e_cos - x1 * e_sin) : (x0 * e_sin + x1 * e_cos);
                    else outv = (e_rr < 8) ? (x0 * e_cos + x1 * e_sin) : (-x0 * e_sin + x1 * e_cos);
                }
            }
            // the block's tiles mem * tpm .. are consecutive tiles of the group (q heads, then k, then v: four tiles each)
            gran16_publish(Q.xq + ((size_t)cl * (G + 2) * 4 + (size_t)mem * P.tpm) * GPT, P.tpm, tag, outv, lane);
        }
        TP_STAMP(sslot, 4);
        if (!RUNNER && Q.pf_early) warm();
    }
    if (!RUNNER && !wg_has_wo && !helper) return;

    if (RUNNER || helper) {
        // ---- the attention of query head cl * G + mem (go/model.go:557-587): gather the head's q and the group's k | v (a helper:
        //      q of the head it helps; k | v of this position belong to the last pass) ----
        const int h = RUNNER ? cl * G + mem : hhead;
        const int qcl = RUNNER ? cl : hhead / G, qmem = RUNNER ? mem : hhead - (hhead / G) * G;
        const int ntl = RUNNER ? 12 : 4;
        if (tid < 128) {     // 12 tiles x 6 granules on 72 threads (two wavefronts)
            const bool dead = __hip_atomic_load(P.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
            const int gi = min(tid, ntl * GPT - 1), t12 = gi / GPT, j = gi - t12 * GPT;
            const int sect = t12 >> 2, tj = t12 & 3, hq = sect == 0 ? qmem : G + sect - 1;
            const u32x4 g = gran16_wait(Q.xq + ((size_t)qcl * (G + 2) * 4 + hq * 4 + tj) * GPT + j, 0, 1, tag, P.status, P.host_status, P.spin_limit, 8u, dead);
            if (tid < ntl * GPT) {
                const unsigned vals[3] = {g.y, g.z, g.w};
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const int rr = 3 * j + c;       // row of the tile: rows 0-7 hold element tj * 8 + rr, rows 8-15 element 32 + tj * 8 + rr - 8
                    if (rr < 16) qs[sect * 64 + tj * 8 + (rr & 7) + (rr >> 3) * (HD / 2)] = __uint_as_float(vals[c]);   // qs | kcur | vcur are contiguous
                }
            }
        }
        __syncthreads();
        TP_STAMP(sslot, 5);
        if (RUNNER && P.qk_norm) {   // RMSNormBare per head on q and k after RoPE, go/model.go:542-549 (the host keeps helpers off with it)
            if (wave < 2) {
                float *vec = wave == 0 ? qs : kcur;
                const float val = vec[lane];
                const double s2 = wave_sum_f64((double)val * (double)val);
                const float inv = (float)(1.0 / sqrt(s2 / (double)HD + (double)P.eps));
                vec[lane] = val * inv;
            }
            __syncthreads();
        }
        if (RUNNER && mem == 0 && tid < 128)   // KV store go/model.go:552-554, once per kv head
            (tid < 64 ? P.kcache : P.vcache)[soff + ((long long)cl * P.seq_len + pos) * HD + (tid & 63)] = kcur[tid];

        // ---- positions 0..pos, 128 per pass; every cache wavefront reduces its 16 positions to ONE (max, sum, sum p*v)
        //      partial in registers, eight partials per pass meet in LDS behind one barrier (nl_block.h) ----
        for (int ch = ch0; ch < chend; ch++) {
            const int t0 = ch * TP_PASS, n = min(TP_PASS, pos + 1 - t0);
            {
                if (ch > ch0) {
#pragma unroll
                    for (int kk = 0; kk < NV; kk++) {
                        // (the launch with helpers fetches a second pass's K rows at its start: the prefetch a pass ahead costs sixteen
                        //  registers, and this launch at its 128 spilled 22 of them -- with wrong records from the blocks past the kv groups)
                        if (!HELP) kreg[kk] = kregn[kk];
                        else kreg[kk] = K4[(long long)(t0 + min(wave * 16 + kr, n - 1)) * R4 + kq + 4 * kk];
                        vreg[kk] = V4[(long long)(t0 + min(wave * 16 + vg + 4 * kk, n - 1)) * R4 + vc];
                    }
                }
                if (!HELP && ch + 1 < chend) {
                    const int n1 = min(TP_PASS, pos + 1 - t0 - TP_PASS);
#pragma unroll
                    for (int kk = 0; kk < NV; kk++) kregn[kk] = K4[(long long)(t0 + TP_PASS + min(wave * 16 + kr, n1 - 1)) * R4 + kq + 4 * kk];
                }
                const int krow = wave * 16 + kr;
                const bool kcurrow = t0 + krow == pos;   // the row this launch produced: not in memory yet for this workgroup
                float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll
                for (int kk = 0; kk < NV; kk++) {
                    const float4 q4 = *reinterpret_cast<const float4 *>(qs + (kq + 4 * kk) * 4);
                    const float4 kc4 = *reinterpret_cast<const float4 *>(kcur + (kq + 4 * kk) * 4);
                    const float4 k4 = kcurrow ? kc4 : kreg[kk];
                    d0 = fmaf(q4.x, k4.x, d0); d1 = fmaf(q4.y, k4.y, d1); d2 = fmaf(q4.z, k4.z, d2); d3 = fmaf(q4.w, k4.w, d3);
                }
                const float sv = krow < n ? quad_sum((d0 + d1) + (d2 + d3)) * P.scale : -INFINITY;
                const float mw = wave_max_f32(sv);
                const float p = krow < n ? exp_f64_as_f32(sv - (mw == -INFINITY ? 0.f : mw)) : 0.f;     // go/quant.go:619
                const float lw = wave_sum_f32(kq == 0 ? p : 0.f);
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
            if (wave == 0) {
                // the sixteen partials of the pass merged like position splits (fixed order); the weights exp(m_w - M) are this
                // engine's own construct and use the f32 exponential like every split merge here (nl_block.h: the float64 form
                // on this one wavefront costs 350 cycles of the launch's critical path)
                const float mw = lane < 16 ? wpart[min(lane, 15) * 68] : -INFINITY;
                const float Mx = wave_max_f32(mw);
                const float wgt = (lane < 16 && mw != -INFINITY) ? __expf(mw - Mx) : 0.f;
                const float L = wave_sum_f32(lane < 16 ? wgt * wpart[min(lane, 15) * 68 + 1] : 0.f);
                float ov = 0.f;
#pragma unroll
                for (int w = 0; w < 16; w++) ov = fmaf(__shfl(wgt, w), wpart[w * 68 + 4 + lane], ov);
                chunk[(ch - ch0) * 66 + 2 + lane] = ov;
                if (lane == 0) { chunk[(ch - ch0) * 66] = Mx; chunk[(ch - ch0) * 66 + 1] = L; }
                if (!RUNNER) {      // a helper: the pass's record straight to the head's runner (slot = part, pass of the part)
                    __builtin_amdgcn_wave_barrier();
                    const float *rec = chunk + (ch - ch0) * 66;
                    if (lane < 22) gran16_store(Q.xp + (unsigned)(((hhead * 3 + (hpart - 1)) * 2 + (ch - ch0)) * 22 + lane), tag, rec[3 * lane], rec[3 * lane + 1], rec[3 * lane + 2]);
                }
            }
            if (ch + 1 < chend) __syncthreads();   // wpart is rewritten by the next pass
        }
        TP_STAMP(sslot, 6);
        int nrec = chend - ch0;      // records in LDS: this block's own passes ...
        if (RUNNER && shared) {      // ... and the helpers' (the nch - per passes below the runner's, in order of part and pass), 22 granules each
            const int nhr = nch - per;
            __syncthreads();
            if (tid < nhr * 22) {
                const bool dead = __hip_atomic_load(P.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                const int r = tid / 22, j = tid - r * 22, pp = r / per, ci = r - pp * per;
                const u32x4 g = gran16_wait(Q.xp + (unsigned)(((h * 3 + pp) * 2 + ci) * 22 + j), 0, 1, tag, P.status, P.host_status, P.spin_limit, 16u, dead);
                float *rec = chunk + (chend - ch0 + r) * 66 + 3 * j;
                rec[0] = __uint_as_float(g.y); rec[1] = __uint_as_float(g.z); rec[2] = __uint_as_float(g.w);
            }
            nrec += nhr;
        }
        if (HELP) __syncthreads();   // (without helpers wavefront 0 reads back its own records)
        if (!RUNNER) {               // (a helper has sent its records)
        } else if (wave == 0) {
            // merge the records as the WO prologue of the general plan does (load_x4<PRO_ATTN>); publish the head's output
            float outv;
            if (nrec == 1) outv = chunk[2 + lane] * (1.0f / chunk[1]);
            else {
                float Mx = chunk[0];
                for (int c = 1; c < nrec; c++) Mx = fmaxf(Mx, chunk[c * 66]);
                float v = 0.f, L = 0.f;
                for (int c = 0; c < nrec; c++) {
                    const float w = __expf(chunk[c * 66] - Mx);
                    L += w * chunk[c * 66 + 1];
                    v += w * chunk[c * 66 + 2 + lane];
                }
                outv = v * (1.0f / L);
            }
            gran16_publish(Q.xo + (size_t)h * 4 * GPT, 4, tag, outv, lane);
        }
        TP_STAMP(sslot, 7);
        if (!wg_has_wo) return;
    }

    if (!RUNNER && !Q.pf_early) warm();

    // ---- WO (go/model.go:590-594): gather every local head's output, this block's rows of the column slice ----
    // (<= 2 granules per thread: 64 local heads -- the whole 7.9B layer on one GPU -- are 1536 granules)
    tp_gather16<2, false, true>(Q.xo, Q.xo, Q.n_heads_local * 4, tag, ao, ao, Q.n_heads_local * HD, P.status, P.host_status, P.spin_limit, 32u);
    __syncthreads();
    TP_STAMP(sslot, 8);
    if (wslot < tpw) {      // (on the vector pipe in every mode: the heads' outputs serve one tile, their digit image would cost what it saves)
        const float a1 = PairDot<WT>::run(wc, wsc, ao + min(wgg * KL + k, Q.wo_npairs - 1) * XS_PAIR, 0.f);
        float a = wlv ? a1 : 0.f;
        a = quad_sum(a);
        if (k == 0) red2[wave * TR + r] = a;
    }
    __syncthreads();
    TP_STAMP(sslot, 9);
    if (w_act) {
        float v = 0.f;
        for (int g = 0; g < ngw; g++) v += red2[(w_rowt * ngw + g) * TR + w_rr];   // fixed order
        v += e_bo;
        tp_allreduce_row(Q.seam, e_tag, w_row, v, e_resid, Q.x);
    }
    TP_STAMP(sslot, 10);
    pf_done(pf0 ^ pf2v, pf1);
}

template <int WT, int NF, bool MF = false, bool HELP = false>
__global__ void __launch_bounds__(TP_THREADS) tp_attn_kernel(TpAttnParams Q) {
    const GroupParams &P = Q.G;
    NL_KARGS8(P.qkv_q, P.qkv_s, P.x, P.normw, P.rope_cos, P.rope_sin, P.kcache, P.vcache);
    NL_KARGS8(P.ctl, P.bias_q, Q.wo_q, Q.wo_s, Q.xq, P.tick, P.status, P.host_status);
    NL_KARGS8(P.D, P.npairs, P.n_q_heads, P.n_kv_heads, P.seq_len, P.single_stream, P.tpm, P.members);
    NL_KARGS8(P.eps, P.scale, P.kv_stream_stride, Q.wo_npairs, P.layer_tag, P.rope_conj, P.qk_norm, P.bias_k);
    NL_KARGS8(P.spin_limit, P.gqa, Q.wo_ntiles, Q.wo_gshift, Q.xo, Q.x, Q.bias_out, Q.n_heads_local);
    NL_KARGS4(Q.seam.slots, Q.seam.epoch, Q.seam.status, Q.seam.n);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int M = P.members;
    const int b8 = (int)(blockIdx.x >> 3);
    const int cl = (int)udiv_by(blockIdx.x, 8u * (unsigned)M, P.m8_inv) * 8 + (blockIdx.x & 7), mem = b8 - (int)udiv_by((unsigned)b8, (unsigned)M, P.m_inv) * M;
    const bool live = cl < P.n_kv_heads;          // holds projection tiles; the other blocks only multiply WO rows (or exit)
    TP_CENSUS(0, P.layer_tag, 0);
    if (!live) {
        if ((int)blockIdx.x * Q.wo_tpw >= Q.wo_ntiles) return;
        tp_attn_body<WT, NF, false, false, MF, HELP>(Q, smem, 0, 0);
    } else if (mem < (int)P.gqa) tp_attn_body<WT, NF, true, true, MF, HELP>(Q, smem, cl, mem);
    else tp_attn_body<WT, NF, true, false, MF, HELP>(Q, smem, cl, mem);
    TP_CENSUS(0, P.layer_tag, 1);
}

// --------------------------------------------------------------------------------------------- feed-forward half ---

struct TpFfnParams {
    const uint8_t *gate_q, *up_q;     // this rank's gate / up rows, packed [I / 16 tiles][D]
    const uint32_t *gate_s, *up_s;
    const uint8_t *dn_q;              // this rank's W_down column slice, packed [D / 16 tiles][I columns]
    const uint32_t *dn_s;
    int D, I, npairs, gu_tiles;       // npairs of a gate / up row (D / 64); gu_tiles = ceil(I / 16) per matrix
    int dn_npairs, dn_ntiles;
    int pair;                         // 1: a producer holds gate tile t AND up tile t and publishes h; 0: one tile, publishes g or u
    int n_prod;                       // blocks [0, n_prod) project a gate / up tile
    int n_cons, ct_shift;             // blocks [0, n_cons) own 2^ct_shift W_down tiles each (ct * n_cons >= dn_ntiles)
    const float *normw;
    float eps;
    float *x;                         // residual stream: RMSNorm input, rewritten by the row owners
    u32x4 *hx;                        // granules: pair ? h[gu_tiles][6] : g[gu_tiles][6] | u[gu_tiles][6]
    const unsigned *tick;
    unsigned layer_tag;
    unsigned *status, *host_status;
    int spin_limit;
    TpSeam seam;
};

__host__ __device__ constexpr size_t tp_ffn_lds_bytes(int dn_npairs) {
    return sizeof(float) * (size_t)(16 * XS_WAVE + 16 * TR) + 16 * sizeof(double) + sizeof(float) * (size_t)(dn_npairs * XS_PAIR + dn_npairs * PAIR);
}

// NF: 256-column groups of a gate / up row per producer wavefront; NGC: groups of a W_down row per wavefront (16 wavefronts
// share the block's one tile); NR: granules of h gathered per thread (6 * gu_tiles <= NR * 1024).  PROD: the block projects a
// gate / up tile first -- each role straight-line code, every load unconditional (see tp_attn_body).
template <int WT, int NF, int NGC, int NR, bool PROD>
__device__ __forceinline__ void tp_ffn_body(const TpFfnParams &P, char *smem) {
    constexpr int CPP = WTraits<WT>::CPP, NW = TP_THREADS / 64;
    constexpr bool MF = false;        // (the vector-pipe dot products: a rank's W_down slice is not whole 256-column groups)
    double *dred = reinterpret_cast<double *>(smem);                 // [16]
    float *xs = reinterpret_cast<float *>(dred + 16);                // [16][XS_WAVE]
    float *red = xs + NW * XS_WAVE;                                  // [16][16]
    float *hs = red + NW * TR;                                       // [dn_npairs][XS_PAIR]
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int r = lane >> 2, k = lane & 3, D = P.D;
    const int b = (int)blockIdx.x;
    const bool is_cons = b < P.n_cons;
#ifdef NL_TP_STAMPS
    const int sslot = P.layer_tag != NL_TP_STAMPS ? -1 : b == 0 ? 4 : b == P.n_prod - 1 ? 5 : b == P.n_cons / 2 ? 6 : b == P.n_cons - 1 ? 7 : -1;
#endif
    TP_STAMP(sslot, 0);
    const unsigned tag = ((unsigned)sload_i32(reinterpret_cast<const int *>(P.tick)) << 8) | P.layer_tag;   // (scalar cache: no vector wait before the weight requests)
    const unsigned e_tag = P.seam.n > 0 ? ((unsigned)sload_i32(reinterpret_cast<const int *>(P.seam.epoch)) << 8) | P.seam.seam : 0u;

    // ---- producer part: one gate or up tile (16 wavefronts), or gate tile t + up tile t (8 wavefronts each) ----
    const int wpt = P.pair ? NW / 2 : NW;
    const int wsel = P.pair ? wave >> 3 : 0, cs = P.pair ? wave & 7 : wave;
    const int msel = P.pair ? wsel : (b >= P.gu_tiles ? 1 : 0);
    const int ptile = P.pair ? b : b - msel * P.gu_tiles;
    const int ngroups = (P.npairs + KL - 1) / KL;
    float4 xv[NF], gv[NF];
    uint4 cw[NF][CPP];
    uint2 sw[NF];
    bool lv[NF], xin[NF];
    if (PROD) {
        const uint8_t *const Wq = msel ? P.up_q : P.gate_q;
        const uint32_t *const Ws = msel ? P.up_s : P.gate_s;
#pragma unroll
        for (int f = 0; f < NF; f++) {
            const int g = cs + f * wpt, gg = min(g, ngroups - 1);
            const int gs = min(KL, P.npairs - gg * KL);
            lv[f] = g < ngroups && k < gs;
            const int xcol = gg * (KL * PAIR) + lane * 4;
            xin[f] = xcol < D;
            xv[f] = ld_off<float4>(P.x, (unsigned)(xin[f] ? xcol : 0) * 4u);
            gv[f] = ld_off<float4>(P.normw, (unsigned)(xin[f] ? xcol : 0) * 4u);
        }
#pragma unroll
        for (int f = 0; f < NF; f++) {
            const int g = cs + f * wpt, gg = min(g, ngroups - 1);
            const int gs = min(KL, P.npairs - gg * KL);
            load_pair<WT>(Wq, Ws, (long long)ptile * P.npairs, gg, gs, r, min(k, gs - 1), cw[f], sw[f]);
        }
    }
    // ---- consumer part: W_down tiles b * ct .. (clamped), 16 / ct wavefronts per tile share its 256-column groups.  A block
    //      that projects first requests them after its dot products (see tp_attn_body), the others at entry ----
    const int dgroups = (P.dn_npairs + KL - 1) / KL;
    const int ct = 1 << P.ct_shift, wptc = NW >> P.ct_shift;
    const int dslot = wave >> (4 - P.ct_shift), dcs = wave & (wptc - 1);
    const int dtile = b * ct + dslot;
    uint4 dw[NGC][CPP];
    uint2 dsw[NGC];
    bool dlv[NGC];
    int gsel[NGC], dgg[NGC];
    const int o_slot = tid >> 4, o_rr = tid & 15;
    const int o_row = (b * ct + o_slot) * TR + o_rr;
    const bool o_act = is_cons && tid < ct * TR && b * ct + o_slot < P.dn_ntiles && o_row < D;
    float e_resid = 0.f;
    auto second_half_loads = [&]() {
#pragma unroll
        for (int j = 0; j < NGC; j++) {
            const int g = dcs + j * wptc, gg = min(g, dgroups - 1);
            const int gs = min(KL, P.dn_npairs - gg * KL);
            dlv[j] = is_cons && dtile < P.dn_ntiles && g < dgroups && k < gs;
            gsel[j] = min(gg * KL + k, P.dn_npairs - 1);
            dgg[j] = gg;
            load_pair<WT>(P.dn_q, P.dn_s, (long long)min(dtile, P.dn_ntiles - 1) * P.dn_npairs, gg, gs, r, min(k, gs - 1), dw[j], dsw[j]);
        }
        e_resid = P.x[min(o_row, D - 1)];
    };
    if (!PROD) second_half_loads();
    if (PROD) {
        // RMSNorm scaling into wave-private LDS, dot products of this wavefront's column groups (one barrier: see tp_attn_body)
        float *xw = xs + wave * XS_WAVE;
        double ss = 0.0;
        float acc = 0.f;
#pragma unroll
        for (int f = 0; f < NF; f++) {
            float4 xa = xin[f] ? xv[f] : make_float4(0.f, 0.f, 0.f, 0.f);
            if (wsel == 0 && cs + f * wpt < ngroups) {   // the wavefronts of the first tile see every column exactly once
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
        second_half_loads();
        __builtin_amdgcn_sched_barrier(0);
        acc = quad_sum(acc);
        if (k == 0) red[wave * TR + r] = acc;
        ss = wave_sum_f64(ss);
        if (wsel == 0 && lane == 0) dred[cs] = ss;
        TP_STAMP(sslot, 2);
        __syncthreads();
        TP_STAMP(sslot, 3);
        float inv = 0.f;
        if (tid < TR) {
            double tot = 0.0;
            for (int w = 0; w < wpt; w++) tot += dred[w];
            inv = (float)(1.0 / sqrt(tot / (double)D + (double)P.eps));
        }
        if (wave == 0) {
            float outv = 0.f;
            if (lane < TR) {
                float a = 0.f, bsum = 0.f;
                for (int w = 0; w < wpt; w++) { a += red[w * TR + lane]; if (P.pair) bsum += red[(wpt + w) * TR + lane]; }   // fixed order
                outv = a * inv;
                if (P.pair) {   // SiLU go/quant.go:629-631, * up go/model.go:604-606
                    const float g = outv, u = bsum * inv;
                    outv = (g / (1.0f + exp_f64_as_f32(-g))) * u;
                }
            }
            gran16_publish(P.hx + ((size_t)(P.pair ? 0 : msel * P.gu_tiles) + ptile) * GPT, 1, tag, outv, lane);
        }
        TP_STAMP(sslot, 4);
    }
    if (!is_cons) return;

    // ---- gather h (pair) or g | u (SiLU(g) * u applied here) into LDS as padded pairs ----
    for (int i = P.I + tid; i < P.dn_npairs * PAIR; i += TP_THREADS) hs[(i >> 6) * XS_PAIR + (i & 63)] = 0.f;   // ragged last pair
    // (a block that has just published goes straight to the sweep: the weight streams it could disturb are over by then;
    //  a block without a producer part is early and spins on a few probe granules first)
    float *us = hs + P.dn_npairs * XS_PAIR;     // [I] raw up values (split mode)
    if (P.pair) tp_gather16<NR, false, !PROD>(P.hx, P.hx, P.gu_tiles, tag, hs, us, P.I, P.status, P.host_status, P.spin_limit, 64u);
    else tp_gather16<NR, true, !PROD>(P.hx, P.hx + (size_t)P.gu_tiles * GPT, P.gu_tiles, tag, hs, us, P.I, P.status, P.host_status, P.spin_limit, 64u);
    __syncthreads();
    TP_STAMP(sslot, 5);
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < NGC; j++) {
        const float a1 = PairDot<WT>::run(dw[j], dsw[j], hs + gsel[j] * XS_PAIR, acc);
        acc = dlv[j] ? a1 : acc;
    }
    acc = quad_sum(acc);
    if (k == 0) red[wave * TR + r] = acc;
    __syncthreads();
    TP_STAMP(sslot, 6);
    if (o_act) {
        float v = 0.f;
        const int nwg = min(wptc, dgroups);
        for (int w = 0; w < nwg; w++) v += red[(o_slot * wptc + w) * TR + o_rr];   // fixed order
        tp_allreduce_row(P.seam, e_tag, o_row, v, e_resid, P.x);
    }
    TP_STAMP(sslot, 7);
}

template <int WT, int NF, int NGC, int NR>
__global__ void __launch_bounds__(TP_THREADS) tp_ffn_kernel(TpFfnParams P) {
    NL_KARGS8(P.gate_q, P.up_q, P.gate_s, P.up_s, P.dn_q, P.dn_s, P.x, P.normw);
    NL_KARGS8(P.D, P.I, P.npairs, P.gu_tiles, P.dn_npairs, P.dn_ntiles, P.pair, P.n_prod);
    NL_KARGS8(P.eps, P.hx, P.tick, P.layer_tag, P.status, P.host_status, P.spin_limit, P.seam.n);
    NL_KARGS4(P.seam.slots, P.seam.epoch, P.seam.status, P.seam.seam);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    TP_CENSUS(1, P.layer_tag, 0);
    if ((int)blockIdx.x < P.n_prod) tp_ffn_body<WT, NF, NGC, NR, true>(P, smem);
    else if ((int)blockIdx.x < P.n_cons) tp_ffn_body<WT, NF, NGC, NR, false>(P, smem);
    TP_CENSUS(1, P.layer_tag, 1);
}

// ------------------------------------------------------------- feed-forward half of a WHOLE wide layer on one GPU ---
//
// go/model.go:597-612 for the 7.9B tier without tensor parallelism: gate || up (51 MB) and down (25 MB) were two GEMV launches,
// 12.7 + 6.8 us, each with its ~3 us of launch-fixed cost.  Here one launch of one 1024-thread workgroup per W_down tile
// (256 for D = 4096 -- every compute unit) keeps HBM streaming from its first request to the last W_down byte:
//   rounds r = 0 .. R-1: the workgroup projects gate tile t and up tile t, t = block + r * grid (8 wavefronts each, a
//     wavefront holds NF 256-column groups of its tile's rows; the next round's weights are requested before this round's dot
//     products), sums them in LDS, applies RMSNorm's 1 / rms, SiLU(gate) * up (go/quant.go:629-631) and publishes the 16
//     values of h as six tagged 16-byte granules;
//   its W_down tile (16 rows x I columns, NGC 256-column groups per wavefront: 99 KB per workgroup, in registers) and its
//     residual rows are requested behind the LAST round's weights and arrive while that round and the gather run;
//   every workgroup gathers all of h (I / 16 tiles) into LDS, multiplies its rows and stores x = resid + row.
// No workgroup waits for anything but the h granules, so the grid only has to be resident together (one workgroup per compute
// unit: the host enables the launch only when the device has that many); polls are bounded like every exchange here.
struct WideFfnParams {
    const uint8_t *gate_q, *up_q;
    const uint32_t *gate_s, *up_s;
    const uint8_t *dn_q;
    const uint32_t *dn_s;
    int D, I, npairs, gu_tiles;       // npairs of a gate / up row (D / 64); gu_tiles = ceil(I / 16)
    int dn_npairs, dn_ntiles, rounds; // rounds = ceil(gu_tiles / grid)
    const float *normw;
    float eps;
    float *x;                         // residual stream: RMSNorm input, rewritten row by row by the tile owners
    u32x4 *hx;                        // [gu_tiles][6] granules of h
    const unsigned *tick;
    unsigned layer_tag;
    unsigned *status, *host_status;
    int spin_limit;
    TpSeam seam;                      // n = -1: one GPU (x = resid + row); else a tensor-parallel rank's all-reduce seam
    PfQkv pf;                         // the projection tiles the attention launch of the NEXT layer requests at entry (T.nmat 0 = off)
    int pf_ahead;                     // 1: a round's request is followed by one dword per line of the round after it (into this XCD's L2)
};

__host__ __device__ constexpr size_t wide_ffn_lds_bytes(int nf, int dn_npairs) {
    return 16 * sizeof(double) + sizeof(float) * (size_t)(16 * nf * XS_WAVE + 2 * 16 * TR + dn_npairs * XS_PAIR + 16 * TR) +
           (size_t)dn_npairs * (256 + 16);        // (the digit image of h: matrix-pipe dot products)
}

// R = rounds (compile time: every load below is unconditional and in program order, so hipcc's counted waits are exact -- the
// first version looped over a runtime round count with the prefetches behind branches and waited vmcnt(0) between single loads:
// 29 us for what two GEMV launches do in 19.6).  One weight buffer: a round's registers are re-requested for the next round as
// soon as its dot products have consumed them; the W_down tile goes out behind the last round.  Measured and dropped: two
// rounds in flight (20.9 against 19.7 us per launch, and 2 us per layer slower inside the chain); W_down requested by twelve
// wavefronts only so that the other four -- whose queues stay empty: a wavefront's loads return in order, a granule cannot
// overtake weights in flight -- gather h while it streams (21.6 us: four wavefronts need four dependent sweeps for the 4128
// granules).
template <int WT, int NF, int NGC, int R, bool MF = false>
__global__ void __launch_bounds__(TP_THREADS) wide_ffn_kernel(WideFfnParams P) {
    static_assert(!MF || WT == WT_Q4_0, "matrix-pipe dot products: Q4_0");
    NL_KARGS8(P.gate_q, P.up_q, P.gate_s, P.up_s, P.dn_q, P.dn_s, P.x, P.normw);
    NL_KARGS8(P.D, P.I, P.npairs, P.gu_tiles, P.dn_npairs, P.dn_ntiles, P.rounds, P.eps);
    NL_KARGS8(P.hx, P.tick, P.layer_tag, P.status, P.host_status, P.spin_limit, P.x, P.normw);
    NL_KARGS4(P.seam.slots, P.seam.epoch, P.seam.status, P.seam.n);
    constexpr int CPP = WTraits<WT>::CPP, NW = TP_THREADS / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *dred = reinterpret_cast<double *>(smem);                 // [16]
    float *xs = reinterpret_cast<float *>(dred + 16);                // [16 wavefronts][NF][XS_WAVE]: x * g of the wavefront's groups
    float *red = xs + NW * NF * XS_WAVE;                             // [2][16][16]: a round's per-wavefront row sums (parity of the round)
    float *hs = red + 2 * NW * TR;                                   // [dn_npairs][XS_PAIR]: h
    float *red2 = hs + P.dn_npairs * XS_PAIR;                        // [16][16]
    unsigned char *himg = reinterpret_cast<unsigned char *>(red2 + NW * TR);   // MF: [2 dn_npairs blocks][4 digits][32]: the digit image of h ...
    float2 *hbs = reinterpret_cast<float2 *>(himg + P.dn_npairs * 256);         // ... and {1 / scale, offset} per block
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int r = lane >> 2, k = lane & 3, D = P.D;
    const int b = (int)blockIdx.x, nb = (int)gridDim.x;
    const unsigned tag = ((unsigned)sload_i32(reinterpret_cast<const int *>(P.tick)) << 8) | P.layer_tag;
    const unsigned e_tag = P.seam.n > 0 ? ((unsigned)sload_i32(reinterpret_cast<const int *>(P.seam.epoch)) << 8) | P.seam.seam : 0u;
    const int wsel = wave >> 3, cs = wave & 7;                       // 0: the gate tile's wavefronts, 1: the up tile's
    const uint8_t *const Wq = wsel ? P.up_q : P.gate_q;
    const uint32_t *const Ws = wsel ? P.up_s : P.gate_s;
    const int ngroups = (P.npairs + KL - 1) / KL;

    // ---- x and the norm weights of this wavefront's groups, the residual rows of the block's tile, round 0's weights ----
    float4 xv[NF], gv[NF];
    uint4 cw[NF][CPP];
    uint2 sw[NF];
    bool lv[NF], xin[NF];
    int ggs[NF], gsz[NF];
#pragma unroll
    for (int f = 0; f < NF; f++) {
        const int g = cs + f * 8;
        ggs[f] = min(g, ngroups - 1);
        gsz[f] = min(KL, P.npairs - ggs[f] * KL);
        lv[f] = g < ngroups && k < gsz[f];
        const int xcol = ggs[f] * (KL * PAIR) + lane * 4;
        xin[f] = g < ngroups && xcol < D;
        xv[f] = ld_off<float4>(P.x, (unsigned)(xcol < D ? xcol : 0) * 4u);
        gv[f] = ld_off<float4>(P.normw, (unsigned)(xcol < D ? xcol : 0) * 4u);
    }
    const int o_row = b * TR + (tid & 15);
    const bool o_act = tid < TR && b < P.dn_ntiles && o_row < D;
    const float e_resid = P.x[min(o_row, D - 1)];
    // (One weight buffer.  A second one -- a round's dot products running while the next round's weights are in flight -- was
    //  measured twice and lost both times: 20.9 against 19.7 us on the vector pipe (round 4), 19.1 against 17.8 us with the
    //  matrix-pipe dot products and 98 registers (round 5): two rounds in flight are 38 MB against 32 MB of L2.)
    auto load_round = [&](int rnd) {
        const int t = min(b + rnd * nb, P.gu_tiles - 1);                       // (a block without a tile in the last round repeats a request)
#pragma unroll
        for (int f = 0; f < NF; f++) load_pair<WT>(Wq, Ws, (long long)t * P.npairs, ggs[f], gsz[f], r, min(k, gsz[f] - 1), cw[f], sw[f]);
    };
    load_round(0);
    const int dgroups = (P.dn_npairs + KL - 1) / KL;
    const int dtile = min(b, P.dn_ntiles - 1);
    uint4 dw[NGC][CPP];
    uint2 dsw[NGC];
    bool dlv[NGC];
    int gsel[NGC], dgg[NGC];
#pragma unroll
    for (int j = 0; j < NGC; j++) { dlv[j] = false; gsel[j] = 0; dgg[j] = 0; }
    auto down_loads = [&]() {
#pragma unroll
        for (int j = 0; j < NGC; j++) {
            const int g = wave + j * NW, gg = min(g, dgroups - 1);
            const int gs = min(KL, P.dn_npairs - gg * KL);
            dlv[j] = b < P.dn_ntiles && g < dgroups && k < gs;
            gsel[j] = min(gg * KL + k, P.dn_npairs - 1);
            dgg[j] = gg;
            load_pair<WT>(P.dn_q, P.dn_s, (long long)dtile * P.dn_npairs, gg, gs, r, min(k, gs - 1), dw[j], dsw[j]);
        }
    };
    __builtin_amdgcn_sched_barrier(0);

    // ---- x * g into wave-private LDS (once), the sum of squares by the gate wavefronts ----
    float *xw = xs + wave * NF * XS_WAVE;
    double ss = 0.0;
#pragma unroll
    for (int f = 0; f < NF; f++) {
        float4 xa = xin[f] ? xv[f] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (wsel == 0) {
            ss = fma((double)xa.x, (double)xa.x, ss); ss = fma((double)xa.y, (double)xa.y, ss);
            ss = fma((double)xa.z, (double)xa.z, ss); ss = fma((double)xa.w, (double)xa.w, ss);
        }
        xa.x *= gv[f].x; xa.y *= gv[f].y; xa.z *= gv[f].z; xa.w *= gv[f].w;
        if (!MF) *reinterpret_cast<float4 *>(xw + f * XS_WAVE + (lane >> 4) * XS_PAIR + (lane & 15) * 4) = xa;
        else {
            unsigned char *img = reinterpret_cast<unsigned char *>(xw + f * XS_WAVE);
            mf_digits4(xa, img + (lane >> 3) * 128, reinterpret_cast<float2 *>(img + 1024) + (lane >> 3), lane & 7);
        }
    }
    ss = wave_sum_f64(ss);
    if (wsel == 0 && lane == 0) dred[cs] = ss;
    __builtin_amdgcn_wave_barrier();

    float inv = 0.f;
    unsigned pfa[2] = {0u, 0u}, pfn = 0u;
#pragma unroll
    for (int rnd = 0; rnd < R; rnd++) {
        const int t = b + rnd * nb;
        const bool has = t < P.gu_tiles;                     // (block-uniform; a block without a tile computes on a repeated one and drops it)
        float acc = 0.f;
#pragma unroll
        for (int f = 0; f < NF; f++) {
            if (!MF) {
                const float a1 = PairDot<WT>::run(cw[f], sw[f], xw + f * XS_WAVE + k * XS_PAIR, acc);
                acc = lv[f] ? a1 : acc;
            } else {
                const unsigned char *img = reinterpret_cast<const unsigned char *>(xw + f * XS_WAVE);
                const float2 *bs = reinterpret_cast<const float2 *>(img + 1024);
                const int b0 = lane >> 4;
                const float a1 = acc + mf_unit2_q4(cw[f], sw[f].x, img + b0 * 128, bs + b0, lane & 3);
                acc = lv[f] ? a1 : acc;
                if (rnd + 1 < R) {       // this group's registers go back out before the next group's products
                    __builtin_amdgcn_sched_barrier(0);
                    const int t1 = min(b + (rnd + 1) * nb, P.gu_tiles - 1);
                    load_pair<WT>(Wq, Ws, (long long)t1 * P.npairs, ggs[f], gsz[f], r, min(k, gsz[f] - 1), cw[f], sw[f]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!MF && rnd + 1 < R) load_round(rnd + 1);         // the freed registers go straight back out
        if (rnd + 1 == R) down_loads();
        if (rnd + 2 < R && P.pf_ahead) {                     // ... and the round after it starts moving into this XCD's L2
            const int t2 = b + (rnd + 2) * nb;
            const unsigned lpt = (unsigned)P.npairs * (CPP * TR * 16 + TR * 4 * scale_words(WT)) / 128u, ql = (unsigned)P.npairs * (CPP * TR * 16) / 128u;
            if (t2 < P.gu_tiles && (unsigned)tid < 2u * lpt) {
                const unsigned mat = (unsigned)tid >= lpt ? 1u : 0u, l = (unsigned)tid - mat * lpt;
                const uint8_t *pq = (mat ? P.up_q : P.gate_q) + ((size_t)t2 * ql + l) * 128u;
                const uint8_t *ps = reinterpret_cast<const uint8_t *>(mat ? P.up_s : P.gate_s) + ((size_t)t2 * (lpt - ql) + (l - ql)) * 128u;
                pfa[rnd & 1] = *reinterpret_cast<const unsigned *>(l < ql ? pq : ps);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        float *rd = red + (rnd & 1) * NW * TR;
        if (!MF) {
            acc = quad_sum(acc);
            if (k == 0) rd[wave * TR + r] = acc;
        } else {
            acc = mf_rows4_sum(acc);
            if (lane < TR) rd[wave * TR + lane] = acc;
        }
        __syncthreads();
        if (rnd == 0 && tid < TR) {
            double tot = 0.0;
            for (int w = 0; w < 8; w++) tot += dred[w];
            inv = (float)(1.0 / sqrt(tot / (double)D + (double)P.eps));
        }
        if (wave == 0 && has) {
            float outv = 0.f;
            if (lane < TR) {
                float a = 0.f, u = 0.f;
                for (int w = 0; w < 8; w++) { a += rd[w * TR + lane]; u += rd[(8 + w) * TR + lane]; }   // fixed order
                const float g = a * inv;
                outv = (g / (1.0f + exp_f64_as_f32(-g))) * (u * inv);                                     // go/quant.go:629-631, go/model.go:604-606
            }
            gran16_publish(P.hx + (size_t)t * GPT, 1, tag, outv, lane);
        }
    }
    if (b >= P.dn_ntiles) return;

    // ---- gather h into LDS as padded pairs (every granule fetched once, five in flight per thread, every tag checked) ----
    {
        for (int i = P.I + tid; i < P.dn_npairs * PAIR; i += TP_THREADS) hs[(i >> 6) * XS_PAIR + (i & 63)] = 0.f;   // ragged last pair
        constexpr int NRG = 5, GT = TP_THREADS;
        const int ng = P.gu_tiles * GPT;
        const bool dead = __hip_atomic_load(P.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        for (int q0 = 0; q0 < ng; q0 += NRG * GT) {
            u32x4 ga[NRG];
            for (int spins = 0;; spins++) {
                bool ok = true;
                const u32x4 *gp[NRG];
#pragma unroll
                for (int q = 0; q < NRG; q++) gp[q] = P.hx + min(q0 + q * GT + tid, ng - 1);
                gran16_load5(gp[0], gp[1], gp[2], gp[3], gp[4], ga[0], ga[1], ga[2], ga[3], ga[4]);
#pragma unroll
                for (int q = 0; q < NRG; q++) ok = ok && ga[q].x == tag;
                if (__all(ok)) break;
                if (dead || spins >= P.spin_limit) { if (lane == 0) { atomicOr(P.status, 64u); *P.host_status = 64u; } break; }
                __builtin_amdgcn_s_sleep(2);
            }
#pragma unroll
            for (int q = 0; q < NRG; q++) {
                const int gi = q0 + q * GT + tid;
                if (gi < ng) {
                    const int t = gi / GPT, j = gi - t * GPT;
                    const unsigned va[3] = {ga[q].y, ga[q].z, ga[q].w};
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        const int i = t * 16 + 3 * j + c;
                        if (3 * j + c < 16 && i < P.I) hs[(i >> 6) * XS_PAIR + (i & 63)] = __uint_as_float(va[c]);
                    }
                }
            }
        }
    }
    __syncthreads();
    if (MF) {       // h -> digit image: a thread = four consecutive values, a block = eight adjacent threads
        for (int i4 = tid; i4 < P.dn_npairs * (PAIR / 4); i4 += TP_THREADS) {
            const int e = i4 * 4;
            const float4 v = *reinterpret_cast<const float4 *>(hs + (e >> 6) * XS_PAIR + (e & 63));
            mf_digits4(v, himg + (i4 >> 3) * 128, hbs + (i4 >> 3), i4 & 7);
        }
        __syncthreads();
    }
    if (P.pf.T.nmat > 0) {
        // h is here and the W_down tile has landed: nothing else of this launch will ask memory for anything but the row stores.
        // The projection tiles block b of the NEXT layer's attention launch requests at entry (tp_attn_body: tile_of) start moving
        // into this XCD's L2 while the last dot products, the tail and the launch boundary pass.
        const PfQkv &N = P.pf;
        const int M = N.members, b8 = b >> 3;
        const int cl = (int)udiv_by((unsigned)b, 8u * (unsigned)M, N.m8_inv) * 8 + (b & 7), mem = b8 - (int)udiv_by((unsigned)b8, (unsigned)M, N.m_inv) * M;
        const unsigned lpt = pf_lines(N.T);
        if (cl < N.n_kv_heads && (unsigned)tid < (unsigned)N.tpm * lpt) {
            const unsigned slot = (unsigned)tid / lpt, l = (unsigned)tid - slot * lpt;
            const int u = mem * N.tpm + (int)slot, jj = u & 3, hq = u >> 2;
            const int sect = hq < N.gqa ? 0 : hq == N.gqa ? 1 : 2;
            const int tile = (sect == 0 ? cl * N.gqa + hq : sect == 1 ? N.n_q_heads + cl : N.n_q_heads + N.n_kv_heads + cl) * 4 + jj;
            if (hq < N.gqa + 2 && tile < N.T.ntiles) pfn = pf_touch(N.T, 0, tile, l);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < NGC; j++) {
        if (!MF) {
            const float a1 = PairDot<WT>::run(dw[j], dsw[j], hs + gsel[j] * XS_PAIR, acc);
            acc = dlv[j] ? a1 : acc;
        } else {
            const int b0 = dgg[j] * 8 + (lane >> 4);
            const float a1 = acc + mf_unit2_q4(dw[j], dsw[j].x, himg + b0 * 128, hbs + b0, lane & 3);
            acc = dlv[j] ? a1 : acc;
        }
    }
    if (!MF) {
        acc = quad_sum(acc);
        if (k == 0) red2[wave * TR + r] = acc;
    } else {
        acc = mf_rows4_sum(acc);
        if (lane < TR) red2[wave * TR + lane] = acc;
    }
    __syncthreads();
    if (o_act) {
        float v = 0.f;
        for (int w = 0; w < NW; w++) v += red2[w * TR + (tid & 15)];      // fixed order (a wavefront without a group left 0)
        tp_allreduce_row(P.seam, e_tag, o_row, v, e_resid, P.x);
    }
    pf_done(pfa[0] ^ pfa[1], pfn);
}

// Soak test of the property the 16-byte granules rest on (nl_op_gran16_soak; tests/test_gpu_tp_fused.py): one dwordx4
// write-through store is seen whole or not at all by a dwordx4 L1-bypassing load of another compute unit.  Blocks [0, n) write
// generations 1 .. iters of a granule whose three payload words are functions of its tag; blocks [n, 2n) -- on other XCDs: the
// reader of writer b is block n + (b + 1) % n -- read it as fast as they can and count every copy whose words disagree.
__global__ void __launch_bounds__(256) gran16_soak_kernel(u32x4 *slots, int n, unsigned iters, unsigned long long *torn, unsigned long long *seen) {
    const int b = (int)blockIdx.x, tid = (int)threadIdx.x;
    if (b < n) {
        u32x4 *p = slots + (size_t)b * 256 + tid;
        for (unsigned g = 1; g <= iters; g++) {
            const u32x4 v = {g, g * 2654435761u + 1u, g ^ 0x5bd1e995u, ~g};
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
            if ((g & 7u) == 0u) __builtin_amdgcn_s_sleep(1);
        }
        return;
    }
    const u32x4 *p = slots + (size_t)((b - n + 1) % n) * 256 + tid;
    unsigned long long bad = 0, reads = 0;
    for (unsigned spins = 0; spins < 40000000u; spins++) {
        const u32x4 v = gran16_load(p);
        reads++;
        if (v.x != 0u && (v.y != v.x * 2654435761u + 1u || v.z != (v.x ^ 0x5bd1e995u) || v.w != ~v.x)) bad++;
        if (v.x == iters) break;
    }
    atomicAdd(torn, bad);
    atomicAdd(seen, reads);
}

// x[i] += sum[i]: the in-process shard group's counterpart of the owner lanes' store (nl_group_forward)
__global__ void tp_add_kernel(float *x, const float *sum, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = x[i] + sum[i];
}

}  // namespace nl
