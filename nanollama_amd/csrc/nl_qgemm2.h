// nl_qgemm2.h -- the multi-token GEMM for LONG token runs (prompt prefill, Q4_0): weights through LDS, activations in registers.
//
// qgemm_kernel (nl_qgemm.h) gives every wavefront its own 16 weight rows and shares the activation fragments of 64 tokens
// through LDS: per 32-column block a wavefront issues 8 ds_read_b128 for 8 MFMAs, expands its weight fragment itself
// and spends 16 VALU FMAs on the block scale -- LDS reads, VALU and MFMA issue are co-limited (measured: MFMA pipe
// 29 % busy at 2047 tokens, and neither a 3-MFMA scaled-weight form nor two row tiles per wavefront helped).
// Here the roles are swapped.  A workgroup owns RT (4) 16-row tiles of weights and WAVES * NTW 16-token tiles; all its
// threads expand the RT*16 x 128-column chunk of weights ONCE into fp16 MFMA fragments in LDS (one (row, block) item per
// thread, a quarter of the expansion after each of the chunk's four blocks, raw quants requested two chunks ahead), every
// wavefront owns NTW token tiles whose hi / lo activation fragments it loads straight from the fragment store into
// registers one block ahead, and per block it multiplies them against all RT row tiles: RT ds_read_b128 feed
// 2 * RT * NTW MFMAs.  The weight fragment is the MFMA's A operand, so a lane owns four consecutive rows of one token:
// block scales arrive as one ds_read_b128, `acc += z * d` is two v_pk_fma_f32, results leave as float4 stores.
// The written order is pinned (sched_barrier + an empty asm on the accumulators): hipcc otherwise sinks every "* d" FMA
// of a chunk below its 64 MFMAs (128 live result registers, occupancy 2) and the activation prefetch next to its use.
// Same arithmetic per output as qgemm_kernel (exact integer quants in fp16, x = hi + lo, f32 block sums, * d in f32):
// outputs are bit-identical between the two kernels (tests/test_gpu_parity.py, tools/qgemm2_bench.hip).  DESIGN.md 4.2b.
#pragma once
#include "nl_qgemm.h"

namespace nl {

#ifdef QG2_STAMPS
__device__ long long g_qg2_stamps[4096];
#define QG2_STAMP(i) do { if (blockIdx.x == QG2_STAMPS && blockIdx.y == 1 && blockIdx.z == 0 && tid == 64) g_qg2_stamps[(i)] = clock64(); } while (0)
#else
#define QG2_STAMP(i) do { } while (0)
#endif


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
// RT: 16-row tiles per workgroup (8 or 4); NTW: 16-token tiles per wavefront (every weight fragment read from LDS feeds
// 2 * NTW MFMAs); WAVES: wavefronts per workgroup, so a workgroup covers RT * 16 rows x WAVES * NTW * 16 tokens.
// The weight fragment is the MFMA's A operand: D[weight row = (lane>>4)*4 + j][token = lane & 15], so a lane owns four
// consecutive rows of one token -- the block scales arrive as one float4 from LDS and the result leaves as float4 stores.
// EPI as qgemm_kernel's: QG_EPI_SWIGLU (RT = 4: two gate tiles and the same two tiles of up; h leaves as the down
// projection's fragments), QG_EPI_ROPE (the matrix is a layer's packed Q|K|V: bias, RoPE, KV store).
template <int WT, int RT, int NTW, int WAVES, int EPI = QG_EPI_PLAIN>
__global__ void __launch_bounds__(WAVES * 64, (RT * NTW <= 8 ? 4 : 2)) qgemm2_kernel(QGemmParams P) {
    static_assert(EPI != QG_EPI_SWIGLU || RT == 4, "the fused gate/up workgroup is 2 + 2 row tiles");
    constexpr int ROWS = RT * TR, NTHR = WAVES * 64, ITEMS = ROWS * QG_KC, IPT = (ITEMS + NTHR - 1) / NTHR;
    __shared__ __attribute__((aligned(16))) uint4 wfrag[2][QG_KC][RT][64];   // [buffer][block][row tile][lane] fp16 x 8
    __shared__ __attribute__((aligned(16))) float wsc[2][QG_KC][ROWS];       // block scales
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int li = lane & 15, lq = lane >> 4;
    const int nblocks = P.cols / 32, nchunks = (nblocks + QG_KC - 1) / QG_KC;
    constexpr bool FUSED = EPI == QG_EPI_SWIGLU;
    const int tile0 = FUSED ? blockIdx.x * 2 : blockIdx.x * RT, row0 = tile0 * TR;   // FUSED: of gate and of up
    const int ttile0 = (blockIdx.y * WAVES + wave) * NTW;            // this wavefront's first 16-token tile

    // staging items of this thread: (row of the workgroup, block of the chunk), item = tid + k * NTHR
    auto item_live = [&](int k) { return ITEMS % NTHR == 0 || tid + k * NTHR < ITEMS; };
    auto stage_load = [&](int chunk, RowBlock<WT> (&rb)[IPT], uint32_t (&sw)[IPT]) {
#pragma unroll
        for (int k = 0; k < IPT; k++) {
            const int it = min(tid + k * NTHR, ITEMS - 1), s_row = it % ROWS, s_blk = it / ROWS, s_i = s_row & 15;
            const int s_rt = s_row >> 4;
            const int s_tile = min(tile0 + (FUSED ? s_rt & 1 : s_rt), P.ntiles - 1);
            const uint8_t *const Wq = FUSED && s_rt >= 2 ? P.q1 : P.q;
            const uint32_t *const Ws = FUSED && s_rt >= 2 ? P.s1 : P.s;
            const int blk = min(chunk * QG_KC + s_blk, nblocks - 1);
            const int g = blk >> 3, gsz = min(KL, P.npairs - g * KL);
            const unsigned gp = (unsigned)(s_tile * P.npairs + g * KL);
            rb[k] = RowBlock<WT>::load(Wq, gp, gsz, s_i, blk);
            sw[k] = Ws[(size_t)gp * TR + (size_t)s_i * gsz + ((blk >> 1) & 3)];
        }
    };
    // one quarter of the expansion (fragment w of every item; the scale with fragment 0): spread over the chunk's four blocks
    // (X1, the single-product precision mode: the block scale is multiplied into the fp16 weight fragment here, once per
    //  workgroup and chunk, so the K loop is a bare MFMA chain -- no per-block "* d" on the vector pipe.  (n - 8) * d rounded to
    //  fp16 costs 2^-12 relative per weight, the size of the mode's activation rounding.)
    auto stage_store_part = [&](auto x1_tag, int chunk, int buf, int w, const RowBlock<WT> (&rb)[IPT], const uint32_t (&sw)[IPT]) {
        constexpr bool X1 = decltype(x1_tag)::value;
#pragma unroll
        for (int k = 0; k < IPT; k++) {
            if (!item_live(k)) continue;
            const int it = tid + k * NTHR, s_row = it % ROWS, s_blk = it / ROWS, s_i = s_row & 15, s_rt = s_row >> 4;
            const float d = chunk * QG_KC + s_blk < nblocks ? scale_of(sw[k], chunk * QG_KC + s_blk) : 0.f;   // a block past the end of K contributes nothing
            half8_t f = rb[k].frag(w);
            if (X1) {
                const _Float16 dh = (_Float16)d;        // (d is an fp16 value: exact)
#pragma unroll
                for (int e = 0; e < 8; e++) f[e] = f[e] * dh;
            }
            wfrag[buf][s_blk][s_rt][w * 16 + s_i] = __builtin_bit_cast(uint4, f);
            if (!X1 && w == 0) wsc[buf][s_blk][s_row] = d;
        }
    };
    // activation fragments of this wavefront's token tiles: [block][tile][hi/lo][lane] x 16 B.  Addresses are "kernel
    // argument + unsigned 32-bit offset" with the block part wave-uniform (scalar base, saddr loads: no 64-bit VGPR math)
    const char *const xsrc = reinterpret_cast<const char *>(P.xf);
    unsigned xlane[NTW];
#pragma unroll
    for (int t = 0; t < NTW; t++) xlane[t] = ((unsigned)min(ttile0 + t, P.nt16 - 1) * 2u * QG_FRAG + (unsigned)lane) * 16u;
    const unsigned xblock = (unsigned)P.nt16 * 2u * QG_FRAG * 16u;
    // (X1: the single-product precision mode -- the lo halves are neither fetched nor multiplied)
    auto xload1 = [&](auto x1_tag, int blk, uint4 (&h)[NTW], uint4 (&l)[NTW]) {
        constexpr bool X1 = decltype(x1_tag)::value;
        const unsigned uo = (unsigned)min(blk, nblocks - 1) * xblock;
#pragma unroll
        for (int t = 0; t < NTW; t++) {
            h[t] = *reinterpret_cast<const uint4 *>(xsrc + (uo + xlane[t]));
            if (!X1) l[t] = *reinterpret_cast<const uint4 *>(xsrc + (uo + xlane[t]) + QG_FRAG * 16);
        }
    };

    f32x4_t acc[RT][NTW];
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
        for (int t = 0; t < NTW; t++) acc[rt][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    constexpr int RG = NTW == 1 ? (RT % 4 == 0 ? 4 : RT % 3 == 0 ? 3 : RT % 2 == 0 ? 2 : 1) : (RT % 2 == 0 ? 2 : 1);   // row tiles per MFMA group: RG * NTW <= 4 independent accumulator chains
    constexpr int NG = RT / RG;            // groups per block
    static_assert((QG_KC * NG) % 2 == 0, "weight fragment ping-pong");
    int chunk = blockIdx.z;
    RowBlock<WT> rb[IPT], rbn[IPT];        // raw weights of the next chunk (being expanded) and of the one after (in flight)
    uint32_t sw[IPT], swn[IPT];
    uint4 xh[2][NTW], xl[2][NTW];          // ping-pong: block b of a chunk sits in [b & 1] (QG_KC is even)
    uint4 wf[2][RG];                       // ping-pong: group idx of a chunk sits in [idx & 1]
    int buf = 0;
    [[maybe_unused]] int it_ = 0;
    // The K loop exists as TWO instruction streams chosen once by a uniform test: x = hi + lo (two MFMAs per product, float32-
    // grade results) and the fp16x1 precision mode (hi only: half the matrix work, activations rounded to 11 bits).
    auto mainloop = [&](auto x1_tag) {
    constexpr bool X1 = decltype(x1_tag)::value;
    if (chunk < nchunks) {
        stage_load(chunk, rb, sw);
        xload1(x1_tag, chunk * QG_KC, xh[0], xl[0]);
#pragma unroll
        for (int w = 0; w < 4; w++) stage_store_part(x1_tag, chunk, 0, w, rb, sw);
        stage_load(chunk + P.ksplit < nchunks ? chunk + P.ksplit : chunk, rb, sw);
    }
    __syncthreads();
    QG2_STAMP(0);
    while (chunk < nchunks) {
        const int nxt = chunk + P.ksplit, nxt2 = nxt + P.ksplit;
        const bool more = nxt < nchunks;
        // raw weights two chunks ahead (one HBM round trip is longer than a chunk of MFMAs); the activation fragments are
        // requested one block ahead (an L2 round trip)
        stage_load(nxt2 < nchunks ? nxt2 : chunk, rbn, swn);
#pragma unroll
        for (int q = 0; q < RG; q++) wf[0][q] = wfrag[buf][0][q][lane];
        QG2_STAMP(1 + it_ * 8);
#pragma unroll
        for (int b = 0; b < QG_KC; b++) {
            xload1(x1_tag, b + 1 < QG_KC ? chunk * QG_KC + b + 1 : (more ? nxt : chunk) * QG_KC, xh[(b + 1) & 1], xl[(b + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);   // the request stays HERE: a block ahead of its use
#pragma unroll
            for (int g = 0; g < NG; g++) {
                constexpr int dummy = 0; (void)dummy;
                const int idx = b * NG + g, r0 = g * RG;
                // the next group's weight fragments leave LDS while this group is on the matrix cores
                if (idx + 1 < QG_KC * NG) {
                    const int b1 = (idx + 1) / NG, g1 = (idx + 1) % NG;
#pragma unroll
                    for (int q = 0; q < RG; q++) wf[(idx + 1) & 1][q] = wfrag[buf][b1][g1 * RG + q][lane];
                }
                if constexpr (X1) {
                    // scaled weights: the products accumulate across blocks inside the MFMA accumulators
#pragma unroll
                    for (int q = 0; q < RG; q++)
#pragma unroll
                        for (int t = 0; t < NTW; t++)
                            acc[r0 + q][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, wf[idx & 1][q]), __builtin_bit_cast(half8_t, xh[b & 1][t]),
                                                                                  acc[r0 + q][t], 0, 0, 0);
                } else {
                f32x4_t z[RG][NTW];
                // the lo-part MFMAs of the group, then the hi-part MFMAs: each dependent pair is four issues apart, so no
                // MFMA waits for its predecessor's accumulator
#pragma unroll
                for (int q = 0; q < RG; q++)
#pragma unroll
                    for (int t = 0; t < NTW; t++)
                        z[q][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, wf[idx & 1][q]), __builtin_bit_cast(half8_t, xl[b & 1][t]),
                                                                        (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < RG; q++)
#pragma unroll
                    for (int t = 0; t < NTW; t++)
                        z[q][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, wf[idx & 1][q]), __builtin_bit_cast(half8_t, xh[b & 1][t]), z[q][t], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < RG; q++) {   // acc += z * d, two rows per instruction (v_pk_fma_f32: same roundings as fmaf)
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    const float4 d = *reinterpret_cast<const float4 *>(&wsc[buf][b][(r0 + q) * TR + lq * 4]);
                    const f2 d01 = {d.x, d.y}, d23 = {d.z, d.w};
#pragma unroll
                    for (int t = 0; t < NTW; t++) {
                        const f2 lo = __builtin_elementwise_fma((f2){z[q][t][0], z[q][t][1]}, d01, (f2){acc[r0 + q][t][0], acc[r0 + q][t][1]});
                        const f2 hi = __builtin_elementwise_fma((f2){z[q][t][2], z[q][t][3]}, d23, (f2){acc[r0 + q][t][2], acc[r0 + q][t][3]});
                        acc[r0 + q][t] = (f32x4_t){lo[0], lo[1], hi[0], hi[1]};
                    }
                }
                }
                // Pin the updated accumulators here: instruction selection otherwise sinks every "* d" FMA of the chunk
                // below its 64 MFMAs and keeps 128 result registers live (occupancy 2 instead of 4).
#pragma unroll
                for (int q = 0; q < RG; q++)
#pragma unroll
                    for (int t = 0; t < NTW; t++) asm volatile("" : "+v"(acc[r0 + q][t]));
                if (g == NG - 1 && more) stage_store_part(x1_tag, nxt, buf ^ 1, b, rb, sw);   // the other buffer: nobody reads it during this chunk
                __builtin_amdgcn_sched_barrier(0);
            }
            QG2_STAMP(2 + it_ * 8 + b);
        }
        QG2_STAMP(6 + it_ * 8);
        __syncthreads();
        QG2_STAMP(7 + it_ * 8);
        it_++;
#pragma unroll
        for (int k = 0; k < IPT; k++) { rb[k] = rbn[k]; sw[k] = swn[k]; }
        buf ^= 1;
        chunk = nxt;
    }
    };
    if (P.x1) mainloop(std::true_type{});
    else mainloop(std::false_type{});

    QG2_STAMP(1 + it_ * 8);
    // folded RMSNorm, consumer side: inv of this lane's tokens from the producer's per-block sums of squares (block order:
    // deterministic), applied to the accumulators before bias / RoPE / SiLU -- the decode GEMV's "scale the output" form
    [[maybe_unused]] float inv[NTW];
    if constexpr (EPI != QG_EPI_PLAIN) {
#pragma unroll
        for (int t = 0; t < NTW; t++) inv[t] = 1.0f;
        if (P.nrm_in.ssq) {
#pragma unroll
            for (int t = 0; t < NTW; t++) {
                const double *sp = P.nrm_in.ssq + (size_t)min((ttile0 + t) * 16 + li, P.n_tokens - 1) * P.nrm_in.nrb;
                double tot = 0.0;
                for (int r0 = 0; r0 < P.nrm_in.nrb; r0 += 8) {    // eight partial sums per memory round trip (clamped, masked)
                    double v[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) v[k] = sp[min(r0 + k, P.nrm_in.nrb - 1)];
#pragma unroll
                    for (int k = 0; k < 8; k++) tot += r0 + k < P.nrm_in.nrb ? v[k] : 0.0;
                }
                inv[t] = (float)(1.0 / sqrt(tot / (double)P.nrm_in.dim + (double)P.nrm_in.eps));
            }
            if (P.nrm_in.scale) {     // the producer's power-of-two pre-scale (norm_prescale, nl_qgemm.h): undone exactly
                float sc[NTW];
#pragma unroll
                for (int t = 0; t < NTW; t++) sc[t] = P.nrm_in.scale[min((ttile0 + t) * 16 + li, P.n_tokens - 1)];
#pragma unroll
                for (int t = 0; t < NTW; t++) {
                    if (blockIdx.x == 0 && lq == 0) P.nrm_in.scale_next[min((ttile0 + t) * 16 + li, P.n_tokens - 1)] = norm_prescale(inv[t]);
                    inv[t] *= 1.0f / sc[t];
                }
            }
#pragma unroll
            for (int rt = 0; rt < RT; rt++)
#pragma unroll
                for (int t = 0; t < NTW; t++) acc[rt][t] = acc[rt][t] * inv[t];
        }
    }
    if constexpr (EPI == QG_EPI_SWIGLU) {
        // h = SiLU(gate) * up (go/quant.go:629-631, go/model.go:604-606).  A lane holds rows 4*lq..+3 of both 16-row tiles of
        // one token, i.e. the float4 groups lq and 4 + lq of the workgroup's 32-row block of h: for a Q4_0 consumer those
        // are exactly the two groups of k-slot group w = lq (slot_offsets), for a linear consumer neighbouring lq pairs
        // swap one group.  The fragments go straight from registers to the store: no f32 gate / up / h in memory.
        const int hblk = blockIdx.x;
        if (hblk * 32 >= P.rows) return;
#pragma unroll
        for (int t = 0; t < NTW; t++) {
            const int n = (ttile0 + t) * 16 + li;
            float hv[2][4];
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float gv = acc[r][t][j], uv = acc[r + 2][t][j];
                    const float ex = exp_f64_as_f32(-gv);
                    hv[r][j] = (gv / (1.0f + ex)) * uv;
                }
            float v[8];
            int w;
            if (P.out_q4) {
                w = lq;
                slots_from(1, make_float4(hv[0][0], hv[0][1], hv[0][2], hv[0][3]), make_float4(hv[1][0], hv[1][1], hv[1][2], hv[1][3]), v);
            } else {
                // groups 2w, 2w+1 make k-slot group w: even lq keeps tile 0's group and takes its odd neighbour's, odd lq the reverse
                float got[4];
#pragma unroll
                for (int j = 0; j < 4; j++) got[j] = __shfl_xor((lq & 1) ? hv[0][j] : hv[1][j], 16);
                w = (lq & 1) ? 2 + (lq >> 1) : (lq >> 1);
                const float4 a = (lq & 1) ? make_float4(got[0], got[1], got[2], got[3]) : make_float4(hv[0][0], hv[0][1], hv[0][2], hv[0][3]);
                const float4 b = (lq & 1) ? make_float4(hv[1][0], hv[1][1], hv[1][2], hv[1][3]) : make_float4(got[0], got[1], got[2], got[3]);
                slots_from(0, a, b, v);
            }
            if (n < P.n_tokens) store_frag(P.xf_out, P.nt16, n, hblk, w, v);
        }
        return;
    }
    if constexpr (EPI == QG_EPI_ROPE) {
        // RoPE (go/model.go:449-477) + attention biases (:525-527) + KV store (:552-554) on the accumulators: tile rows 0-7
        // hold element i, rows 8-15 element i + hd/2 of one head (ROWMAP_HEADPERM), so a value's rotation partner sits in
        // lane ^ 32 (rows 4*lq + j <-> 4*(lq ^ 2) + j, same token).
        const QGemmParams::Rope &R = P.rope;
        const int hd = R.head_dim, half = hd >> 1, tph = hd / 16, nq = R.n_q_heads * hd;
        int pos[NTW], strm[NTW];
#pragma unroll
        for (int t = 0; t < NTW; t++) {
            const int nn = min((ttile0 + t) * 16 + li, P.n_tokens - 1);
            pos[t] = R.pos[nn];
            strm[t] = R.stream[nn];
        }
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            const int tile = min(tile0 + rt, P.ntiles - 1);   // (a tile past the end: computed, never stored)
            const int head = tile / tph, i0 = (tile % tph) * 8 + 4 * (lq & 1), e0 = i0 + (lq >> 1) * half;
            const bool is_q = head < R.n_q_heads, is_k = !is_q && head < R.n_q_heads + R.n_kv_heads;
            const int kvh = head - R.n_q_heads - (is_k ? 0 : R.n_kv_heads);
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (R.bias_q) bv = *reinterpret_cast<const float4 *>((is_q ? R.bias_q + head * hd : is_k ? R.bias_k + kvh * hd : R.bias_v + kvh * hd) + e0);
            const float bj[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
            for (int t = 0; t < NTW; t++) {
                const int n = (ttile0 + t) * 16 + li;
                const float4 c4 = *reinterpret_cast<const float4 *>(R.cos + pos[t] * half + i0), s4 = *reinterpret_cast<const float4 *>(R.sin + pos[t] * half + i0);
                const float cj[4] = {c4.x, c4.y, c4.z, c4.w}, sj[4] = {s4.x, s4.y, s4.z, s4.w};
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float v = acc[rt][t][j] + bj[j];
                    const float partner = __shfl_xor(v, 32);
                    float outv = v;
                    if (is_q || is_k) {
                        const float x0 = lq < 2 ? v : partner, x1 = lq < 2 ? partner : v;
                        if (!R.conj) outv = lq < 2 ? (x0 * cj[j] - x1 * sj[j]) : (x0 * sj[j] + x1 * cj[j]);
                        else outv = lq < 2 ? (x0 * cj[j] + x1 * sj[j]) : (-x0 * sj[j] + x1 * cj[j]);
                    }
                    o[j] = outv;
                }
                if (n >= P.n_tokens || tile0 + rt >= P.ntiles) continue;
                float *dstp = is_q ? R.q + ((long long)n * nq + head * hd + e0)
                                   : (is_k ? R.kcache : R.vcache) + ((long long)strm[t] * R.kv_stream_stride + ((long long)kvh * R.seq_len + pos[t]) * hd + e0);
                *reinterpret_cast<float4 *>(dstp) = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
        return;
    }
    const bool split = P.ksplit > 1;
    float *const dst = split ? P.part + (long long)blockIdx.z * P.n_tokens * P.ldo : P.out;
    const float *const resid = split ? nullptr : P.resid, *const bias = split ? nullptr : P.bias;
    const bool vec = (P.ldo & 3) == 0;     // rows of four land on 16-byte boundaries
    if (vec && tile0 + RT <= P.ntiles && row0 + RT * TR <= P.rows) {
        // whole row tiles (every workgroup but a ragged last one): the residual / bias operands of the tile are requested
        // first with clamped token indices -- ONE branch around all the loads, one memory latency -- then added and stored
        float4 rv[NTW][RT], bv[RT];
        size_t off[NTW][RT];
#pragma unroll
        for (int t = 0; t < NTW; t++)
#pragma unroll
            for (int rt = 0; rt < RT; rt++)
                off[t][rt] = (size_t)min((ttile0 + t) * 16 + li, P.n_tokens - 1) * P.ldo + (row0 + rt * TR + lq * 4);
        if (resid) {
#pragma unroll
            for (int t = 0; t < NTW; t++)
#pragma unroll
                for (int rt = 0; rt < RT; rt++) rv[t][rt] = *reinterpret_cast<const float4 *>(resid + off[t][rt]);
        }
        if (bias) {
#pragma unroll
            for (int rt = 0; rt < RT; rt++) bv[rt] = *reinterpret_cast<const float4 *>(bias + row0 + rt * TR + lq * 4);
        }
#pragma unroll
        for (int t = 0; t < NTW; t++)
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                float4 v = make_float4(acc[rt][t][0], acc[rt][t][1], acc[rt][t][2], acc[rt][t][3]);
                if (bias) { v.x += bv[rt].x; v.y += bv[rt].y; v.z += bv[rt].z; v.w += bv[rt].w; }
                if (resid) { v.x += rv[t][rt].x; v.y += rv[t][rt].y; v.z += rv[t][rt].z; v.w += rv[t][rt].w; }
                if ((ttile0 + t) * 16 + li < P.n_tokens) *reinterpret_cast<float4 *>(dst + off[t][rt]) = v;
                acc[rt][t] = (f32x4_t){v.x, v.y, v.z, v.w};
            }
        if constexpr (RT % 2 == 0) {
            if (P.nrm_out.w && !split) {
                // folded RMSNorm, producer side (QGemmParams::NormOut): acc now holds rows row0 + 16 rt + 4 lq .. + 3 of the new
                // residual stream for token (ttile0 + t) * 16 + li.  Sum of squares of the workgroup's 64 rows per token
                // (float64, fixed order: a lane's 16 values, then the four lanes of a token), and x * g as the consumer's
                // fragments -- tiles 2b and 2b + 1 are one 32-column block, and for a Q4_0 consumer the two float4 groups of
                // k-slot group lq are exactly this lane's (as in the SwiGLU epilogue above).
                float4 gw[RT];
#pragma unroll
                for (int rt = 0; rt < RT; rt++) gw[rt] = *reinterpret_cast<const float4 *>(P.nrm_out.w + row0 + rt * TR + lq * 4);
                float psc[NTW];      // exact power of two near 1 / rms of the token (norm_prescale, nl_qgemm.h)
#pragma unroll
                for (int t = 0; t < NTW; t++) psc[t] = P.nrm_out.scale ? P.nrm_out.scale[min((ttile0 + t) * 16 + li, P.n_tokens - 1)] : 1.0f;
#pragma unroll
                for (int t = 0; t < NTW; t++) {
                    const int n = (ttile0 + t) * 16 + li;
                    double ss = 0.0;
#pragma unroll
                    for (int rt = 0; rt < RT; rt++)
#pragma unroll
                        for (int j = 0; j < 4; j++) ss = fma((double)acc[rt][t][j], (double)acc[rt][t][j], ss);
                    ss += __shfl_xor(ss, 16);
                    ss += __shfl_xor(ss, 32);
                    if (lq == 0 && n < P.n_tokens) P.nrm_out.ssq[(size_t)n * (P.rows / 64) + blockIdx.x * (RT / 4)] = ss;
#pragma unroll
                    for (int b = 0; b < RT / 2; b++) {
                        float y[2][4];
#pragma unroll
                        for (int r = 0; r < 2; r++) {
                            const float4 g = gw[2 * b + r];
                            y[r][0] = (acc[2 * b + r][t][0] * g.x) * psc[t]; y[r][1] = (acc[2 * b + r][t][1] * g.y) * psc[t];
                            y[r][2] = (acc[2 * b + r][t][2] * g.z) * psc[t]; y[r][3] = (acc[2 * b + r][t][3] * g.w) * psc[t];
                        }
                        // (the consumer is a qgemm2 GEMM, i.e. Q4_0: k-slot group lq = this lane's float4 groups of the two tiles)
                        float v[8];
                        const int w = lq;
                        slots_from(1, make_float4(y[0][0], y[0][1], y[0][2], y[0][3]), make_float4(y[1][0], y[1][1], y[1][2], y[1][3]), v);
                        if (n < P.n_tokens) store_frag(P.nrm_out.xf, P.nt16, n, row0 / 32 + b, w, v);
                    }
                }
            }
        }
        QG2_STAMP(2 + it_ * 8);
        return;
    }
#pragma unroll
    for (int t = 0; t < NTW; t++) {
        const int n = (ttile0 + t) * 16 + li;
        if (n >= P.n_tokens) continue;
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            const int row = row0 + rt * TR + lq * 4;
            if (tile0 + rt >= P.ntiles || row >= P.rows) continue;
            const size_t off = (size_t)n * P.ldo + row;
            if (vec && row + 3 < P.rows) {
                float4 v = make_float4(acc[rt][t][0], acc[rt][t][1], acc[rt][t][2], acc[rt][t][3]);
                if (bias) { const float4 bv = *reinterpret_cast<const float4 *>(bias + row); v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w; }
                if (resid) { const float4 rv = *reinterpret_cast<const float4 *>(resid + off); v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w; }
                *reinterpret_cast<float4 *>(dst + off) = v;
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (row + j >= P.rows) continue;
                    float v = acc[rt][t][j] + (bias ? bias[row + j] : 0.f);
                    if (resid) v += resid[off + j];
                    dst[off + j] = v;
                }
            }
        }
    }
    QG2_STAMP(2 + it_ * 8);
}

}  // namespace nl
