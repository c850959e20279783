// nl_dgemm.h -- the multi-token GEMM for SHORT token runs (decode batches of <= 64 streams, go/serve.go's concurrent requests
// stepped together; Q4_0): one launch per projection, no split-K launch, no reduce launch, every epilogue of the layer fused.
//
// What bounds a GEMM of 64 tokens on this chip is neither HBM (a goldie layer is 14 MB: 2 us) nor the matrix cores, but
// what ONE compute unit can ingest (64 B per clock from L2) and issue.  The activation matrix of 64 tokens is 4 bytes per
// element (fp16 hi + lo, nl_qgemm.h) -- 393 KB for K = 1536 -- so a workgroup that sees all 64 tokens and all of K needs 2.6 us
// just to read its B operand; qgemm_kernel therefore split K over workgroups (<= 6 chunks each), wrote partial slabs and left
// the sum, the norm, the rotation and SiLU to four more launches per layer: 8 launches of 5-9 us.
// Here the TOKEN TILE is what a workgroup keeps: 16 tokens x all of K (98 KB for K = 1536) against T 16-row weight tiles, so a
// result element is finished inside one workgroup and the qgemm2 epilogues (RoPE + KV store, SiLU(gate) * up as the next
// GEMM's fragments, residual + the folded RMSNorm of the next GEMM) apply as they are.  The four workgroups that share a row
// group (one per token tile) sit on the same XCD (block b runs on XCD b % 8, observed -- for speed only), so a weight byte
// crosses the fabric once and is an L2 hit for the other three.
//
//   * K is split over the workgroup's wavefronts by quant block: wavefront w owns block w % 8 of every 8-block group it visits
//     and multiplies it against ALL T row tiles, so an activation fragment is read by exactly one wavefront and NOTHING is
//     shared inside the K loop: no barrier.  (The first form shared a chunk ring between all wavefronts behind one barrier per
//     chunk: with 1-2 (block, tile) products per wavefront between barriers it ran at 140 SIMD cycles per product.)
//   * Everything a wavefront multiplies is requested at its first instruction: its NS activation fragment pairs straight into
//     registers (global_load_dwordx4 sc1: the fragment store is in MFMA operand order, and a load that skips the vector L1
//     streams L2-resident data at 110 GB/s per compute unit against 37 through the L1 and 25-45 measured here through
//     LDS-DMA: EXPERIMENTS, tools/ingest_probe.hip), its NS x T x (256 B of nibbles + 64 B of fp16 scale words) by LDS-DMA
//     (the source picked per lane) into its own LDS area.  The issue order is static, so "step s has landed" is one
//     s_waitcnt vmcnt(immediate) on the wavefront's own counter (loads return in order).
//   * The nibbles of (block, tile) land as [row][16 B]: the B-operand read -- dword lq of row li -- is conflict-free.  The
//     MFMA runs as D[token][row] = x . W^T, so a lane needs ONE scale per (block, tile): 1 ds_read_b32, 4 v_fma_mix.
//     Per (block, tile): 9 VALU (nibbles -> fp16), 2 MFMA (x = hi + lo), 4 FMA ("* d"), 2 + 2/T LDS reads.
//   * The NWV partial accumulators of a tile meet in LDS (one barrier per launch) in wavefront order and are read back
//     transposed, as D[row][token] = the layout qgemm2_kernel's epilogues are written for.
// Arithmetic per output = qgemm_kernel / qgemm2_kernel (exact integer quants in fp16, f32 block sums, * d in f32); the
// summation order over blocks is block-interleaved (NWV partial sums), inside the stated tolerances; deterministic.
// hipcc waits vmcnt(0) for a register load once LDS-DMA pieces are in flight behind it (it treats the two as unordered): the
// fragment loads are inline asm with hand-placed waits, and the epilogue operands (compiler-tracked, requested first) are first
// used behind the loop.
// Reference: go/quant.go:45-94 (MatMulQ4_0), go/model.go:513-613 (the layer), per stream.
#pragma once
#include <algorithm>
#include "nl_qgemm2.h"

namespace nl {

constexpr int DG_LDS_BYTES = 160 * 1024;
constexpr int DG_SSQ_MAX_NRB = 64;              // partial sums of squares per token the consumer side takes (dim <= 2048)
constexpr int DG_SSQ_BYTES = 16 * DG_SSQ_MAX_NRB * 8;

template <int T, int NWV, int NS, bool SSQ> struct DgLds {
    static constexpr int NP = (16 * T + 63) / 64;               // 64-lane pieces that cover the T x 16 rows of a block
    static constexpr int REG_U4 = 16 * T + 4 * T;               // one step of one wavefront: nibbles [T][16 rows][16 B] | scales [T][16 rows] fp16 pair words
    static constexpr int SSQ_U4 = SSQ ? DG_SSQ_BYTES / 16 : 0;
    static constexpr int WAVE_U4 = NS * REG_U4 > T * 64 ? NS * REG_U4 : T * 64;     // ... or its T partial accumulator tiles behind the loop
    static constexpr int TOTAL_U4 = SSQ_U4 + NWV * WAVE_U4;
};

#ifdef DG_STAMPS
__device__ long long g_dg_stamps[64];
__device__ long long g_dg_census[2 * 2048];      // wall clock (100 MHz) at entry / exit of wavefront 0 of every workgroup
#define DG_LIN (blockIdx.y * gridDim.x + blockIdx.x)
// (inside the engine -- tools/dg_stamps.sh builds a copy of the library with -DDG_STAMPS -DDG_STAMP_EPI=<epilogue> -- only the launches
//  of that epilogue stamp, so that the numbers read back after a step are the last layer's launch of that kind)
#ifndef DG_STAMP_EPI
#define DG_STAMP_EPI -1
#endif
#define DG_STAMP(i) do { if (DG_STAMP_EPI < 0 || DG_STAMP_EPI == DG_THIS_EPI) { \
    if (DG_LIN == 9 && threadIdx.x == 0 && (i) < 64) g_dg_stamps[(i)] = clock64(); \
    if (((i) == 0 || (i) == 40) && threadIdx.x == 0 && DG_LIN < 2048) g_dg_census[2 * DG_LIN + ((i) ? 1 : 0)] = wall_clock64(); } } while (0)
// ... and a log of every launch: entry and exit (wall clock, 10 ns) of workgroup 0's first wavefront, in launch order
__device__ long long g_dg_log[4 * 4096];          // [launch][entry, barrier, end of wavefront 0's epilogue, -]
__device__ unsigned g_dg_log_n;
#define DG_LOG_ENTRY() [[maybe_unused]] unsigned dg_log_i_ = 0xffffffffu; \
    do { if (DG_LIN == 0 && threadIdx.x == 0) { dg_log_i_ = atomicAdd(&g_dg_log_n, 1u) & 4095u; g_dg_log[4 * dg_log_i_] = wall_clock64(); } } while (0)
#define DG_LOG_EXIT() do { if (dg_log_i_ != 0xffffffffu) g_dg_log[4 * dg_log_i_ + 1] = wall_clock64(); } while (0)
#define DG_LOG_END() do { if (dg_log_i_ != 0xffffffffu) g_dg_log[4 * dg_log_i_ + 2] = wall_clock64(); } while (0)
#else
#define DG_STAMP(i) do { } while (0)
#define DG_LOG_ENTRY() do { } while (0)
#define DG_LOG_EXIT() do { } while (0)
#define DG_LOG_END() do { } while (0)
#endif

// cache policy of the LDS-DMA pieces (CPol bits of global_load_lds: 1 = sc0, 2 = nt, 16 = sc1).  The vector L1's fill path gives
// a compute unit 37 GB/s of L2-resident data whatever is in flight; loads that skip the L1 (a scope bit) reach 110 (EXPERIMENTS,
// tools/ingest_probe.hip) -- and every byte here is used once per compute unit, so the L1 has nothing to offer.
#ifndef DG_AUX
#define DG_AUX 16
#endif

// vmcnt immediate of s_waitcnt on gfx9 (vmcnt[3:0] | expcnt[6:4] = 7 | lgkmcnt[11:8] = 15 | vmcnt[5:4] in [15:14])
#define DG_WAIT_VM(n) __builtin_amdgcn_s_waitcnt(0x0f70 | ((n) & 15) | (((n) >> 4) << 14))

typedef unsigned dg_u32x4 __attribute__((ext_vector_type(4)));
// 1 / sqrt(ss / dim + eps) as float32 (go/quant.go:597-607 computes it in float64): v_rsq_f64 + two Newton steps -- relative error
// < 1e-30 before the rounding to float32, the same float32 as the IEEE divide and square root give except at rounding boundaries of
// measure zero -- instead of sixty dependent float64 instructions behind the last product of the launch (nl_persist.h pd_inv_rms)
__device__ __forceinline__ float dg_inv_rms(double ss, int dim, float eps) {
    const double rdim = 1.0 / (double)dim;                 // (independent of ss: off the critical path)
    const double m = fma(ss, rdim, (double)eps);
    double y = __builtin_amdgcn_rsq(m);
    y = y * fma(-0.5 * m * y, y, 1.5);
    y = y * fma(-0.5 * m * y, y, 1.5);
    return (float)y;
}
template <int I, int N, class F>
__device__ __forceinline__ void dg_static_for(F &&f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); dg_static_for<I + 1, N>(f); }
}

// T: row tiles of a workgroup (SWIGLU: T / 2 gate tiles + the same T / 2 tiles of up); NWV: wavefronts (8 or 16); NS: quant
// blocks per wavefront = cols / 32 / NWV exactly.  grid.x = ceil(row groups / 8) * 8 * token tiles (dg_grid).
template <int T, int NWV, int NS, int EPI>
__global__ void __launch_bounds__(NWV * 64, NWV / 4) dgemm_kernel(QGemmParams P) {
    typedef DgLds<T, NWV, NS, EPI != QG_EPI_PLAIN> L;
    constexpr int REG_U4 = L::REG_U4, SSQ_U4 = L::SSQ_U4, WAVE_U4 = L::WAVE_U4;
    constexpr int NP = L::NP, PPS = 2 + 2 * NP;                  // memory operations per step: 2 fragment loads, NP nibble pieces (16 B per lane), NP scale pieces (4 B per lane)
    constexpr bool FUSED = EPI == QG_EPI_SWIGLU;
    constexpr int EG = EPI == QG_EPI_ROPE ? 1 : 2;               // tiles per epilogue group (one wavefront each; FUSED: a gate tile and its up tile)
    constexpr int NEG = T / EG;
    constexpr int TPG = FUSED ? T / 2 : T;                       // distinct row tiles of the workgroup per matrix
    static_assert(NWV == 8 || NWV == 16, "a wavefront owns block w % 8 of the groups it visits");
    static_assert((NS - 1) * PPS <= 63, "vmcnt is six bits");
    static_assert(T % EG == 0 && NEG <= NWV, "epilogue groups");
    __shared__ __attribute__((aligned(16))) uint4 lds_all[L::TOTAL_U4];
    [[maybe_unused]] constexpr int DG_THIS_EPI = EPI + (EPI == QG_EPI_PLAIN && NWV == 16 ? 10 : 0);      // (stamps: 0 WO, 10 down, 1 gate || up, 2 Q|K|V)
    DG_STAMP(0);
    DG_LOG_ENTRY();
    // every kernel argument the launch reads, requested as ONE batch of s_load (hipcc otherwise fetches each field before its first
    // use with a wait behind it: the prologue was fourteen dependent scalar round trips, 2 of the launch's 4.5 us)
    NL_KARGS8(P.q, P.s, P.xf, P.q1, P.s1, P.out, P.resid, P.bias);
    NL_KARGS8(P.rows, P.cols, P.npairs, P.ntiles, P.nt16, P.n_tokens, P.ldo, P.xf_out);
    if constexpr (EPI != QG_EPI_PLAIN) NL_KARGS8(P.nrm_in.ssq, P.nrm_in.scale, P.nrm_in.scale_next, P.nrm_in.nrb, P.nrm_in.dim, P.nrm_in.eps, P.q, P.s);
    if constexpr (EPI == QG_EPI_PLAIN) NL_KARGS8(P.nrm_out.w, P.nrm_out.xf, P.nrm_out.ssq, P.nrm_out.scale, P.q, P.s, P.xf, P.out);
    if constexpr (EPI == QG_EPI_ROPE) {
        NL_KARGS8(P.rope.tcos, P.rope.tsin, P.rope.tkv, P.rope.q, P.rope.kcache, P.rope.vcache, P.rope.bias_q, P.rope.bias_k);
        NL_KARGS8(P.rope.bias_v, P.rope.head_dim, P.rope.n_q_heads, P.rope.n_kv_heads, P.rope.seq_len, P.rope.conj, P.q, P.s);
    }

    const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int li = lane & 15, lq = lane >> 4;
    uint4 *const ring = lds_all + SSQ_U4 + wv * WAVE_U4;      // this wavefront's area
    // block -> (row group, token tile): gridDim.x = 8 x token tiles, blockIdx.y = row group / 8 -- the linear dispatch order puts the
    // token tiles of one row group on the same XCD (block b runs on XCD b % 8), next to each other; no division
    const int rg = ((int)blockIdx.x & 7) + 8 * (int)blockIdx.y, tt = (int)blockIdx.x >> 3;
    if (rg * TPG >= P.ntiles) return;                            // (padding of the row groups to a multiple of 8: the whole workgroup)
    const int tile_base = rg * TPG;                              // first row tile of the workgroup (FUSED: of gate and of up)
    const int nblocks = P.cols >> 5;
    // this wavefront's blocks: g = wv, wv + NWV, ... (block g % 8 = wv % 8 of group g / 8): NS of them
    const int blk0 = wv, k0 = (blk0 >> 1) & 3, cc = blk0 & 1;      // pair of the group, block of the pair
    // ---- per-lane source pointers of this wavefront's loads, advanced step by step ----
    const unsigned xblock = (unsigned)P.nt16 * (2 * QG_FRAG * 16);    // bytes of one block's fragments
    const char *xsrc = reinterpret_cast<const char *>(P.xf) + ((size_t)blk0 * xblock + ((size_t)(tt * 2) * QG_FRAG + lane) * 16);
    auto tile_of = [&](int ti, int &mat) {          // entry ti of the workgroup's weight area -> (row tile, matrix)
        mat = FUSED ? ti / (T / 2) : 0;
        return min(tile_base + (FUSED ? ti % (T / 2) : ti), P.ntiles - 1);
    };
    const char *wsrc[NP], *ssrc[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) {
        // lane -> (tile, row) of the workgroup: the 16 nibble bytes of this block, and the scale word of its pair
        const int u = min(i * 64 + lane, 16 * T - 1), ti = u >> 4, r = u & 15;
        int mat;
        const int tile = tile_of(ti, mat);
        const size_t grp = (size_t)tile * P.npairs + (size_t)(blk0 >> 3) * KL;
        // (block-major copies of the packed matrix, dg_permute_kernel: the 16 rows of a block are 256 / 64 contiguous bytes.  In
        //  the decode GEMV's order -- (parity, row, pair) -- a lane's 16 bytes sit 64 bytes apart and every request moves a
        //  whole sector for a quarter of it: the nibbles cost four times their size on the way in)
        wsrc[i] = reinterpret_cast<const char *>(mat ? P.q1 : P.q) + (grp * (2 * TR) + (size_t)((k0 * 2 + cc) * TR + r)) * 16;
        ssrc[i] = reinterpret_cast<const char *>(mat ? P.s1 : P.s) + (grp * TR + (size_t)(k0 * TR + r)) * 4;
    }
    dg_u32x4 xh[NS], xl[NS];
    auto issue = [&](int s) {
        uint4 *const base = ring + s * REG_U4;
        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(xh[s]) : "v"(xsrc) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:1024 sc1" : "=&v"(xl[s]) : "v"(xsrc) : "memory");
        xsrc += (size_t)NWV * xblock;
        // (lanes past the T x 16 rows are masked off: they neither request nor write)
#pragma unroll
        for (int i = 0; i < NP; i++) {
            if ((i + 1) * 64 <= 16 * T || i * 64 + lane < 16 * T)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)wsrc[i],
                                                 (__attribute__((address_space(3))) void *)(base + i * 64), 16, 0, DG_AUX);
            wsrc[i] += KL * 2 * TR * 16 * (NWV / 8);        // the same block of this wavefront's next group
        }
#pragma unroll
        for (int i = 0; i < NP; i++) {
            if ((i + 1) * 64 <= 16 * T || i * 64 + lane < 16 * T)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)ssrc[i],
                                                 (__attribute__((address_space(3))) void *)(base + 16 * T + i * 16), 4, 0, DG_AUX);
            ssrc[i] += KL * TR * 4 * (NWV / 8);
        }
    };

    // ---- what the epilogue needs from memory is requested FIRST (by the wavefronts that run an epilogue group) and first USED
    //      behind the K loop; the RoPE position of the lane's token comes through the scalar cache (lgkmcnt) ----
    const int n = tt * 16 + li, nn = min(n, P.n_tokens - 1);   // epilogue: this lane's token; its rows are 4 * lq .. + 3 of every tile
    const bool live = n < P.n_tokens;
    const int etile0 = tile_base + (wv < NEG ? (FUSED ? wv : wv * EG) : 0);      // first row tile of this wavefront's epilogue group (wv < NEG; FUSED: the h tile)
    [[maybe_unused]] float nsc = 1.0f, psc = 1.0f;
    [[maybe_unused]] long long kvoff = 0;       // ROPE: stream * kv_stream_stride + pos * hd of the lane's token
    [[maybe_unused]] float4 rv[EG], bv[EG], gw[EG], rc4, rs4;
    const float *const dummy = reinterpret_cast<const float *>(P.q);     // (an absent operand reads the first bytes of the weights -- value
                                                                         //  unused -- so that no load sits behind a branch)
    {   // (every wavefront: a load inside a branch would be waited for at the end of the branch)
        if constexpr (EPI != QG_EPI_PLAIN) nsc = *(P.nrm_in.ssq && P.nrm_in.scale ? P.nrm_in.scale + nn : dummy);
        if constexpr (EPI == QG_EPI_ROPE) {
            const int hd = P.rope.head_dim, half = hd >> 1, tsh = hd == 64 ? 2 : 1;      // (head_dim is 32 or 64: tiles per head 2 or 4)
            const int tile = min(etile0, P.ntiles - 1), i0 = (tile & ((1 << tsh) - 1)) * 8 + 4 * (lq & 1);
            rc4 = *reinterpret_cast<const float4 *>(P.rope.tcos + (size_t)nn * half + i0);
            rs4 = *reinterpret_cast<const float4 *>(P.rope.tsin + (size_t)nn * half + i0);
            kvoff = P.rope.tkv[nn];
        }
        if constexpr (EPI == QG_EPI_PLAIN) {
            const int row0 = min(etile0, P.ntiles - EG) * TR;
            const size_t off0 = (size_t)nn * P.ldo + (row0 + lq * 4);
#pragma unroll
            for (int rt = 0; rt < EG; rt++) {
                rv[rt] = *reinterpret_cast<const float4 *>(P.resid ? P.resid + off0 + rt * TR : dummy);
                bv[rt] = *reinterpret_cast<const float4 *>(P.bias ? P.bias + row0 + rt * TR + lq * 4 : dummy);
                gw[rt] = *reinterpret_cast<const float4 *>(P.nrm_out.w ? P.nrm_out.w + row0 + rt * TR + lq * 4 : dummy);
            }
            psc = *(P.nrm_out.w && P.nrm_out.scale ? P.nrm_out.scale + nn : dummy);
        }
    }
    asm volatile("" ::: "memory");
    if constexpr (EPI != QG_EPI_PLAIN) {
        // the producer's partial sums of squares of this tile's 16 tokens ([token][nrb] float64, contiguous) into LDS, older than
        // every step's pieces; read behind the loop
        if (P.nrm_in.ssq) {
            const unsigned bytes = 16u * (unsigned)P.nrm_in.nrb * 8u;
            const char *const src = reinterpret_cast<const char *>(P.nrm_in.ssq + (size_t)tt * 16 * P.nrm_in.nrb);
            for (unsigned p0 = (unsigned)wv * 1024u; p0 < bytes; p0 += NWV * 1024u)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + min(p0 + lane * 16u, bytes - 16u)),
                                                 (__attribute__((address_space(3))) void *)(lds_all + p0 / 16), 16, 0, DG_AUX);
        }
    }
    // all NS steps: fragments, nibbles, scales, in that order per step
#pragma unroll
    for (int s = 0; s < NS; s++) issue(s);
    asm volatile("" ::: "memory");
    DG_STAMP(1);

    f32x4_t acc[T];
#pragma unroll
    for (int ti = 0; ti < T; ti++) acc[ti] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    dg_static_for<0, NS>([&](auto s_) {
        constexpr int s = decltype(s_)::value;
        // step s has landed when at most the operations of the steps issued after it are outstanding
        DG_WAIT_VM((NS - 1 - s) * PPS);
        asm volatile("" : "+v"(xh[s]), "+v"(xl[s]) :: "memory");          // (the fragment registers are read from here on)
        const uint32_t *const sb32 = reinterpret_cast<const uint32_t *>(ring + s * REG_U4);
        const half8_t xhv = __builtin_bit_cast(half8_t, xh[s]), xlv = __builtin_bit_cast(half8_t, xl[s]);
        // TG tiles at a time: TG independent MFMA chains (lo product, then hi product on the same accumulator)
        constexpr int TG = T == 8 ? 2 : T % 4 == 0 ? 4 : T % 3 == 0 ? 3 : T % 2 == 0 ? 2 : 1;
#pragma unroll
        for (int t0 = 0; t0 < T; t0 += TG) {
            half8_t a[TG];
            uint32_t sw[TG];
            f32x4_t z[TG];
#pragma unroll
            for (int g = 0; g < TG; g++) {
                a[g] = WFrag<WT_Q4_0>::expand(sb32[((t0 + g) * 16 + li) * 4 + lq]);
                sw[g] = sb32[(16 * T) * 4 + (t0 + g) * 16 + li];
            }
#pragma unroll
            for (int g = 0; g < TG; g++) z[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xlv, a[g], (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
            for (int g = 0; g < TG; g++) z[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xhv, a[g], z[g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < TG; g++) {
                const float d = scale_of(sw[g], cc);
#pragma unroll
                for (int r = 0; r < 4; r++) acc[t0 + g][r] = fmaf(z[g][r], d, acc[t0 + g][r]);
            }
        }
        DG_STAMP(2 + s);
    });
    // ---- the NWV partial accumulators of a tile meet in LDS: D[token 4 lq + r][row li] goes out as [tile][token][row], the
    //      epilogue wavefronts read [row 4 lq ..][token li] back, partials in wavefront order ----
    if constexpr (T * 64 > NS * REG_U4) __syncthreads();      // (the partial tiles of a wavefront reach into its neighbour's weights)
    {
        float *const mine = reinterpret_cast<float *>(ring);
#pragma unroll
        for (int ti = 0; ti < T; ti++)
#pragma unroll
            for (int r = 0; r < 4; r++) mine[ti * 256 + (4 * lq + r) * 16 + li] = acc[ti][r];
    }
    __syncthreads();
    DG_STAMP(40);
    DG_LOG_EXIT();
    if (wv >= NEG) return;
    f32x4_t e[EG];
#pragma unroll
    for (int g = 0; g < EG; g++) {
        // epilogue group wv: PLAIN / ROPE tiles wv * EG + g; FUSED: gate tile wv, then up tile wv (four wavefronts share the
        // SiLUs of a workgroup -- float64 exponentials: two wavefronts with eight each were 3.5 us behind the last product)
        const int ti = FUSED ? g * (T / 2) + wv : wv * EG + g;
        e[g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int w = 0; w < NWV; w++)      // (four reads in flight: fully unrolled, hipcc hoists all NWV x EG reads -- 256 registers)
            e[g] += *reinterpret_cast<const f32x4_t *>(reinterpret_cast<const float *>(lds_all + SSQ_U4 + w * WAVE_U4) + ti * 256 + li * 16 + lq * 4);
    }
    // ---- folded RMSNorm, consumer side (QGemmParams::NormIn): inv of this lane's token from the producer's per-32-row sums of
    //      squares: lane lq adds partials lq, lq + 4, ... in ascending order, the four lanes of a token meet in lq order ----
    if constexpr (EPI != QG_EPI_PLAIN) {
        float inv = 1.0f;
        if (P.nrm_in.ssq) {
            asm volatile("" : "+v"(nsc));          // (first use HERE, behind the loop)
            const double *const sq = reinterpret_cast<const double *>(lds_all) + li * P.nrm_in.nrb;
            double tot = 0.0;
            for (int r0 = lq; r0 < P.nrm_in.nrb; r0 += 16) {      // (four reads in flight; added in ascending order)
                double v4[4];
#pragma unroll
                for (int u = 0; u < 4; u++) v4[u] = sq[min(r0 + 4 * u, P.nrm_in.nrb - 1)];
#pragma unroll
                for (int u = 0; u < 4; u++) tot += r0 + 4 * u < P.nrm_in.nrb ? v4[u] : 0.0;
            }
            const double t1 = __shfl_xor(tot, 16);
            const double lo2 = (lq & 1) ? t1 + tot : tot + t1;          // (lq 0 + lq 1) resp. (lq 2 + lq 3), same order in both lanes
            const double t2 = __shfl_xor(lo2, 32);
            tot = (lq & 2) ? t2 + lo2 : lo2 + t2;
            inv = dg_inv_rms(tot, P.nrm_in.dim, P.nrm_in.eps);
            if (P.nrm_in.scale) {     // the producer's power-of-two pre-scale (norm_prescale, nl_qgemm.h): undone exactly
                if (rg == 0 && wv == 0 && lq == 0 && live) P.nrm_in.scale_next[n] = norm_prescale(inv);
                inv *= 1.0f / nsc;
            }
        }
#pragma unroll
        for (int g = 0; g < EG; g++) e[g] = e[g] * inv;
    }
    if constexpr (EPI == QG_EPI_SWIGLU) {
        // h = SiLU(gate) * up (go/quant.go:629-631, go/model.go:604-606): rows 4*lq..+3 of this wavefront's 16-row tile of one token are
        // one float4 group of k-slot group lq of the tile's 32-row block of h -- half a fragment entry (store_frag_half)
        if (etile0 * TR >= P.rows || !live) return;
        float hv[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float gv = e[0][j], uv = e[1][j];
            const float ex = exp_f64_as_f32(-gv);
            hv[j] = (gv / (1.0f + ex)) * uv;
        }
        store_frag_half(P.xf_out, P.nt16, n, etile0 >> 1, lq, etile0 & 1, make_float4(hv[0], hv[1], hv[2], hv[3]));
        DG_LOG_END();
        return;
    }
    if constexpr (EPI == QG_EPI_ROPE) {
        // RoPE (go/model.go:449-477) + attention biases (:525-527) + KV store (:552-554), as qgemm2_kernel's epilogue: the
        // rotation partner of rows 4*lq + j is rows 4*(lq ^ 2) + j of the same tile and token: lane ^ 32
        const QGemmParams::Rope &R = P.rope;
        const int hd = R.head_dim, half = hd >> 1, tsh = hd == 64 ? 2 : 1, nq = R.n_q_heads * hd;
        const int tile = min(etile0, P.ntiles - 1);
        const int head = tile >> tsh, i0 = (tile & ((1 << tsh) - 1)) * 8 + 4 * (lq & 1), e0 = i0 + (lq >> 1) * half;
        const bool is_q = head < R.n_q_heads, is_k = !is_q && head < R.n_q_heads + R.n_kv_heads;
        const int kvh = head - R.n_q_heads - (is_k ? 0 : R.n_kv_heads);
        float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
        if (R.bias_q) bq = *reinterpret_cast<const float4 *>((is_q ? R.bias_q + head * hd : is_k ? R.bias_k + kvh * hd : R.bias_v + kvh * hd) + e0);
        const float bj[4] = {bq.x, bq.y, bq.z, bq.w};
        float4 c4 = rc4, s4 = rs4;
        asm volatile("" : "+v"(c4.x), "+v"(c4.y), "+v"(c4.z), "+v"(c4.w));      // (first use behind the loop)
        asm volatile("" : "+v"(s4.x), "+v"(s4.y), "+v"(s4.z), "+v"(s4.w));
        const float cj[4] = {c4.x, c4.y, c4.z, c4.w}, sj[4] = {s4.x, s4.y, s4.z, s4.w};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float v = e[0][j] + bj[j];
            const float partner = __shfl_xor(v, 32);
            float outv = v;
            if (is_q || is_k) {
                const float x0 = lq < 2 ? v : partner, x1 = lq < 2 ? partner : v;
                if (!R.conj) outv = lq < 2 ? (x0 * cj[j] - x1 * sj[j]) : (x0 * sj[j] + x1 * cj[j]);
                else outv = lq < 2 ? (x0 * cj[j] + x1 * sj[j]) : (-x0 * sj[j] + x1 * cj[j]);
            }
            o[j] = outv;
        }
        if (!live || etile0 >= P.ntiles) return;
        asm volatile("" : "+v"(kvoff));
        float *dstp = is_q ? R.q + ((long long)n * nq + head * hd + e0)
                           : (is_k ? R.kcache : R.vcache) + (kvoff + (long long)kvh * R.seq_len * hd + e0);
        *reinterpret_cast<float4 *>(dstp) = make_float4(o[0], o[1], o[2], o[3]);
        DG_LOG_END();
        return;
    }
    if constexpr (EPI == QG_EPI_PLAIN) {
        // out = resid + bias + y; then (QGemmParams::NormOut) the rows as the NEXT GEMM's fragments, times its norm weights and
        // the token's power-of-two pre-scale, plus the float64 sum of squares of this group's 32 rows (qgemm2_kernel's
        // producer epilogue with one partial per 32-row block)
        const int row0 = etile0 * TR;
        if (etile0 + EG > P.ntiles) return;        // (row counts are multiples of 32 on this path: never taken)
        const size_t off0 = (size_t)nn * P.ldo + (row0 + lq * 4);
#pragma unroll
        for (int rt = 0; rt < EG; rt++) {
            asm volatile("" : "+v"(rv[rt].x), "+v"(rv[rt].y), "+v"(rv[rt].z), "+v"(rv[rt].w));      // (first use behind the loop)
            float4 v = make_float4(e[rt][0], e[rt][1], e[rt][2], e[rt][3]);
            if (P.bias) { v.x += bv[rt].x; v.y += bv[rt].y; v.z += bv[rt].z; v.w += bv[rt].w; }
            if (P.resid) { v.x += rv[rt].x; v.y += rv[rt].y; v.z += rv[rt].z; v.w += rv[rt].w; }
            if (live) *reinterpret_cast<float4 *>(P.out + off0 + rt * TR) = v;
            e[rt] = (f32x4_t){v.x, v.y, v.z, v.w};
        }
        if (P.nrm_out.w) {
            const float ps = P.nrm_out.scale ? psc : 1.0f;
            double ss = 0.0;
#pragma unroll
            for (int rt = 0; rt < EG; rt++)
#pragma unroll
                for (int j = 0; j < 4; j++) ss = fma((double)e[rt][j], (double)e[rt][j], ss);
            ss += __shfl_xor(ss, 16);
            ss += __shfl_xor(ss, 32);
            if (lq == 0 && live) P.nrm_out.ssq[(size_t)n * (P.rows / 32) + (etile0 >> 1)] = ss;
            float y[2][4];
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const float4 g = gw[r];
                y[r][0] = (e[r][0] * g.x) * ps; y[r][1] = (e[r][1] * g.y) * ps;
                y[r][2] = (e[r][2] * g.z) * ps; y[r][3] = (e[r][3] * g.w) * ps;
            }
            float v[8];
            slots_from(1, make_float4(y[0][0], y[0][1], y[0][2], y[0][3]), make_float4(y[1][0], y[1][1], y[1][2], y[1][3]), v);
            if (live) store_frag(P.nrm_out.xf, P.nt16, n, etile0 >> 1, lq, v);
        }
        DG_LOG_END();
    }
}

// The block-major copy dgemm_kernel reads (P.q / P.s of its launches): within every 256-column group of a tile the 16-byte
// chunks go from (parity c, row r, pair k) order to (pair k, parity c, row r), the scale words from (row, pair) to (pair, row).
__global__ void dg_permute_kernel(const uint4 *q, const uint32_t *s, uint4 *q3, uint32_t *s3, long long ngroups) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < ngroups * 128; i += (long long)gridDim.x * blockDim.x) {
        const long long g = i >> 7;
        const int u = (int)(i & 127), k = u >> 5, c = (u >> 4) & 1, r = u & 15;       // destination chunk (k, c, r)
        q3[i] = q[g * 128 + (c * TR + r) * KL + k];
        if (u < 64) s3[g * 64 + u] = s[g * 64 + (u & 15) * KL + (u >> 4)];           // destination word (pair u >> 4, row u & 15)
    }
}

// grid of a dgemm launch: x = 8 x token tiles, y = row groups / 8 (row groups padded to a multiple of 8: the block -> XCD map)
inline dim3 dg_grid(int ntiles, int tiles_per_wg, int n_tokens) {
    const int nrg = (ntiles + tiles_per_wg - 1) / tiles_per_wg;
    return dim3((unsigned)(8 * ((n_tokens + 15) / 16)), (unsigned)((nrg + 7) / 8), 1);
}

// the three launches of a layer: T = 3 (Q|K|V), 8 (gate || up), 2 (WO, down); the wavefront count and the blocks per wavefront
// follow from K.  An instantiation that does not fit the LDS is not made (the caller keeps the split-K launches)
template <int T, int NWV, int NS, int EPI>
inline hipError_t dg_launch(QGemmParams P, hipStream_t st) {
    P.nt16 = ((P.n_tokens + 63) / 64) * 4;
    P.ksplit = 1;
    constexpr int TILES_PER_WG = EPI == QG_EPI_SWIGLU ? T / 2 : T;
    if constexpr (DgLds<T, NWV, NS, EPI != QG_EPI_PLAIN>::TOTAL_U4 * 16 <= DG_LDS_BYTES - 512) {
        hipLaunchKernelGGL((dgemm_kernel<T, NWV, NS, EPI>), dg_grid(P.ntiles, TILES_PER_WG, P.n_tokens), dim3(NWV * 64), 0, st, P);
        return hipGetLastError();
    } else return hipErrorInvalidValue;
}
// wavefronts for a K of nb quant blocks: eight while that leaves a wavefront <= 8 blocks (measured faster than sixteen on every
// K = 1536 launch: 7.85 / 5.7 / 11.75 us against 8.4 / 6.6 / 12.8, tools/dgemm_bench.hip), else sixteen; blocks per wavefront
// in {1, 2, 3, 4, 6, 8}
inline int dg_waves(int nb) { return nb <= 64 ? 8 : 16; }
inline bool dg_cols_ok(int cols, int T, bool ssq) {
    if (cols <= 0 || cols % 256) return false;
    const int nb = cols / 32, nwv = dg_waves(nb), ns = nb / nwv;
    if (nb % nwv || !(ns == 1 || ns == 2 || ns == 3 || ns == 4 || ns == 6 || ns == 8)) return false;
    const int wave_u4 = std::max(ns * 20 * T, 64 * T);
    return ((ssq ? DG_SSQ_BYTES / 16 : 0) + nwv * wave_u4) * 16 <= DG_LDS_BYTES - 512;
}
template <int T, int EPI>
inline hipError_t dg_launch_k(const QGemmParams &P, hipStream_t st) {
    const int nb = P.cols / 32, nwv = dg_waves(nb), ns = nb / nwv;
    if (nwv == 16) {
        switch (ns) {      // (nb > 64)
        case 6: return dg_launch<T, 16, 6, EPI>(P, st);
        case 8: return dg_launch<T, 16, 8, EPI>(P, st);
        }
    } else {
        switch (ns) {
        case 1: return dg_launch<T, 8, 1, EPI>(P, st);
        case 3: return dg_launch<T, 8, 3, EPI>(P, st);
        case 2: return dg_launch<T, 8, 2, EPI>(P, st);
        case 4: return dg_launch<T, 8, 4, EPI>(P, st);
        case 6: return dg_launch<T, 8, 6, EPI>(P, st);
        case 8: return dg_launch<T, 8, 8, EPI>(P, st);
        }
    }
    return hipErrorInvalidValue;
}
inline hipError_t dg_launch_rope(const QGemmParams &P, hipStream_t st) { return dg_launch_k<3, QG_EPI_ROPE>(P, st); }
inline hipError_t dg_launch_swiglu(const QGemmParams &P, hipStream_t st) { return dg_launch_k<8, QG_EPI_SWIGLU>(P, st); }
inline hipError_t dg_launch_plain(const QGemmParams &P, hipStream_t st) { return dg_launch_k<2, QG_EPI_PLAIN>(P, st); }
// ---- the LM head of a decode batch (go/model.go:616-619) ------------------------------------------------------------------------
// 2000 row tiles (vocabulary 32000) against 4 token tiles would be 1000-2000 workgroups of dgemm_kernel: four to eight rounds on 256
// compute units, every round re-reading its token tile's fragments (98 KB for K = 1536) and paying ~3 us between a workgroup's exit
// and its successor's entry (tools/dgemm_bench.hip: 41 us; the split-K launches it would replace: 45).  Here a workgroup STAYS: one
// per compute unit, its 16 tokens' fragments in registers for the whole launch (NS pairs per wavefront, loaded once), and it walks
// the row groups rg, rg + stride, ... of four row tiles each; the weights of the NEXT row group are requested (LDS-DMA into the other
// of two buffers) before the products of this one start, so HBM latency and streaming sit under the matrix work.  Same per-wavefront
// block ownership, operand layout and arithmetic as dgemm_kernel; the partial tiles overlay the buffer just consumed, one wavefront
// per row tile reduces them in wavefront order, scales by the folded final RMSNorm (QGemmParams::NormIn), stores the logits rows and
// one (max, index) candidate per token and row tile for the argmax launch behind (P.part1: [token][rows / 16] {value, index};
// go/main.go:400-408: strict '>', the earlier index wins; row 0 is taken whatever it holds, as the reference's loop starts from it).

// A workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every vector-memory operation of the wavefront
// (vmcnt(0)) -- here that would drain the next row group's LDS-DMA pieces at every barrier (32000 rows x 64 tokens: 36 us
// with it and with compiler-visible LDS reads, 30 us without)
__device__ __forceinline__ void dg_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// ... and LDS reads hipcc does not see: in front of a ds_read it emits, it waits for every LDS-DMA piece in flight (they might write
// what is read) -- the partial tiles and the sums of squares are read while the next row group streams in.  The caller waits
// (dg_lds_wait) before the first use.
__device__ __forceinline__ unsigned dg_lds_addr(const void *p) { return (unsigned)(size_t)(const __attribute__((address_space(3))) void *)p; }
__device__ __forceinline__ void dg_lds_read16(f32x4_t &v, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory"); }
__device__ __forceinline__ void dg_lds_read8(double &v, unsigned addr) { asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr) : "memory"); }
// (the values are operands of the wait: nothing that uses them may be scheduled in front of it)
__device__ __forceinline__ void dg_lds_wait(double &v) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v) :: "memory"); }
template <int N>
__device__ __forceinline__ void dg_lds_wait(f32x4_t (&v)[N]) {
    static_assert(N == 8 || N == 16, "partial tiles of 8 or 16 wavefronts");
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) :: "memory");
    if constexpr (N == 16)
        asm volatile("" : "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) :: "memory");
}
constexpr int DGH_T = 4;
template <int NWV, int NS> struct DgHeadLds {
    static constexpr int REG_U4 = 20 * DGH_T;
    static constexpr int WAVE_U4 = NS * REG_U4 > DGH_T * 64 ? NS * REG_U4 : DGH_T * 64;
    static constexpr int BUF_U4 = NWV * WAVE_U4;
    static constexpr int SSQ_U4 = DG_SSQ_BYTES / 16 + 16;          // + 256 bytes: the producer's pre-scale of the tile's 16 tokens
    static constexpr int TOTAL_U4 = SSQ_U4 + 2 * BUF_U4;
};
template <int NWV, int NS>
__global__ void __launch_bounds__(NWV * 64, NWV / 4) dghead_kernel(QGemmParams P) {
    typedef DgHeadLds<NWV, NS> L;
    constexpr int T = DGH_T, REG_U4 = L::REG_U4, WAVE_U4 = L::WAVE_U4, BUF_U4 = L::BUF_U4, SSQ_U4 = L::SSQ_U4;
    constexpr int OPS = 2;                         // memory operations per step: one nibble piece (64 lanes = 4 tiles x 16 rows), one scale piece
    static_assert((2 * NS - 1) * OPS <= 63, "vmcnt is six bits");
    static_assert(NWV >= T, "one epilogue wavefront per row tile");
    __shared__ __attribute__((aligned(16))) uint4 lds_all[L::TOTAL_U4];
    NL_KARGS8(P.q, P.s, P.xf, P.out, P.part1, P.nrm_in.ssq, P.nrm_in.scale, P.nrm_in.scale_next);
    NL_KARGS8(P.rows, P.cols, P.npairs, P.ntiles, P.nt16, P.n_tokens, P.ldo, P.nrm_in.nrb);
    NL_KARGS2(P.nrm_in.dim, P.nrm_in.eps);
    const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int li = lane & 15, lq = lane >> 4;
    // block -> (first row group, token tile) as dgemm_kernel: the token tiles of a row group on one XCD
    const int rg0 = ((int)blockIdx.x & 7) + 8 * (int)blockIdx.y, tt = (int)blockIdx.x >> 3, stride = 8 * (int)gridDim.y;
    [[maybe_unused]] constexpr int DG_THIS_EPI = 3;      // (stamps: the LM head)
    DG_STAMP(0);
    DG_LOG_ENTRY();
    const int nrg = (P.ntiles + T - 1) / T;
    if (rg0 >= nrg) return;
    const int niter = (nrg - rg0 + stride - 1) / stride;
    const int blk0 = wv, k0 = (blk0 >> 1) & 3, cc = blk0 & 1;
    const int n = tt * 16 + li, nn = min(n, P.n_tokens - 1);
    const bool live = n < P.n_tokens;
    if (P.nrm_in.ssq) {      // the producer's partial sums of squares of this tile's 16 tokens -> LDS (older than every weight piece)
        const unsigned bytes = 16u * (unsigned)P.nrm_in.nrb * 8u;
        const char *const src = reinterpret_cast<const char *>(P.nrm_in.ssq + (size_t)tt * 16 * P.nrm_in.nrb);
        for (unsigned p0 = (unsigned)wv * 1024u; p0 < bytes; p0 += NWV * 1024u)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + min(p0 + lane * 16u, bytes - 16u)),
                                             (__attribute__((address_space(3))) void *)(lds_all + p0 / 16), 16, 0, DG_AUX);
        // ... and their power-of-two pre-scales (lane -> token lane & 15; a register load would have to be waited for by hand at
        // a point the compiler may already have copied the register)
        if (wv == 0 && P.nrm_in.scale)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(P.nrm_in.scale + nn),
                                             (__attribute__((address_space(3))) void *)(lds_all + DG_SSQ_BYTES / 16), 4, 0, DG_AUX);
    }
    // this wavefront's NS fragment pairs of the token tile: registers, for the whole launch
    dg_u32x4 xh[NS], xl[NS];
    {
        const unsigned xblock = (unsigned)P.nt16 * (2 * QG_FRAG * 16);
        const char *xsrc = reinterpret_cast<const char *>(P.xf) + ((size_t)blk0 * xblock + ((size_t)(tt * 2) * QG_FRAG + lane) * 16);
#pragma unroll
        for (int s = 0; s < NS; s++) {
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(xh[s]) : "v"(xsrc) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:1024 sc1" : "=&v"(xl[s]) : "v"(xsrc) : "memory");
            xsrc += (size_t)NWV * xblock;
        }
    }
    // the NS steps of row group rg into buffer b: lane -> (tile lane >> 4, row lane & 15) of the group
    auto issue = [&](int rg, int b) {
        const int tile = min(rg * T + (lane >> 4), P.ntiles - 1), r = lane & 15;
        const size_t grp = (size_t)tile * P.npairs + (size_t)(blk0 >> 3) * KL;
        const char *wsrc = reinterpret_cast<const char *>(P.q) + (grp * (2 * TR) + (size_t)((k0 * 2 + cc) * TR + r)) * 16;
        const char *ssrc = reinterpret_cast<const char *>(P.s) + (grp * TR + (size_t)(k0 * TR + r)) * 4;
        uint4 *const ring = lds_all + SSQ_U4 + b * BUF_U4 + wv * WAVE_U4;
#pragma unroll
        for (int s = 0; s < NS; s++) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)wsrc,
                                             (__attribute__((address_space(3))) void *)(ring + s * REG_U4), 16, 0, DG_AUX);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)ssrc,
                                             (__attribute__((address_space(3))) void *)(ring + s * REG_U4 + 16 * T), 4, 0, DG_AUX);
            wsrc += KL * 2 * TR * 16 * (NWV / 8);
            ssrc += KL * TR * 4 * (NWV / 8);
        }
    };
    issue(rg0, 0);
    if (niter > 1) issue(rg0 + stride, 1);
    asm volatile("" ::: "memory");
    // the fragments have landed when only the weight pieces issued behind them are outstanding.  They are pinned HERE, once, in
    // straight-line code: a "+v" inside the loop's two product variants made hipcc copy the registers at the branch -- in front
    // of the wait (wrong products for whatever had not landed; seen with one row group per workgroup)
    if (niter > 1) DG_WAIT_VM(2 * NS * OPS); else DG_WAIT_VM(NS * OPS);
#pragma unroll
    for (int s = 0; s < NS; s++) asm volatile("" : "+v"(xh[s]), "+v"(xl[s]) :: "memory");
    float inv = 1.0f;
    for (int it = 0; it < niter; it++) {
        const int rg = rg0 + it * stride, b = it & 1;
        const bool has_next = it + 1 < niter;
        if (it > 0 && has_next) issue(rg + stride, b ^ 1);
        asm volatile("" ::: "memory");
        DG_STAMP(1 + 4 * it);
        uint4 *const ring = lds_all + SSQ_U4 + b * BUF_U4 + wv * WAVE_U4;
        f32x4_t acc[T];
#pragma unroll
        for (int ti = 0; ti < T; ti++) acc[ti] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        auto products = [&](auto next_) {
            constexpr int AHEAD = decltype(next_)::value ? NS * OPS : 0;      // the next row group's pieces, issued behind this one's
            dg_static_for<0, NS>([&](auto s_) {
                constexpr int s = decltype(s_)::value;
                DG_WAIT_VM((NS - 1 - s) * OPS + AHEAD);
                asm volatile("" ::: "memory");
                const uint32_t *const sb32 = reinterpret_cast<const uint32_t *>(ring + s * REG_U4);
                const half8_t xhv = __builtin_bit_cast(half8_t, xh[s]), xlv = __builtin_bit_cast(half8_t, xl[s]);
                half8_t a[T];
                uint32_t sw[T];
                f32x4_t z[T];
#pragma unroll
                for (int g = 0; g < T; g++) {
                    a[g] = WFrag<WT_Q4_0>::expand(sb32[(g * 16 + li) * 4 + lq]);
                    sw[g] = sb32[(16 * T) * 4 + g * 16 + li];
                }
    #pragma unroll
                for (int g = 0; g < T; g++) z[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xlv, a[g], (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    #pragma unroll
                for (int g = 0; g < T; g++) z[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xhv, a[g], z[g], 0, 0, 0);
    #pragma unroll
                for (int g = 0; g < T; g++) {
                    const float d = scale_of(sw[g], cc);
#pragma unroll
                    for (int r = 0; r < 4; r++) acc[g][r] = fmaf(z[g][r], d, acc[g][r]);
                }
            });
        };
        if (has_next) products(std::true_type{}); else products(std::false_type{});
        DG_STAMP(2 + 4 * it);
        // partial tiles over the buffer just consumed: D[token 4 lq + r][row li] out as [tile][token][row]
        {
            float *const mine = reinterpret_cast<float *>(ring);
#pragma unroll
            for (int ti = 0; ti < T; ti++)
#pragma unroll
                for (int r = 0; r < 4; r++) mine[ti * 256 + (4 * lq + r) * 16 + li] = acc[ti][r];
        }
        dg_lds_barrier();
        DG_STAMP(3 + 4 * it);
        if (it == 0 && P.nrm_in.ssq) {       // inv of this lane's token, as dgemm_kernel's consumer side (every wavefront: the epilogue rotates)
            const unsigned sq = dg_lds_addr(lds_all) + (unsigned)(li * P.nrm_in.nrb) * 8u;
            double tot = 0.0;
            for (int r = lq; r < P.nrm_in.nrb; r += 4) {
                double v;
                dg_lds_read8(v, sq + (unsigned)r * 8u);
                dg_lds_wait(v);
                tot += v;
            }
            const double t1 = __shfl_xor(tot, 16);
            const double lo2 = (lq & 1) ? t1 + tot : tot + t1;
            const double t2 = __shfl_xor(lo2, 32);
            tot = (lq & 2) ? t2 + lo2 : lo2 + t2;
            inv = dg_inv_rms(tot, P.nrm_in.dim, P.nrm_in.eps);
            if (P.nrm_in.scale) {
                float nsc;
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(nsc) : "v"(dg_lds_addr(lds_all + DG_SSQ_BYTES / 16) + (unsigned)li * 4u) : "memory");
                if (rg0 == 0 && wv == 0 && lq == 0 && live) P.nrm_in.scale_next[n] = norm_prescale(inv);
                inv *= 1.0f / nsc;
            }
        }
        // row tile ti of the group: wavefront (ti + it) % NWV (the duty moves so that no wavefront falls behind for the whole launch)
        const int ti = (wv - it) & (NWV - 1);
        const int etile = rg * T + ti;
        if (ti < T && etile < P.ntiles) {
            f32x4_t e = f32x4_t{0.f, 0.f, 0.f, 0.f}, pt[NWV];
            const unsigned pa = dg_lds_addr(lds_all + SSQ_U4 + b * BUF_U4) + (unsigned)(ti * 256 + li * 16 + lq * 4) * 4u;
#pragma unroll
            for (int w = 0; w < NWV; w++) dg_lds_read16(pt[w], pa + (unsigned)(w * WAVE_U4) * 16u);
            dg_lds_wait(pt);
#pragma unroll
            for (int w = 0; w < NWV; w++) e += pt[w];
            e = e * inv;
            if (live) {
                const int row0 = etile * TR + lq * 4;
                *reinterpret_cast<float4 *>(P.out + (size_t)n * P.ldo + row0) = make_float4(e[0], e[1], e[2], e[3]);
                float best = -INFINITY;
                int bidx = 0x7fffffff;
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (e[j] > best || row0 + j == 0) { best = e[j]; bidx = row0 + j; }
#pragma unroll
                for (int o = 16; o <= 32; o <<= 1) {       // (the four lanes of a token are live together)
                    const float ov = __shfl_xor(best, o);
                    const int oi = __shfl_xor(bidx, o);
                    if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
                }
                if (lq == 0) reinterpret_cast<uint2 *>(P.part1)[(size_t)n * P.ntiles + etile] = make_uint2(__float_as_uint(best), (unsigned)bidx);
            }
        }
        DG_STAMP(4 + 4 * it);
        if (has_next) dg_lds_barrier();      // (the row group after next lands in this buffer: requested behind this barrier)
    }
    DG_STAMP(40);
    DG_LOG_EXIT();
}
inline bool dg_head_ok(int rows, int cols) {
    if (cols <= 0 || cols % 256 || rows % 16) return false;
    const int nb = cols / 32, nwv = dg_waves(nb), ns = nb / nwv;
    if (nb % nwv || !(ns == 1 || ns == 2 || ns == 3 || ns == 4 || ns == 6)) return false;
    return (DG_SSQ_BYTES / 16 + 2 * nwv * std::max(ns * 20 * DGH_T, 64 * DGH_T)) * 16 <= DG_LDS_BYTES - 512;
}
template <int NWV, int NS>
inline hipError_t dg_launch_head_k(QGemmParams P, hipStream_t st, int num_cus) {
    P.nt16 = ((P.n_tokens + 63) / 64) * 4;
    P.ksplit = 1;
    if constexpr (DgHeadLds<NWV, NS>::TOTAL_U4 * 16 <= DG_LDS_BYTES - 512) {
        const int ntt = (P.n_tokens + 15) / 16, nrg = (P.ntiles + DGH_T - 1) / DGH_T;
        const int gy = std::max(1, std::min((nrg + 7) / 8, num_cus / (8 * ntt)));      // one workgroup per compute unit
        hipLaunchKernelGGL((dghead_kernel<NWV, NS>), dim3((unsigned)(8 * ntt), (unsigned)gy), dim3(NWV * 64), 0, st, P);
        return hipGetLastError();
    } else return hipErrorInvalidValue;
}
inline hipError_t dg_launch_head(const QGemmParams &P, hipStream_t st, int num_cus = 256) {
    const int nb = P.cols / 32, nwv = dg_waves(nb), ns = nb / nwv;
    if (nwv == 8) {
        switch (ns) {
        case 1: return dg_launch_head_k<8, 1>(P, st, num_cus);
        case 2: return dg_launch_head_k<8, 2>(P, st, num_cus);
        case 3: return dg_launch_head_k<8, 3>(P, st, num_cus);
        case 4: return dg_launch_head_k<8, 4>(P, st, num_cus);
        case 6: return dg_launch_head_k<8, 6>(P, st, num_cus);
        }
    }
    return hipErrorInvalidValue;
}

// argmax of a decode batch behind the LM-head launch: its candidates [token][nc] {value, index} -> ids; one workgroup per token,
// the merge of bargmax_kernel (nl_batch.h)
__global__ void __launch_bounds__(1024) dg_argmax_kernel(const uint2 *cand, int nc, int *ids) {
    __shared__ float bv[16];
    __shared__ int bi[16];
    const uint2 *c = cand + (size_t)blockIdx.x * nc;
    const int tid = threadIdx.x;
    float best = -INFINITY;
    int idx = 0x7fffffff;
    for (int i = tid; i < nc; i += (int)blockDim.x) {
        const uint2 v = c[i];
        const float f = __uint_as_float(v.x);
        if (f > best || idx == 0x7fffffff) { best = f; idx = (int)v.y; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o);
        const int oi = __shfl_xor(idx, o);
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
