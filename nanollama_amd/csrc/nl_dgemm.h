// nl_dgemm.h -- the multi-token GEMM for SHORT token runs (decode batches of <= 64 streams, go/serve.go's concurrent requests
// stepped together; Q4_0): one launch per projection, no split-K, no reduce launch, every epilogue of the layer fused.
//
// What bounds a GEMM of 64 tokens on this chip is neither HBM (a goldie layer is 14 MB: 2 us) nor the matrix cores, but
// what ONE compute unit can ingest: 64 B per clock from L2.  The activation matrix of 64 tokens is 4 bytes per element
// (fp16 hi + lo, nl_qgemm.h) -- 393 KB for K = 1536 -- so a workgroup that sees all 64 tokens and all of K needs 2.6 us just
// to read its B operand; qgemm_kernel therefore split K over workgroups (<= 6 chunks each), wrote partial slabs and left the
// sum, the norm, the rotation and SiLU to four more launches per layer: 8 launches of 5-9 us.
// Here the TOKEN TILE is what a workgroup keeps: 16 tokens x all of K (98 KB for K = 1536) against a few 16-row weight
// tiles, so a result element is finished inside one workgroup and the qgemm2 epilogues (RoPE + KV store, SiLU(gate) * up as
// the next GEMM's fragments, residual + the folded RMSNorm of the next GEMM) apply as they are.  The four workgroups that
// share a row group (one per token tile) are placed on the same XCD (block b runs on XCD b % 8, observed -- for speed only), so
// a weight byte crosses the fabric once and is an L2 hit for the other three.
//
//   * Everything a workgroup multiplies arrives by LDS-DMA (global_load_lds_dwordx4, 1 KB per wavefront instruction) into a
//     ring of chunk slots, a chunk = one 256-column group of the tile layout = 8 quant blocks: 16 KB of activation fragments,
//     2 KB of nibbles + 256 B of fp16 scales per weight tile.  The ring is as deep as the 160 KB of LDS allow (4-7 slots:
//     3-6 chunks in flight, more than one HBM round trip of ingest), every wavefront issues the same number of pieces per
//     chunk, so "chunk c has landed" is ONE s_waitcnt vmcnt(immediate) + s_barrier; no register staging, no weight registers.
//   * The nibble chunks of a (tile, group) are stored XOR-swizzled (the DMA picks its source per lane), so the A-operand
//     read -- dword lq of the 16-byte chunk of (row li, block) -- is 2-way instead of 8-way bank-conflicted.
//   * wavefront = RT row tiles x one token tile x 1/KS of every chunk's blocks; the KS partial accumulators meet in LDS in a
//     fixed order.  Per (block, tile): 1 ds_read_b32, 9 VALU (nibbles -> fp16), 2 MFMA (x = hi + lo), 4 cvt + 4 FMA ("* d").
// Arithmetic per output = qgemm_kernel / qgemm2_kernel (exact integer quants in fp16, f32 block sums, * d in f32); the
// summation order over blocks differs with KS > 1 (block-interleaved partial sums), inside the stated tolerances.
// Reference: go/quant.go:45-94 (MatMulQ4_0), go/model.go:513-613 (the layer), per stream.
#pragma once
#include "nl_qgemm2.h"

namespace nl {

constexpr int DG_KB = 8;                        // quant blocks per chunk (one KL-pair group of the tile layout)
constexpr int DG_ACT_U4 = DG_KB * 2 * QG_FRAG;  // uint4 of activation fragments per chunk: [block][hi/lo][lane]
constexpr int DG_LDS_BYTES = 160 * 1024;

constexpr int DG_SSQ_MAX_NRB = 64;              // partial sums of squares per token the consumer side takes (dim <= 2048)
constexpr int DG_SSQ_BYTES = 16 * DG_SSQ_MAX_NRB * 8;
template <int T> struct DgLds {                 // T = (weight tile, matrix) entries of a workgroup
    static constexpr int WQ_U4 = T * 128, WS_PIECES = (T + 3) / 4, WS_U4 = WS_PIECES * 64;
    static constexpr int SLOT_U4 = DG_ACT_U4 + WQ_U4 + WS_U4;
    static constexpr int NSLOT_FIT = (DG_LDS_BYTES - DG_SSQ_BYTES - 1024) / (SLOT_U4 * 16);
    static constexpr int NSLOT = NSLOT_FIT > 8 ? 8 : NSLOT_FIT;
};

#ifdef DG_STAMPS
__device__ long long g_dg_stamps[64];
#define DG_STAMP(i) do { if (blockIdx.x == 9 && threadIdx.x == 0 && (i) < 64) g_dg_stamps[(i)] = clock64(); } while (0)
#else
#define DG_STAMP(i) do { } while (0)
#endif

// vmcnt immediate of s_waitcnt on gfx9 (vmcnt[3:0] | expcnt[6:4] = 7 | lgkmcnt[11:8] = 15 | vmcnt[5:4] in [15:14])
#define DG_WAIT_VM(n) __builtin_amdgcn_s_waitcnt(0x0f70 | ((n) & 15) | (((n) >> 4) << 14))

// RT: row tiles per wavefront (SWIGLU: 4 = two gate tiles + the same two tiles of up); NW: wavefronts over rows; KS: wavefronts
// over the blocks of a chunk.  grid.x = ceil(row groups / 8) * 8 * token tiles, see dg_block_of().
template <int RT, int NW, int KS, int EPI>
__global__ void __launch_bounds__(NW * KS * 64, (NW * KS + 3) / 4) dgemm_kernel(QGemmParams P) {
    constexpr int NWV = NW * KS, T = NW * RT, BPW = DG_KB / KS;
    typedef DgLds<T> L;
    constexpr int NSLOT = L::NSLOT, PD = NSLOT - 1;
    // DMA pieces of a chunk: 16 of activations, 2 T of nibbles, WS_PIECES of scales.  Every wavefront issues the same number per
    // chunk (so one vmcnt immediate serves all); with more wavefronts than pieces of a kind, several fetch the same piece
    constexpr int ACT_PPW = 16 / NWV > 0 ? 16 / NWV : 1, WQ_PPW = 2 * T / NWV > 0 ? 2 * T / NWV : 1, PPW = ACT_PPW + WQ_PPW + 1;
    static_assert(NWV <= 16 && (16 % NWV == 0) && ((2 * T) % NWV == 0 || NWV % (2 * T) == 0), "pieces divide evenly over the wavefronts");
    static_assert((PD - 1) * PPW <= 63, "vmcnt is six bits");
    static_assert(EPI != QG_EPI_SWIGLU || RT == 4, "gate tile pair + up tile pair per wavefront");
    static_assert(EPI != QG_EPI_PLAIN || RT == 2, "a producer wavefront owns one 32-row block of the next GEMM's K");
    static_assert(NWV * RT * 64 <= NSLOT * L::SLOT_U4, "the K-split partials fit the ring");
    __shared__ __attribute__((aligned(16))) uint4 lds_all[DG_SSQ_BYTES / 16 + NSLOT * L::SLOT_U4];
    uint4 *const lds = lds_all + DG_SSQ_BYTES / 16;
    DG_STAMP(0);

    const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int li = lane & 15, lq = lane >> 4;
    const int wr = wv % NW, wk = wv / NW;
    // block -> (row group, token tile): the token tiles of one row group sit on the same XCD (b % 8), next to each other in
    // dispatch order
    const int ntt = (P.n_tokens + 15) >> 4;
    const int b = blockIdx.x, rg = (b & 7) + 8 * (b / (8 * ntt)), tt = (b >> 3) % ntt;
    constexpr bool FUSED = EPI == QG_EPI_SWIGLU;
    constexpr int TPW = FUSED ? 2 : RT;                      // distinct row tiles of a wavefront
    if (rg * NW * TPW >= P.ntiles) return;                   // (padding of the row groups to a multiple of 8: the whole workgroup)
    const int nchunks = P.cols / (32 * DG_KB);
    const int tile0 = (rg * NW + wr) * TPW;                  // this wavefront's first row tile (FUSED: of gate and of up)

    // ---- DMA pieces of this wavefront: per-lane source pointers, advanced chunk by chunk ----
    const char *asrc[ACT_PPW];
    const unsigned xblock = (unsigned)P.nt16 * (2 * QG_FRAG * 16);    // bytes of one block's fragments
#pragma unroll
    for (int i = 0; i < ACT_PPW; i++) {
        const int q = (i * NWV + wv) % 16, bi = q >> 1, part = q & 1;
        asrc[i] = reinterpret_cast<const char *>(P.xf) + ((size_t)bi * xblock + ((size_t)(tt * 2 + part) * QG_FRAG + lane) * 16);
    }
    auto tile_of = [&](int ti, int &mat) {          // entry ti of the workgroup's weight area -> (row tile, matrix)
        const int w = ti / RT, rt = ti % RT;
        mat = FUSED ? rt >> 1 : 0;
        return min((rg * NW + w) * TPW + (FUSED ? rt & 1 : rt), P.ntiles - 1);
    };
    const char *wsrc[WQ_PPW];
#pragma unroll
    for (int i = 0; i < WQ_PPW; i++) {
        const int q = (i * NWV + wv) % (2 * T), ti = q >> 1, m = q & 1;
        int mat;
        const int tile = tile_of(ti, mat);
        const int p = m * 64 + lane, u = p ^ ((p >> 3) & 3);        // LDS position p holds chunk u of the (tile, group)
        wsrc[i] = reinterpret_cast<const char *>(mat ? P.q1 : P.q) + ((size_t)tile * P.npairs * 32 + u) * 16;
    }
    const char *ssrc;
    {
        const int piece = wv % L::WS_PIECES, ti = min(piece * 4 + (lane >> 4), T - 1);
        int mat;
        const int tile = tile_of(ti, mat);
        ssrc = reinterpret_cast<const char *>(mat ? P.s1 : P.s) + ((size_t)tile * P.npairs * 16 * 4 + (size_t)(lane & 15) * 16);
    }
    auto issue = [&](int slot, bool adv) {     // adv: the pointers move on to the next chunk
        uint4 *const base = lds + slot * L::SLOT_U4;
#pragma unroll
        for (int i = 0; i < ACT_PPW; i++) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)asrc[i],
                                             (__attribute__((address_space(3))) void *)(base + ((i * NWV + wv) % 16) * QG_FRAG), 16, 0, 0);
            asrc[i] += adv ? (size_t)DG_KB * xblock : (size_t)0;
        }
#pragma unroll
        for (int i = 0; i < WQ_PPW; i++) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)wsrc[i],
                                             (__attribute__((address_space(3))) void *)(base + DG_ACT_U4 + ((i * NWV + wv) % (2 * T)) * 64), 16, 0, 0);
            wsrc[i] += adv ? KL * 2 * TR * 16 : 0;       // the next group of the tile
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)ssrc,
                                         (__attribute__((address_space(3))) void *)(base + DG_ACT_U4 + L::WQ_U4 + (wv % L::WS_PIECES) * 64), 16, 0, 0);
        ssrc += adv ? KL * TR * 4 : 0;
    };
    // ---- what the norm and the epilogue need from memory is requested FIRST and first USED after the K loop.  hipcc waits
    //      vmcnt(0) for a register load once LDS-DMA pieces are in flight behind it (it treats the two as unordered), i.e. any
    //      use inside the loop would drain the ring; the values are pinned behind the loop below.  The RoPE position of the
    //      lane's token comes through the scalar cache (lgkmcnt), so cos / sin can be requested before the first piece. ----
    const int n = tt * 16 + li, nn = min(n, P.n_tokens - 1);   // this lane's token; its rows are 4 * lq .. + 3 of every tile
    const bool live = n < P.n_tokens;
    [[maybe_unused]] float nsc = 1.0f;
    [[maybe_unused]] int pos = 0, strm = 0;
    [[maybe_unused]] float4 rv[RT], bv[RT], gw[RT], rc4[RT], rs4[RT];
    [[maybe_unused]] float psc = 1.0f;
    if constexpr (EPI != QG_EPI_PLAIN) {
        {   // (no load behind a branch: an absent operand reads the first bytes of the weight matrix, value unused)
            const bool on = P.nrm_in.ssq != nullptr;
            nsc = *(on && P.nrm_in.scale ? P.nrm_in.scale + nn : reinterpret_cast<const float *>(P.q));
        }
    }
    if constexpr (EPI == QG_EPI_ROPE) {
        const int nlast = P.n_tokens - 1 - tt * 16;        // (>= 0: the grid has no empty token tile)
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int pk = sload_i32(P.rope.pos + tt * 16 + min(k, nlast)), sk = sload_i32(P.rope.stream + tt * 16 + min(k, nlast));
            pos = li == k ? pk : pos;
            strm = li == k ? sk : strm;
        }
        const int hd = P.rope.head_dim, half = hd >> 1, tph = hd / 16;
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            const int tile = min(tile0 + rt, P.ntiles - 1), i0 = (tile % tph) * 8 + 4 * (lq & 1);
            rc4[rt] = *reinterpret_cast<const float4 *>(P.rope.cos + pos * half + i0);
            rs4[rt] = *reinterpret_cast<const float4 *>(P.rope.sin + pos * half + i0);
        }
    }
    if constexpr (EPI == QG_EPI_PLAIN) {
        const int row0 = min(tile0, P.ntiles - RT) * TR;
        const size_t off0 = (size_t)nn * P.ldo + (row0 + lq * 4);
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            // (an absent operand reads the first bytes of the weight matrix instead -- value unused -- so that no load sits behind a
            //  branch: hipcc waits for a conditional load inside its branch, one round trip each before the first DMA piece)
            const float *const dummy = reinterpret_cast<const float *>(P.q);
            rv[rt] = *reinterpret_cast<const float4 *>(P.resid ? P.resid + off0 + rt * TR : dummy);
            bv[rt] = *reinterpret_cast<const float4 *>(P.bias ? P.bias + row0 + rt * TR + lq * 4 : dummy);
            gw[rt] = *reinterpret_cast<const float4 *>(P.nrm_out.w ? P.nrm_out.w + row0 + rt * TR + lq * 4 : dummy);
        }
        psc = *(P.nrm_out.w && P.nrm_out.scale ? P.nrm_out.scale + nn : reinterpret_cast<const float *>(P.q));
    }
    asm volatile("" ::: "memory");
    if constexpr (EPI != QG_EPI_PLAIN) {
        // the producer's partial sums of squares of this tile's 16 tokens ([token][nrb] float64, contiguous) into LDS, older than
        // every chunk piece; read behind the loop
        if (P.nrm_in.ssq) {
            const unsigned bytes = 16u * (unsigned)P.nrm_in.nrb * 8u;
            const char *const src = reinterpret_cast<const char *>(P.nrm_in.ssq + (size_t)tt * 16 * P.nrm_in.nrb);
            for (unsigned p0 = (unsigned)wv * 1024u; p0 < bytes; p0 += NWV * 1024u)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + min(p0 + lane * 16u, bytes - 16u)),
                                                 (__attribute__((address_space(3))) void *)(lds_all + p0 / 16), 16, 0, 0);
        }
    }
    // PD chunks, unconditionally (a K shorter than the ring fetches its last chunk again into slots nobody reads)
#pragma unroll
    for (int c = 0; c < PD; c++) issue(c, c + 1 < nchunks);
    asm volatile("" ::: "memory");
    DG_STAMP(1);

    f32x4_t acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; rt++) acc[rt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // per-lane parts of the LDS addresses (uint32 units): nibble dword of (row li, pair k) and the four scale words of a pair
    const int swz = (li >> 1) & 3;
    int slot = 0, issued = PD;      // chunks requested so far
    for (int c = 0; c < nchunks; c++) {
        // chunk c has landed when at most the pieces of the chunks issued after it are outstanding
        const int after = issued - 1 - c;
        if (after == PD - 1) DG_WAIT_VM((PD - 1) * PPW);
        else if (PD > 2 && after == PD - 2) DG_WAIT_VM((PD > 2 ? PD - 2 : 0) * PPW);
        else if (PD > 3 && after == PD - 3) DG_WAIT_VM((PD > 3 ? PD - 3 : 0) * PPW);
        else if (PD > 4 && after == PD - 4) DG_WAIT_VM((PD > 4 ? PD - 4 : 0) * PPW);
        else if (PD > 5 && after == PD - 5) DG_WAIT_VM((PD > 5 ? PD - 5 : 0) * PPW);
        else if (PD > 6 && after == PD - 6) DG_WAIT_VM((PD > 6 ? PD - 6 : 0) * PPW);
        else DG_WAIT_VM(0);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();        // every wavefront's pieces of chunk c; and everybody has left chunk c - 1's slot
        asm volatile("" ::: "memory");
        DG_STAMP(2 + 2 * c);
        if (c + PD < nchunks) { issue(slot == 0 ? NSLOT - 1 : slot - 1, true); issued++; }
        const uint4 *const sb = lds + slot * L::SLOT_U4;
        const uint32_t *const wq32 = reinterpret_cast<const uint32_t *>(sb + DG_ACT_U4);
        const uint32_t *const ws32 = reinterpret_cast<const uint32_t *>(sb + DG_ACT_U4 + L::WQ_U4);
        // the fp16 scales of this wavefront's pairs of the chunk: rows 4*lq .. +3 of every tile, NPW pairs each (one LDS read per row)
        constexpr int NPW = BPW >= 2 ? BPW / 2 : 1;
        const int k0 = BPW >= 2 ? wk * NPW : wk >> 1;
        uint32_t sw[RT][4][NPW];
#pragma unroll
        for (int rt = 0; rt < RT; rt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t *src = ws32 + (wr * RT + rt) * 64 + (4 * lq + r) * 4 + k0;
                if constexpr (NPW == 4) { const uint4 v = *reinterpret_cast<const uint4 *>(src); sw[rt][r][0] = v.x; sw[rt][r][1] = v.y; sw[rt][r][2] = v.z; sw[rt][r][3] = v.w; }
                else if constexpr (NPW == 2) { const uint2 v = *reinterpret_cast<const uint2 *>(src); sw[rt][r][0] = v.x; sw[rt][r][1] = v.y; }
                else sw[rt][r][0] = *src;
            }
        // JB blocks at a time: JB * RT independent MFMA chains (lo product, then hi product on the same accumulator)
        constexpr int JB = BPW < (4 / RT > 0 ? 4 / RT : 1) ? BPW : (4 / RT > 0 ? 4 / RT : 1);
#pragma unroll
        for (int j0 = 0; j0 < BPW; j0 += JB) {
            half8_t xh[JB], xl[JB], a[JB][RT];
            f32x4_t z[JB][RT];
#pragma unroll
            for (int jb = 0; jb < JB; jb++) {
                const int j = j0 + jb;
                const int kk = k0 + (BPW >= 2 ? j >> 1 : 0), cc = BPW >= 2 ? j & 1 : wk & 1;      // pair of the group, block of the pair
                const int bi = 2 * kk + cc;
                xh[jb] = __builtin_bit_cast(half8_t, sb[(bi * 2 + 0) * QG_FRAG + lane]);
                xl[jb] = __builtin_bit_cast(half8_t, sb[(bi * 2 + 1) * QG_FRAG + lane]);
#pragma unroll
                for (int rt = 0; rt < RT; rt++)
                    a[jb][rt] = WFrag<WT_Q4_0>::expand(wq32[((wr * RT + rt) * 128 + cc * 64 + li * 4 + (kk ^ swz)) * 4 + lq]);
            }
#pragma unroll
            for (int jb = 0; jb < JB; jb++)
#pragma unroll
                for (int rt = 0; rt < RT; rt++)
                    z[jb][rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[jb][rt], xl[jb], (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
            for (int jb = 0; jb < JB; jb++)
#pragma unroll
                for (int rt = 0; rt < RT; rt++)
                    z[jb][rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[jb][rt], xh[jb], z[jb][rt], 0, 0, 0);
#pragma unroll
            for (int jb = 0; jb < JB; jb++) {
                const int j = j0 + jb, cc = BPW >= 2 ? j & 1 : wk & 1;
#pragma unroll
                for (int rt = 0; rt < RT; rt++)
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        acc[rt][r] = fmaf(z[jb][rt][r], scale_of(sw[rt][r][BPW >= 2 ? j >> 1 : 0], cc), acc[rt][r]);
            }
        }
        DG_STAMP(3 + 2 * c);
        slot = slot + 1 == NSLOT ? 0 : slot + 1;
    }
    // ---- the KS partial accumulators of a row part meet in wavefront wk = 0, ascending wk ----
    if constexpr (KS > 1) {
        __builtin_amdgcn_s_barrier();        // the last chunk's slot is read out
        f32x4_t *const red = reinterpret_cast<f32x4_t *>(lds);
        if (wk > 0) {
#pragma unroll
            for (int rt = 0; rt < RT; rt++) red[(wv * RT + rt) * 64 + lane] = acc[rt];
        }
        __syncthreads();
        if (wk > 0) return;
#pragma unroll
        for (int k2 = 1; k2 < KS; k2++)
#pragma unroll
            for (int rt = 0; rt < RT; rt++) acc[rt] += red[((k2 * NW + wr) * RT + rt) * 64 + lane];
    }
    // ---- folded RMSNorm, consumer side (QGemmParams::NormIn): inv of this lane's token from the producer's per-32-row sums of
    //      squares: lane lq adds partials lq, lq + 4, ... in ascending order, the four lanes of a token meet in lq order ----
    if constexpr (EPI != QG_EPI_PLAIN) {
        float inv = 1.0f;
        if (P.nrm_in.ssq) {
            asm volatile("" : "+v"(nsc));          // (first use HERE, behind the loop)
            const double *const sq = reinterpret_cast<const double *>(lds_all) + li * P.nrm_in.nrb;
            double tot = 0.0;
            for (int r = lq; r < P.nrm_in.nrb; r += 4) tot += sq[r];
            const double t1 = __shfl_xor(tot, 16);
            const double lo2 = (lq & 1) ? t1 + tot : tot + t1;          // (lq 0 + lq 1) resp. (lq 2 + lq 3), same order in both lanes
            const double t2 = __shfl_xor(lo2, 32);
            tot = (lq & 2) ? t2 + lo2 : lo2 + t2;
            inv = (float)(1.0 / sqrt(tot / (double)P.nrm_in.dim + (double)P.nrm_in.eps));
            if (P.nrm_in.scale) {     // the producer's power-of-two pre-scale (norm_prescale, nl_qgemm.h): undone exactly
                if (rg == 0 && wv == 0 && lq == 0 && live) P.nrm_in.scale_next[n] = norm_prescale(inv);
                inv *= 1.0f / nsc;
            }
        }
#pragma unroll
        for (int rt = 0; rt < RT; rt++) acc[rt] = acc[rt] * inv;
    }
    if constexpr (EPI == QG_EPI_SWIGLU) {
        // h = SiLU(gate) * up (go/quant.go:629-631, go/model.go:604-606): rows 4*lq..+3 of both 16-row tiles of one token are
        // the two float4 groups of k-slot group lq of the wavefront's 32-row block of h (Q4_0 consumer: slot_offsets)
        const int hblk = tile0 >> 1;
        if (hblk * 32 >= P.rows || !live) return;
        float hv[2][4];
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float gv = acc[r][j], uv = acc[r + 2][j];
                const float ex = exp_f64_as_f32(-gv);
                hv[r][j] = (gv / (1.0f + ex)) * uv;
            }
        float v[8];
        slots_from(1, make_float4(hv[0][0], hv[0][1], hv[0][2], hv[0][3]), make_float4(hv[1][0], hv[1][1], hv[1][2], hv[1][3]), v);
        store_frag(P.xf_out, P.nt16, n, hblk, lq, v);
        return;
    }
    if constexpr (EPI == QG_EPI_ROPE) {
        // RoPE (go/model.go:449-477) + attention biases (:525-527) + KV store (:552-554), as qgemm2_kernel's epilogue: the
        // rotation partner of rows 4*lq + j is rows 4*(lq ^ 2) + j of the same tile and token: lane ^ 32
        const QGemmParams::Rope &R = P.rope;
        const int hd = R.head_dim, half = hd >> 1, tph = hd / 16, nq = R.n_q_heads * hd;
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            const int tile = min(tile0 + rt, P.ntiles - 1);
            const int head = tile / tph, i0 = (tile % tph) * 8 + 4 * (lq & 1), e0 = i0 + (lq >> 1) * half;
            const bool is_q = head < R.n_q_heads, is_k = !is_q && head < R.n_q_heads + R.n_kv_heads;
            const int kvh = head - R.n_q_heads - (is_k ? 0 : R.n_kv_heads);
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (R.bias_q) bv = *reinterpret_cast<const float4 *>((is_q ? R.bias_q + head * hd : is_k ? R.bias_k + kvh * hd : R.bias_v + kvh * hd) + e0);
            const float bj[4] = {bv.x, bv.y, bv.z, bv.w};
            float4 c4 = rc4[rt], s4 = rs4[rt];
            asm volatile("" : "+v"(c4.x), "+v"(c4.y), "+v"(c4.z), "+v"(c4.w));
            asm volatile("" : "+v"(s4.x), "+v"(s4.y), "+v"(s4.z), "+v"(s4.w));
            const float cj[4] = {c4.x, c4.y, c4.z, c4.w}, sj[4] = {s4.x, s4.y, s4.z, s4.w};
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float v = acc[rt][j] + bj[j];
                const float partner = __shfl_xor(v, 32);
                float outv = v;
                if (is_q || is_k) {
                    const float x0 = lq < 2 ? v : partner, x1 = lq < 2 ? partner : v;
                    if (!R.conj) outv = lq < 2 ? (x0 * cj[j] - x1 * sj[j]) : (x0 * sj[j] + x1 * cj[j]);
                    else outv = lq < 2 ? (x0 * cj[j] + x1 * sj[j]) : (-x0 * sj[j] + x1 * cj[j]);
                }
                o[j] = outv;
            }
            if (!live || tile0 + rt >= P.ntiles) continue;
            float *dstp = is_q ? R.q + ((long long)n * nq + head * hd + e0)
                               : (is_k ? R.kcache : R.vcache) + ((long long)strm * R.kv_stream_stride + ((long long)kvh * R.seq_len + pos) * hd + e0);
            *reinterpret_cast<float4 *>(dstp) = make_float4(o[0], o[1], o[2], o[3]);
        }
        return;
    }
    if constexpr (EPI == QG_EPI_PLAIN) {
        // out = resid + bias + y; then (QGemmParams::NormOut) the rows as the NEXT GEMM's fragments, times its norm weights and
        // the token's power-of-two pre-scale, plus the float64 sum of squares of this wavefront's 32 rows (qgemm2_kernel's
        // producer epilogue with one partial per 32-row block)
        const int row0 = tile0 * TR;
        if (tile0 + RT > P.ntiles) return;        // (row counts are multiples of 32 on this path: never taken)
        const size_t off0 = (size_t)nn * P.ldo + (row0 + lq * 4);
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            asm volatile("" : "+v"(rv[rt].x), "+v"(rv[rt].y), "+v"(rv[rt].z), "+v"(rv[rt].w));      // (first use behind the loop)
            float4 v = make_float4(acc[rt][0], acc[rt][1], acc[rt][2], acc[rt][3]);
            if (P.bias) { v.x += bv[rt].x; v.y += bv[rt].y; v.z += bv[rt].z; v.w += bv[rt].w; }
            if (P.resid) { v.x += rv[rt].x; v.y += rv[rt].y; v.z += rv[rt].z; v.w += rv[rt].w; }
            if (live) *reinterpret_cast<float4 *>(P.out + off0 + rt * TR) = v;
            acc[rt] = (f32x4_t){v.x, v.y, v.z, v.w};
        }
        if (P.nrm_out.w) {
            const float ps = P.nrm_out.scale ? psc : 1.0f;
            double ss = 0.0;
#pragma unroll
            for (int rt = 0; rt < RT; rt++)
#pragma unroll
                for (int j = 0; j < 4; j++) ss = fma((double)acc[rt][j], (double)acc[rt][j], ss);
            ss += __shfl_xor(ss, 16);
            ss += __shfl_xor(ss, 32);
            if (lq == 0 && live) P.nrm_out.ssq[(size_t)n * (P.rows / 32) + (tile0 >> 1)] = ss;
            float y[2][4];
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const float4 g = gw[r];
                y[r][0] = (acc[r][0] * g.x) * ps; y[r][1] = (acc[r][1] * g.y) * ps;
                y[r][2] = (acc[r][2] * g.z) * ps; y[r][3] = (acc[r][3] * g.w) * ps;
            }
            float v[8];
            slots_from(1, make_float4(y[0][0], y[0][1], y[0][2], y[0][3]), make_float4(y[1][0], y[1][1], y[1][2], y[1][3]), v);
            if (live) store_frag(P.nrm_out.xf, P.nt16, n, tile0 >> 1, lq, v);
        }
    }
}

// grid.x of a dgemm launch: row groups padded to a multiple of 8 (the block -> XCD map), times the token tiles
inline unsigned dg_grid(int ntiles, int tiles_per_wg, int n_tokens) {
    const int nrg = (ntiles + tiles_per_wg - 1) / tiles_per_wg;
    return (unsigned)(((nrg + 7) / 8) * 8 * ((n_tokens + 15) / 16));
}

// the three launches of a layer (geometry: DESIGN.md 3.5 / tools/dgemm_bench.hip)
#ifndef DG_ROPE_GEOM
#define DG_ROPE_GEOM 1, 4, 4
#endif
#ifndef DG_SWIGLU_GEOM
#define DG_SWIGLU_GEOM 4, 2, 8
#endif
#ifndef DG_PLAIN_GEOM
#define DG_PLAIN_GEOM 2, 1, 8
#endif
template <int RT, int NW, int KS, int EPI>
inline hipError_t dg_launch(QGemmParams P, hipStream_t st) {
    P.nt16 = ((P.n_tokens + 63) / 64) * 4;
    P.ksplit = 1;
    constexpr int TILES_PER_WG = NW * (EPI == QG_EPI_SWIGLU ? 2 : RT);
    hipLaunchKernelGGL((dgemm_kernel<RT, NW, KS, EPI>), dim3(dg_grid(P.ntiles, TILES_PER_WG, P.n_tokens)), dim3(NW * KS * 64), 0, st, P);
    return hipGetLastError();
}
#define DG_COMMA_EPI(geom, epi) geom, epi
inline hipError_t dg_launch_rope(const QGemmParams &P, hipStream_t st) { return dg_launch<DG_ROPE_GEOM, QG_EPI_ROPE>(P, st); }
inline hipError_t dg_launch_swiglu(const QGemmParams &P, hipStream_t st) { return dg_launch<DG_SWIGLU_GEOM, QG_EPI_SWIGLU>(P, st); }
inline hipError_t dg_launch_plain(const QGemmParams &P, hipStream_t st) { return dg_launch<DG_PLAIN_GEOM, QG_EPI_PLAIN>(P, st); }

}  // namespace nl
