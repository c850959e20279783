// nl_persist.h -- weight-stationary, XCD-pipelined persistent greedy decode for the smallest tier (round 5).
//
// go/main.go:173-219 decodes token after token through go/model.go:490-620.  For nano (75 MB of Q8_0) the HBM time of a
// token is 9 us, yet the launch plans of nl_block.h need 180: thirty dependent launches, each re-fetching its weights cold
// and each meeting its peers through the fabric.  tools/xcd_exchange_probe.hip (profiles/r05_xcd_exchange_probe.log) measured
// what decides the structure here: a hand-off between two compute units of the SAME XCD through that XCD's own L2 (plain
// store, L1-bypassing load) is 0.25 us, against 0.58 us for the write-through form every cross-XCD exchange needs; an
// all-gather among the 32 compute units of one XCD 1.27 us against 2.8 us chip-wide.
//
// So ONE launch decodes n tokens, and every layer lives on ONE XCD:
//   * 256 persistent workgroups (one per compute unit) read HW_REG_XCC_ID and take a ticket from their XCD's counter; roles
//     follow the REAL placement (nothing assumes block b -> XCD b % 8; a census that is not 8 x 32 makes the launch give up
//     and the caller keeps the launch plans);
//   * XCD x owns layers first(x) .. first(x) + nslots(x) - 1 (13 layers: two on XCDs 0-4, one on 5-7).  The int8 quants of its
//     layers sit in the REGISTERS of its 32 compute units for the whole launch (8 wavefronts x 144 of 256 VGPRs: 4.2 MB per
//     layer), their fp16 scales and the LM head's 1/256 slice (85 KB) in each compute unit's LDS.  Nothing but activations and
//     K / V rows moves per token;
//   * inside a layer EVERY matrix is split by rows over all 32 units of the XCD (unit idx: rows idx * R / 32 ...): Q | K | V
//     (go/model.go:517-523), WO + residual (:590-594), gate / up + SiLU (:597-606), down + residual (:609-612).  H of the units
//     additionally run the attention of one head each (RoPE, KV store, softmax over the cache: :449-477, :552-587) on the
//     q | k | v rows that reach them through one more hand-off.  Five hand-offs per layer, all inside the XCD: q|k|v -> heads,
//     o -> all, x' -> all, h -> all, and the layer's output as the next layer's x (write-through only where the next layer
//     lives on another XCD);
//   * after the last layer every compute unit multiplies its LM-head rows, the per-unit (max, index) pairs meet on layer 0's
//     XCD, whose compute units pick the token (go/main.go:400-408: strict '>', lowest index on ties), look up its embedding
//     row (go/model.go:389-446) and start the next token.
//   * from position 128 on a head's 128-position attention passes are shared by up to three units (the owner and two of the
//     units that are not heads in that slot); the owner merges their (max, sum, sum p v) records: one more in-XCD hand-off.
// Dot products run on the matrix pipe: v_mfma_i32_4x4x4_16b_i8 over the int8 weights and four signed base-256 digits of the
// 2^-30-rounded inputs (pd_limbs / pd_units_impl below).
// Hand-offs are 8-byte {tag, value} granules (tag = launch base + step + 1; one aligned store each, the value is its own
// arrival flag), polled with L1-bypassing loads; every poll is bounded and a give-up sets the status word the host checks
// (it then redoes the chunk on the launch plans and retires this path, like the fused plans of nl_block.h).
//
// RESIDENT SESSION (per-call Forward).  go/main.go:173-219 calls Forward once per token and samples on the host, so the
// reference's own loop cannot use a chunked launch.  PdParams::session keeps the launch on the chip between nl_forward calls
// instead: after a step the doorman (unit 0 of layer 0's XCD) stores {step, argmax} into a pinned host word and polls a pinned
// mailbox word for the caller's next token; the LM-head units store every step's logits straight into the pinned host buffer
// (nl_host_logits).  A quit in the mailbox -- or idle_ticks without a command -- raises a status bit every poll of the launch
// watches, and the launch drains; the host starts another one when the next call comes (nl_engine.hip pd_session_step).
//
// Arithmetic per block is exact in int32 up to the 2^-30 input rounding, block products of a row are added in block order as
// go/quant.go:149-165 does; RMSNorm's 1/rms multiplies the row sum (as every GEMV here does); softmax pieces as nl_block.h.
// Held to the oracle's logits within 1e-4 and to its greedy ids (tests/test_gpu_persist.py).
#pragma once
#include "nl_kernels.h"

namespace nl {

typedef unsigned long long pd_u64;
typedef unsigned pd_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned pd_u32x4 __attribute__((ext_vector_type(4)));

constexpr int PD_THREADS = 512, PD_WAVES = 8, PD_XCDS = 8, PD_CUS = 32, PD_GRID = PD_XCDS * PD_CUS;
constexpr int PD_UQ = 2, PD_UW = 1, PD_UG = 4, PD_UD = 2;   // unit slots per lane: this unit's rows of Q | K | V, of WO, of gate + up, of down
constexpr int PD_UNITS = PD_UQ + PD_UW + PD_UG + PD_UD;   // 9: the registers hold 9 units of 32 int8 per layer slot (their fp16 scales sit in LDS)
constexpr int PD_ULM = 5;     // LM-head unit slots per lane (LDS)
constexpr int PD_SLOTS = 2;   // layers per XCD
constexpr int PD_MAXL = PD_SLOTS * PD_XCDS;
constexpr int PD_MAXD = 576, PD_MAXI = 1536, PD_NBD_MAX = PD_MAXD / 32, PD_NBI_MAX = PD_MAXI / 32;
constexpr int PD_PART = 128 * (PD_NBD_MAX + 1) + 128;  // floats: block products of the largest phase (the LM head's 128 rows), rows padded to an odd pitch
constexpr int PD_PARTS = 3;                            // units that share a head's attention passes (its owner + two of the XCD's units that are not heads)
constexpr int PD_MAX_POS = 1024;                       // eight attention passes of 128 positions: at most three per unit
constexpr int PD_MAX_PASSES = 8;                       // chunk records in LDS: a unit's own passes + the partials it merges (<= 3 + 2)

// ---- who holds what (host packer and kernel share these) -----------------------------------------------------------------
__host__ __device__ inline int pd_nslots(int L, int xcd) { return L / PD_XCDS + (xcd < L % PD_XCDS ? 1 : 0); }
__host__ __device__ inline int pd_first(int L, int xcd) { return xcd * (L / PD_XCDS) + (xcd < L % PD_XCDS ? xcd : L % PD_XCDS); }
// Every matrix of a layer is split by rows over ALL 32 units of the layer's XCD (unit idx: rows [idx * R / 32, (idx + 1) * R / 32));
// H of the units additionally run the attention of one head each -- slot 0: units 0 .. H-1, slot 1: units 32-H .. 31, so that a
// unit is a head in at most one slot (H <= 16)
__host__ __device__ inline bool pd_is_head(int slot, int idx, int H) { return slot == 0 ? idx < H : idx >= PD_CUS - H; }
__host__ __device__ inline int pd_head_index(int slot, int idx, int H) { return slot == 0 ? idx : idx - (PD_CUS - H); }
__host__ __device__ inline int pd_lm_rows(int V) { return (V + PD_GRID - 1) / PD_GRID; }
__host__ __device__ inline int pd_pad4(int n) { return (n + 3) & ~3; }
// Unit u of a phase = lane (u & 3) of a group of four lanes that share one 32-column block: rows 4 * (G / NB) + (u & 3) of this
// compute unit's (padded) rows, block G % NB, G = u / 4 -- one v_mfma_i32_4x4x4_16b_i8 is sixteen such groups
__host__ __device__ inline void pd_unit_rc(int u, int NB, int &row, int &blk) { const int G = u >> 2; row = (G / NB) * 4 + (u & 3); blk = G % NB; }

inline bool pd_shape_ok(int D, int I, int H, int KV, int hd, int V, int L) {
    if (hd != 64 || KV < 1 || H % KV || D != H * 64 || D % 32 || I % 32 || H < 1 || H > 16 || L < 1 || L > PD_MAXL) return false;
    // the instantiations of pd_decode_kernel: nano's shape with 9 (MHA) or 3 kv heads; a small test shape with 4 or 2
    if (!((D == 576 && I == 1536 && (KV == 9 || KV == 3)) || (D == 256 && I == 512 && (KV == 4 || KV == 2)))) return false;
    const int RQ = D + 2 * KV * 64;                                              // rows of [Q; K; V]
    if (RQ % PD_CUS || D % PD_CUS || I % PD_CUS) return false;                   // equal row shares
    const int NB = D / 32, NBI = I / 32, qr = pd_pad4(RQ / PD_CUS), wr = pd_pad4(D / PD_CUS), gr = pd_pad4(I / PD_CUS);
    if (qr * NB > PD_UQ * PD_THREADS || wr * NB > PD_UW * PD_THREADS || 2 * gr * NB > PD_UG * PD_THREADS || wr * NBI > PD_UD * PD_THREADS) return false;
    if (4 * qr > PD_THREADS || 4 * gr > PD_THREADS || 4 * wr > PD_THREADS || 2 * gr * (NB + 1) > PD_PART || wr * (NBI + 1) > PD_PART || qr * (NB + 1) > PD_PART) return false;
    const int lr = pd_pad4(pd_lm_rows(V));
    return lr <= 128 && lr * NB <= PD_ULM * PD_THREADS && lr * (NB + 1) <= PD_PART;
}

__host__ __device__ constexpr size_t pd_lds_bytes() {
    return (size_t)PD_ULM * 2 * PD_THREADS * 16 + (size_t)PD_ULM * PD_THREADS * 2       // LM-head slice: quants, fp16 scales
           + 640 * 4 + 20 * 128 + PD_NBI_MAX * 128                                      // x, digit images of x * g and of h
           + (size_t)PD_PART * 4 + 256 * 4 + 16 * 8                                     // block products, row sums, float64 partials
           + (192 + 64 + 8 * 68 + PD_MAX_PASSES * 66) * 4 + 64 * 4 + 80 * 4            // q | k | v, attention output, pass partials, misc, block scales
           + (size_t)PD_SLOTS * PD_UNITS * PD_THREADS * 2;                              // fp16 scales of the register-resident units
}

struct PdParams {
    int D, I, H, V, L, seq_len, rope_conj, n_steps, token0, pos0, spin_limit;
    float eps, scale;
    unsigned tag_base;
    const uint4 *wimg;               // [256][2][9][2][512] x 16 B: lane images of the layer weights (pd_pack_kernel)
    const unsigned short *simg;      // [256][2][9][512] fp16 scale bits
    const uint4 *lmimg;              // [256][5][2][512]
    const unsigned short *lmsimg;    // [256][5][512]
    const float *norms;              // [L][2][D] attn_norm | ffn_norm of every layer, then [D] output_norm (packed by pd_build)
    const uint8_t *embd_raw;         // Q8_0 rows of token_embd (a Q4_0 / Q5_0 file: its values re-blocked as Q8_0 by pd_embd_q8_kernel)
    const float *rope_cos, *rope_sin;
    float *kcache, *vcache;          // layer 0 of the stream: [kv head][seq][64]
    long long kv_layer_stride;
    pd_u64 *gx;                      // [L + 1][D]: input x of layer l >= 1; [L] = the final residual stream
    pd_u64 *gqkv;                    // [L][3 D]: q | k | v rows of the position before RoPE (row-split over the XCD's units, read by the heads)
    pd_u64 *go, *gxp;                // [L][D]: heads' attention outputs; x' = x + WO o
    pd_u64 *gh;                      // [L][I]: SiLU(gate) * up
    pd_u64 *gpart;                   // [L][H][PD_PARTS - 1][66]: a helper's merged (max, sum, sum p v[64]) of its passes, for the head's owner
    pd_u32x4 *gam;                   // [256] {tag, max bits, index, -}
    unsigned *census, *census_next;  // [8] tickets per XCD, [8] arrivals of THIS launch; the next launch's words (zeroed here: no memset between launches)
    int *ids_out;                    // [n_steps]
    float *logits;                   // [V] of the last step
    float *host_logits;              // optional: the last step's logits also go straight into this (pinned, device-visible) host buffer
    unsigned *status, *host_status;
    long long *dbg;                  // optional stamps of (xcd 0, unit 0), thread 0
    // resident session (per-call Forward: go/main.go:173-219 calls Forward once per token and samples on the host): the launch
    // stays on the chip between calls.  After step s the doorman (unit 0 of layer 0's XCD) stores {s + 1, argmax} into *host_done
    // and polls *mbox (both pinned host words) for {s + 1, token} -- the caller's next token -- or for a quit (token 0xffffffff);
    // idle_ticks (100 MHz) without a command end the launch the same way, so a caller that went away costs the chip that long.
    int session;
    long long idle_ticks;
    const pd_u64 *mbox;
    pd_u64 *host_done;
    pd_u64 *gtok;                    // the doorman's {tag, token} for the other units of its XCD
};

struct PdPackParams {
    const uint8_t *raw[PD_MAXL][7];  // q, k, v, o, gate, up, down: raw GGUF tensors on the device, Q8_0 / Q4_0 / Q5_0 blocks (rtype)
    const uint8_t *lm_raw;           // output.weight (or token_embd for tied heads)
    unsigned char rtype[PD_MAXL][7], lm_type;      // WT_Q8_0 / WT_Q4_0 / WT_Q5_0: every one is int8 quants x an fp16 d (n - 8, q - 16)
    int D, I, H, V, L, KV;
    uint4 *wimg; unsigned short *simg; uint4 *lmimg; unsigned short *lmsimg;
};

// One block per (compute unit, unit slot): lane l's 34-byte block of that slot, split into two 16-byte chunks and the scale.
// grid (256, 2 * 9 + 5), 512 threads.
__global__ void __launch_bounds__(PD_THREADS) pd_pack_kernel(PdPackParams P) {
    const int cu = blockIdx.x, xcd = cu / PD_CUS, idx = cu % PD_CUS, unit = blockIdx.y, tid = threadIdx.x;
    const int NB = P.D / 32, NBI = P.I / 32;
    const uint8_t *src = nullptr;     // the (row, block)'s first byte, or null = zeros
    int bt = WT_Q8_0;                 // its block type
    auto bsz = [](int t) { return t == WT_Q8_0 ? 34 : t == WT_Q4_0 ? 18 : 22; };
    if (unit < PD_SLOTS * PD_UNITS) {
        const int s = unit / PD_UNITS, k = unit % PD_UNITS;
        if (s < pd_nslots(P.L, xcd)) {
            const int layer = pd_first(P.L, xcd) + s;
            const int kvd = P.KV * 64;
            const int qr = (P.D + 2 * kvd) / PD_CUS, wr = P.D / PD_CUS, gr = P.I / PD_CUS;       // this unit's rows of Q | K | V, of WO / down, of gate / up
            int row, blk;
            if (k < PD_UQ) {
                const int u = k * PD_THREADS + tid;
                pd_unit_rc(u, NB, row, blk);
                if (u < pd_pad4(qr) * NB && row < qr) {
                    const int R = idx * qr + row, sect = R < P.D ? 0 : R < P.D + kvd ? 1 : 2;          // row R of [q; k; v] (k, v: KV heads)
                    const int rr = sect == 0 ? R : sect == 1 ? R - P.D : R - P.D - kvd;
                    bt = P.rtype[layer][sect];
                    src = P.raw[layer][sect] + ((size_t)rr * NB + blk) * bsz(bt);
                }
            } else if (k < PD_UQ + PD_UW) {
                const int u = (k - PD_UQ) * PD_THREADS + tid;
                pd_unit_rc(u, NB, row, blk);
                if (u < pd_pad4(wr) * NB && row < wr) { bt = P.rtype[layer][3]; src = P.raw[layer][3] + ((size_t)(idx * wr + row) * NB + blk) * bsz(bt); }
            } else if (k < PD_UQ + PD_UW + PD_UG) {
                const int u = (k - PD_UQ - PD_UW) * PD_THREADS + tid, grp = pd_pad4(gr);
                pd_unit_rc(u, NB, row, blk);
                if (u < 2 * grp * NB) {            // rows [0, grp): gate, [grp, 2 grp): up -- padding rows stay zero
                    const bool up = row >= grp;
                    const int rr = up ? row - grp : row;
                    if (rr < gr) { bt = P.rtype[layer][up ? 5 : 4]; src = P.raw[layer][up ? 5 : 4] + ((size_t)(idx * gr + rr) * NB + blk) * bsz(bt); }
                }
            } else {
                const int u = (k - PD_UQ - PD_UW - PD_UG) * PD_THREADS + tid;
                pd_unit_rc(u, NBI, row, blk);
                if (u < pd_pad4(wr) * NBI && row < wr) { bt = P.rtype[layer][6]; src = P.raw[layer][6] + ((size_t)(idx * wr + row) * NBI + blk) * bsz(bt); }
            }
        }
    } else {
        const int k = unit - PD_SLOTS * PD_UNITS, u = k * PD_THREADS + tid, lr = pd_lm_rows(P.V);
        const int r0 = cu * lr, nrows = max(0, min(lr, P.V - r0));
        int row, blk;
        pd_unit_rc(u, NB, row, blk);
        if (u < pd_pad4(lr) * NB && row < nrows) { bt = P.lm_type; src = P.lm_raw + ((size_t)(r0 + row) * NB + blk) * bsz(bt); }
    }
    uint4 lo = make_uint4(0, 0, 0, 0), hi = lo;
    unsigned short d16 = 0;
    if (src) {
        // the block as 32 int8 and its fp16 d: Q8_0 as stored (go/quant.go:120-165); Q4_0 nibble - 8 (low nibbles = elements 0-15, high
        // = 16-31: go/quant.go:45-94); Q5_0 (nibble | high bit << 4) - 16 (go/quant.go:405-420) -- every value an int8, the same d
        uint8_t b[34];
        const int nbytes = bsz(bt);
        for (int i = 0; i < 34; i++) b[i] = i < nbytes ? src[i] : 0;
        d16 = (unsigned short)(b[0] | (b[1] << 8));
        int8_t q[32];
        if (bt == WT_Q8_0) { for (int i = 0; i < 32; i++) q[i] = (int8_t)b[2 + i]; }
        else if (bt == WT_Q4_0) { for (int i = 0; i < 16; i++) { q[i] = (int8_t)((b[2 + i] & 0x0F) - 8); q[16 + i] = (int8_t)((b[2 + i] >> 4) - 8); } }
        else {
            const uint32_t qh = (uint32_t)b[2] | ((uint32_t)b[3] << 8) | ((uint32_t)b[4] << 16) | ((uint32_t)b[5] << 24);
            for (int i = 0; i < 32; i++) {
                const int nib = i < 16 ? (b[6 + i] & 0x0F) : (b[6 + i - 16] >> 4);
                q[i] = (int8_t)((nib | (int)(((qh >> i) & 1u) << 4)) - 16);
            }
        }
        auto w32 = [&](int o) { return (unsigned)(uint8_t)q[o] | ((unsigned)(uint8_t)q[o + 1] << 8) | ((unsigned)(uint8_t)q[o + 2] << 16) | ((unsigned)(uint8_t)q[o + 3] << 24); };
        lo = make_uint4(w32(0), w32(4), w32(8), w32(12));
        hi = make_uint4(w32(16), w32(20), w32(24), w32(28));
    }
    if (unit < PD_SLOTS * PD_UNITS) {
        const size_t base = ((size_t)cu * PD_SLOTS * PD_UNITS + unit);
        P.wimg[(base * 2 + 0) * PD_THREADS + tid] = lo;
        P.wimg[(base * 2 + 1) * PD_THREADS + tid] = hi;
        P.simg[base * PD_THREADS + tid] = d16;
    } else {
        const size_t base = (size_t)cu * PD_ULM + (unit - PD_SLOTS * PD_UNITS);
        P.lmimg[(base * 2 + 0) * PD_THREADS + tid] = lo;
        P.lmimg[(base * 2 + 1) * PD_THREADS + tid] = hi;
        P.lmsimg[base * PD_THREADS + tid] = d16;
    }
}

// token_embd of a Q4_0 / Q5_0 file as Q8_0 blocks (int8 of the same values, the same d): the decode kernel's in-launch embedding
// lookup (embed_value, go/model.go:389-446) then has one format.  One thread per block.
__global__ void pd_embd_q8_kernel(const uint8_t *src, int src_type, long long nblocks, uint8_t *dst) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nblocks; i += (long long)gridDim.x * blockDim.x) {
        const uint8_t *b = src + i * (src_type == WT_Q4_0 ? 18 : 22);
        uint8_t *o = dst + i * 34;
        o[0] = b[0]; o[1] = b[1];
        if (src_type == WT_Q4_0) {
            for (int j = 0; j < 16; j++) { o[2 + j] = (uint8_t)(int8_t)((b[2 + j] & 0x0F) - 8); o[18 + j] = (uint8_t)(int8_t)((b[2 + j] >> 4) - 8); }
        } else {
            const uint32_t qh = (uint32_t)b[2] | ((uint32_t)b[3] << 8) | ((uint32_t)b[4] << 16) | ((uint32_t)b[5] << 24);
            for (int j = 0; j < 32; j++) {
                const int nib = j < 16 ? (b[6 + j] & 0x0F) : (b[6 + j - 16] >> 4);
                o[2 + j] = (uint8_t)(int8_t)((nib | (int)(((qh >> j) & 1u) << 4)) - 16);
            }
        }
    }
}

// ---- device helpers --------------------------------------------------------------------------------------------------------

__device__ __forceinline__ __amdgpu_buffer_rsrc_t pd_rsrc(const void *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
// L1-bypassing loads the compiler counts (buffer_load ... sc1): served by this XCD's L2, or by the fabric for a line a
// write-through store dropped
__device__ __forceinline__ pd_u64 pd_ld8(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    const pd_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)byte_off, 0, 16);
    return ((pd_u64)v.y << 32) | v.x;
}
__device__ __forceinline__ float4 pd_ld16f(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    const pd_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 16);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
// a granule for a consumer on THIS XCD stays in the XCD's L2 (workgroup-scope store: written through the L1 only); one
// for another XCD is written through to the fabric (agent scope)
template <bool CROSS>
__device__ __forceinline__ void pd_publish(pd_u64 *p, unsigned tag, float v) {
    const pd_u64 g = ((pd_u64)tag << 32) | __float_as_uint(v);
    if (CROSS) __hip_atomic_store(p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_store(p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

struct PdPoll {
    unsigned *status, *host_status;
    int spin_limit;
    int *lds_dead;
};
__device__ __forceinline__ void pd_give_up(const PdPoll &Q, unsigned code, int tid) {
    if ((tid & 63) == 0) { const unsigned old = atomicOr(Q.status, code); *Q.host_status = old | code; *Q.lds_dead = 1; }
}
// n granules -> v[k] = value of granule tid + 512 k (a wavefront leaves when all of its granules carry the tag); bounded
template <int NPT>
__device__ __forceinline__ void pd_gather(const pd_u64 *g, int n, unsigned tag, float (&v)[NPT], const PdPoll &Q, unsigned code, int tid) {
    const __amdgpu_buffer_rsrc_t r = pd_rsrc(g, (unsigned)n * 8u);
    unsigned off[NPT];
#pragma unroll
    for (int k = 0; k < NPT; k++) off[k] = (unsigned)min(tid + k * PD_THREADS, n - 1) * 8u;
    pd_u64 q[NPT];
    for (int spins = 0;; spins++) {
#pragma unroll
        for (int k = 0; k < NPT; k++) q[k] = pd_ld8(r, off[k]);
        bool ok = true;
#pragma unroll
        for (int k = 0; k < NPT; k++) ok &= (unsigned)(q[k] >> 32) == tag;
        if (__all(ok)) break;
        if (spins >= Q.spin_limit || ((spins & 63) == 63 && __hip_atomic_load(Q.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
            pd_give_up(Q, code, tid);
            break;
        }
        // (no s_sleep between polls: they are this XCD's own L2 round trips, nobody streams beside them -- 7920 -> 8040 tok/s)
    }
#pragma unroll
    for (int k = 0; k < NPT; k++) v[k] = __uint_as_float((unsigned)q[k]);
}

// U unit slots of a phase: unit u = k * 512 + tid = (row u / NB, block u % NB) of this compute unit's rows; the lane's
// 32 int8 of slot k times the block's 32 inputs (LDS, transposed so that consecutive blocks are consecutive float4s: the
// lanes of a wavefront read conflict-free), times d -> part[row * NBP + block]
typedef int pd_i32x4 __attribute__((ext_vector_type(4)));

// lanes j, j + 16 (permlane16_swap) and j, j + 32 (permlane32_swap) meet on the vector pipe: gfx950's swaps instead of
// ds_bpermute round trips (nl_batch.h rows4_sum)
__device__ __forceinline__ float pd_pair16_max(float v) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
}
__device__ __forceinline__ float pd_rows4_sum(float v) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float dpp_row_sum_f32(float v) {      // sum of the lane's row of 16, in every lane of the row
    v += dpp_f32<DPP_QUAD_XOR1>(v);
    v += dpp_f32<DPP_QUAD_XOR2>(v);
    v += dpp_f32<DPP_HALF_MIRROR>(v);
    v += dpp_f32<DPP_ROW_MIRROR>(v);
    return v;
}
__device__ __forceinline__ double pd_wave_sum_f64(double v) {
    v += dpp_f64<DPP_QUAD_XOR1>(v);
    v += dpp_f64<DPP_QUAD_XOR2>(v);
    v += dpp_f64<DPP_HALF_MIRROR>(v);
    v += dpp_f64<DPP_ROW_MIRROR>(v);   // every lane: sum of its row of 16
    {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
    }
    {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        v = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
    }
    return v;
}

// One 32-element block of a vector as the A operand of v_mfma_i32_4x4x4_16b_i8.  The block's values are scaled by a power of
// two so that the largest fits 31 bits, rounded to integers X (|X| <= 2^30: 2^-30 of the block's maximum per element, 64 times
// finer than a float32 mantissa) and written as four SIGNED base-256 digits X = l0 + 256 l1 + 65536 l2 + 2^24 l3: the image holds,
// per block, digit m of its 32 elements as 32 consecutive bytes ([block][m][32]).  A 4x4x4 product of digit rows with int8
// weight columns is exact in int32; the digits are recombined in float32 afterwards (pd_units).  The calling HALF wavefront
// (32 lanes, all active) holds the block: e = element index of this lane, valid = the element exists.
__device__ __forceinline__ void pd_limbs(float v, bool valid, int e, unsigned char *img, float *scl, int lane) {
    float a = valid ? fabsf(v) : 0.f;
    a = fmaxf(a, dpp_f32<DPP_QUAD_XOR1>(a));
    a = fmaxf(a, dpp_f32<DPP_QUAD_XOR2>(a));
    a = fmaxf(a, dpp_f32<DPP_HALF_MIRROR>(a));
    a = fmaxf(a, dpp_f32<DPP_ROW_MIRROR>(a));          // max of the row of 16
    a = pd_pair16_max(a);                              // ... of the block's 32 lanes
    unsigned ex = __float_as_uint(a) >> 23;            // biased exponent of the block's maximum (a >= 0)
    ex = min(max(ex, 30u), 254u);
    const float scale = __uint_as_float((283u - ex) << 23);      // 2^(29 - (ex - 127)): |v * scale| < 2^30
    const int X = valid ? __float2int_rn(v * scale) : 0;
    const int l0 = (X << 24) >> 24, X1 = (X - l0) >> 8;
    const int l1 = (X1 << 24) >> 24, X2 = (X1 - l1) >> 8;
    const int l2 = (X2 << 24) >> 24, l3 = (X2 - l2) >> 8;
    unsigned char *p = img + (e >> 5) * 128 + (e & 31);
    p[0] = (unsigned char)l0; p[32] = (unsigned char)l1; p[64] = (unsigned char)l2; p[96] = (unsigned char)l3;
    if ((lane & 31) == 0) scl[e >> 5] = __uint_as_float((ex - 29u) << 23);      // 1 / scale
}

// U unit slots of a phase (pd_unit_rc): the lane's 32 int8 of slot k are the B operand of eight chained 4x4x4 products whose A
// operand is digit (lane & 3) of the block (two 16-byte LDS reads); the lane then holds the four digit sums of ITS row, recombines
// them in float32 and multiplies by d and the block's 1 / scale -> part[row * NBP + block].  (The float32 dot product of the
// reference, go/quant.go:149-165, has a rounding per element; this form is exact up to the 2^-30 input rounding and three
// float32 roundings per block.)  One MFMA = 256 weights: a quarter of the instructions of the VALU form.
template <int NU, int NB, class WF>   // NU unit slots; NB blocks per row; wf(k, lo, hi, d): the lane's 32 int8 and fp16 scale bits of slot k
__device__ __forceinline__ void pd_units_impl(WF wf, const uint4 *img, const float *scl, int total, float *part, int tid) {
    constexpr int NBP = NB | 1;
    asm volatile("" : "+v"(tid));        // (addresses are derived per phase, not shared across phases and kept live)
    // Units go two at a time: the two chains of eight dependent products interleave (a 4x4x4 product has a few cycles of latency
    // its successor would otherwise wait for), their digit rows are requested together, and the scheduling fence sits between
    // pairs -- hoisting every unit's operands in front of the first product costs registers this kernel does not have.
#pragma unroll
    for (int k0 = 0; k0 < NU; k0 += 2) {
        if (k0 * PD_THREADS < total) {
            constexpr bool kPair = true;
            (void)kPair;
            const bool has2 = k0 + 1 < NU;
            const int k1 = has2 ? k0 + 1 : k0;
            const bool two = has2 && k1 * PD_THREADS < total;
            unsigned u[2], row[2], blk[2];
            uint4 a0[2], a1[2], wl[2], wh[2];
            unsigned d16[2];
            float sb[2];
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int k = j ? k1 : k0;
                u[j] = (unsigned)(k * PD_THREADS + tid);
                const unsigned G = u[j] >> 2, rgp = G / (unsigned)NB;
                blk[j] = G - rgp * (unsigned)NB; row[j] = rgp * 4u + (u[j] & 3u);
                const uint4 *ap = img + blk[j] * 8u + (u[j] & 3u) * 2u;
                a0[j] = ap[0]; a1[j] = ap[1]; sb[j] = scl[blk[j]];
                wf(k, wl[j], wh[j], d16[j]);
            }
            pd_i32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#define PD_MF(A, W, ACC) ACC = __builtin_amdgcn_mfma_i32_4x4x4i8((int)(A), (int)(W), ACC, 0, 0, 0)
            if (has2) {
                PD_MF(a0[0].x, wl[0].x, acc0); PD_MF(a0[1].x, wl[1].x, acc1);
                PD_MF(a0[0].y, wl[0].y, acc0); PD_MF(a0[1].y, wl[1].y, acc1);
                PD_MF(a0[0].z, wl[0].z, acc0); PD_MF(a0[1].z, wl[1].z, acc1);
                PD_MF(a0[0].w, wl[0].w, acc0); PD_MF(a0[1].w, wl[1].w, acc1);
                PD_MF(a1[0].x, wh[0].x, acc0); PD_MF(a1[1].x, wh[1].x, acc1);
                PD_MF(a1[0].y, wh[0].y, acc0); PD_MF(a1[1].y, wh[1].y, acc1);
                PD_MF(a1[0].z, wh[0].z, acc0); PD_MF(a1[1].z, wh[1].z, acc1);
                PD_MF(a1[0].w, wh[0].w, acc0); PD_MF(a1[1].w, wh[1].w, acc1);
            } else {
                PD_MF(a0[0].x, wl[0].x, acc0); PD_MF(a0[0].y, wl[0].y, acc0); PD_MF(a0[0].z, wl[0].z, acc0); PD_MF(a0[0].w, wl[0].w, acc0);
                PD_MF(a1[0].x, wh[0].x, acc0); PD_MF(a1[0].y, wh[0].y, acc0); PD_MF(a1[0].z, wh[0].z, acc0); PD_MF(a1[0].w, wh[0].w, acc0);
            }
#undef PD_MF
            const float f0 = fmaf(fmaf(fmaf((float)acc0[3], 256.f, (float)acc0[2]), 256.f, (float)acc0[1]), 256.f, (float)acc0[0]);
            if ((int)u[0] < total) part[row[0] * (unsigned)NBP + blk[0]] = f0 * (h2f_bits(d16[0]) * sb[0]);
            if (has2) {
                const float f1 = fmaf(fmaf(fmaf((float)acc1[3], 256.f, (float)acc1[2]), 256.f, (float)acc1[1]), 256.f, (float)acc1[0]);
                if (two && (int)u[1] < total) part[row[1] * (unsigned)NBP + blk[1]] = f1 * (h2f_bits(d16[1]) * sb[1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}
// the layer matrices: quants in registers (slots K0 .. K0 + NU - 1 of the layer slot), scales in LDS
template <int K0, int NU, int NB>
__device__ __forceinline__ void pd_units(const uint4 (&lo)[PD_UNITS], const uint4 (&hi)[PD_UNITS], const unsigned short *sc /* LDS: [9][512] of the slot */, const uint4 *img,
                                         const float *scl, int total, float *part, int tid) {
    pd_units_impl<NU, NB>([&](int k, uint4 &l, uint4 &h, unsigned &d) { l = lo[K0 + k]; h = hi[K0 + k]; d = sc[(K0 + k) * PD_THREADS + tid]; }, img, scl, total, part, tid);
}

// argmax over a wavefront whose lanes hold candidates in ascending index order: the maximum on DPP, the lowest lane that holds
// it (go/main.go:400-408: strict '>', the earlier index wins) by a ballot -- no ds_bpermute round trips
__device__ __forceinline__ void pd_wave_argmax(float &best, int &bidx) {
    const float m = wave_max_f32(best);
    const unsigned long long mask = __ballot(best == m);
    const int l = mask ? (int)__ffsll((long long)mask) - 1 : 0;
    bidx = __builtin_amdgcn_readlane(bidx, __builtin_amdgcn_readfirstlane(l));
    best = m;
}

// exp_f64_as_f32 (nl_kernels.h: float32(exp(float64(x))), go/quant.go:619, :629-631) with its thirteen coefficients made
// opaque at the point of use: they are loop-invariant, and in this kernel -- 154 of 256 registers pinned under weights -- the
// compiler hoisted them in front of the token loop, spilled them and fetched them back from scratch memory at every call site.
// Here each one is two scalar moves next to its use.  Same operations in the same order: bit-identical results.
__device__ __forceinline__ double pd_k(double c) { asm volatile("" : "+s"(c)); return c; }
__device__ __forceinline__ float pd_exp(float xf) {
    const double x = fmin(fmax((double)xf, -750.0), 710.0);
    const double k = rint(x * pd_k(1.44269504088896338700e+00));
    double r = fma(k, pd_k(-6.93147180369123816490e-01), x);
    r = fma(k, pd_k(-1.90821492927058770002e-10), r);
    const double r2 = r * r, r4 = r2 * r2, r8 = r4 * r4;
    const double a0 = 1.0 + r;
    const double a1 = fma(r, pd_k(1.0 / 6), 0.5), a2 = fma(r, pd_k(1.0 / 120), pd_k(1.0 / 24)), a3 = fma(r, pd_k(1.0 / 5040), pd_k(1.0 / 720));
    const double a4 = fma(r, pd_k(1.0 / 362880), pd_k(1.0 / 40320)), a5 = fma(r, pd_k(1.0 / 39916800), pd_k(1.0 / 3628800));
    const double a6 = fma(r, pd_k(1.0 / 6227020800.0), pd_k(1.0 / 479001600));
    const double b0 = fma(a1, r2, a0), b1 = fma(a3, r2, a2), b2 = fma(a5, r2, a4);
    const double d0 = fma(b1, r4, b0), d1 = fma(a6, r4, b2);
    const double p = fma(d1, r8, d0);
    const float res = (float)ldexp(p, (int)k);
    return xf == xf ? res : xf;
}

// sum of squares of a width-D vector spread two elements per thread -> 1 / rms (go/quant.go:597-607: float64 sum); every
// compute unit adds the same partials in the same order.  Contains a workgroup barrier.  1 / sqrt(m) = v_rsq_f64 + two Newton
// steps (relative error < 1e-30 before the float32 rounding) instead of the IEEE divide and square root: sixty dependent
// float64 instructions shorter.
__device__ __forceinline__ float pd_inv_rms(float a, float b, bool v0, bool v1, int lane, int wave, double *dred, int D, float eps) {
    double ss = 0.0;
    if (v0) ss = fma((double)a, (double)a, ss);
    if (v1) ss = fma((double)b, (double)b, ss);
    // (the lane's two squares are exact in float64; the 64 lanes of a wavefront meet in float32 -- one rounding of 2^-24 per
    //  add on a sum of positive terms, well inside the float32 the result is rounded to -- on the DPP / swap path, a third of the
    //  instructions of the float64 form; wavefronts are added in float64)
    const float sw = pd_rows4_sum(dpp_row_sum_f32((float)ss));
    if (lane == 0) dred[wave] = (double)sw;
    __syncthreads();
    double t[PD_WAVES];
#pragma unroll
    for (int w = 0; w < PD_WAVES; w++) t[w] = dred[w];
    double tot = 0.0;
#pragma unroll
    for (int w = 0; w < PD_WAVES; w++) tot += t[w];
    const double m = tot / (double)D + (double)eps;          // (D is a compile-time constant at the call sites: a multiply)
    double y = __builtin_amdgcn_rsq(m);
    y = y * fma(-0.5 * m * y, y, 1.5);
    y = y * fma(-0.5 * m * y, y, 1.5);
    return (float)y;
}

// a row's block products part[0 .. N) added in block order; LANES lanes share a row (q = this lane's index among them, adjacent
// lanes): each adds a contiguous run, the runs are joined in lane order.  All reads are issued before the first add.
template <int N, int LANES>
__device__ __forceinline__ float pd_rowsum(const float *p, int q) {
    constexpr int PER = (N + LANES - 1) / LANES;
    float v[PER];
#pragma unroll
    for (int i = 0; i < PER; i++) v[i] = p[min(q * PER + i, N - 1)];
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < PER; i++) a += (q * PER + i < N) ? v[i] : 0.f;
    if (LANES >= 2) { const float o = dpp_f32<DPP_QUAD_XOR1>(a); a = (q & 1) ? o + a : a + o; }
    if (LANES == 4) { const float o = dpp_f32<DPP_QUAD_XOR2>(a); a = (q & 2) ? o + a : a + o; }
    return a;
}

template <int NB, int NBI, int KVH>   // D / 32, I / 32, kv heads: one instantiation per shape class
__global__ void __launch_bounds__(PD_THREADS, 2) pd_decode_kernel(PdParams P0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4 *lmw = reinterpret_cast<uint4 *>(smem);                                          // [5][2][512]
    unsigned short *lms = reinterpret_cast<unsigned short *>(lmw + PD_ULM * 2 * PD_THREADS);   // [5][512]
    float *xraw = reinterpret_cast<float *>(lms + PD_ULM * PD_THREADS);                    // [640] the residual stream this phase adds to
    unsigned char *xl = reinterpret_cast<unsigned char *>(xraw + 640);                     // [20][4][32] digit image of x * g (or of o)
    unsigned char *hl = xl + 20 * 128;                                                     // [48][4][32] digit image of h
    float *part = reinterpret_cast<float *>(hl + PD_NBI_MAX * 128);                        // [PD_PART]
    float *rs = part + PD_PART;                                                            // [256]: gate | up row sums at 0 / 128
    double *dred = reinterpret_cast<double *>(rs + 256);                                   // [16]
    float *qs = reinterpret_cast<float *>(dred + 16);                                      // q | kcur | vcur [192]
    float *kcur = qs + 64, *vcur = qs + 128;
    float *on = qs + 192;                                                                  // [64]
    float *wpart = on + 64;                                                                // [8][68]
    float *chunk = wpart + 8 * 68;                                                         // [4][66]
    float *bv = chunk + PD_MAX_PASSES * 66;                                                // [8]
    int *bi = reinterpret_cast<int *>(bv + 8);                                             // [8]
    int *misc = bi + 8;                                                                    // [0] ticket, [1] census ok, [2] dead, [3] token
    float *xs = reinterpret_cast<float *>(misc + 8);                                       // [20] 1 / scale of every block of x
    float *hs = xs + 20;                                                                   // [48] ... of h
    unsigned short *wsl = reinterpret_cast<unsigned short *>(hs + 60);                     // [2][9][512] fp16 d of the register-resident units (registers hold the quants only)

    const int tid0 = threadIdx.x;
    constexpr int D = NB * 32, I = NBI * 32, H = D / 64, NBP = NB | 1, NBIP = NBI | 1;
    constexpr int KVD = KVH * 64, GQ = H / KVH;          // GQA (go/model.go:557-587): query head h reads kv head h / GQ
    static_assert(H % KVH == 0, "query heads per kv head");
    constexpr int QR = (D + 2 * KVD) / PD_CUS, QRP = (QR + 3) & ~3, WR = D / PD_CUS, WRP = (WR + 3) & ~3, GR = I / PD_CUS, GRP = (GR + 3) & ~3;   // this unit's rows per matrix
    const int L = P0.L;
    unsigned xcd;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcd));
    xcd &= 7u;

    // ---- census: which unit of which XCD am I (placement is observed, never assumed) ----
    if (blockIdx.x == 0 && tid0 < 16) P0.census_next[tid0] = 0u;     // (the launch that used those words has ended; the next one starts after this one)
    if (tid0 == 0) {
        misc[2] = 0;
        const unsigned t = atomicAdd(P0.census + xcd, 1u);
        __threadfence();
        atomicAdd(P0.census + 8, 1u);
        int spins = 0;
        while (__hip_atomic_load(P0.census + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)PD_GRID && ++spins < 2000000) __builtin_amdgcn_s_sleep(2);
        bool ok = spins < 2000000 && gridDim.x == (unsigned)PD_GRID;
        for (int x = 0; x < PD_XCDS; x++) ok = ok && __hip_atomic_load(P0.census + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)PD_CUS;
        misc[0] = (int)t;
        misc[1] = ok ? 1 : 0;
    }
    __syncthreads();
    const int idx0 = misc[0];
    if (!misc[1]) {
        if (tid0 == 0) { atomicOr(P0.status, 64u); *P0.host_status = 64u; }
        return;
    }
    const int cu = (int)xcd * PD_CUS + idx0;
    const int nslots = pd_nslots(L, (int)xcd);
    const unsigned xcd0 = xcd;
    const PdPoll Q{P0.status, P0.host_status, P0.spin_limit, misc + 2};
#define PD_STAMP(i) do { if (P0.dbg && cu == 0 && tid0 == 0) P0.dbg[i] = wall_clock64(); } while (0)
    PD_STAMP(0);

    // ---- the weights of this unit's layers -> registers, its LM-head rows -> LDS (once per launch) ----
    uint4 wlo[PD_SLOTS][PD_UNITS], whi[PD_SLOTS][PD_UNITS];
#pragma unroll
    for (int s = 0; s < PD_SLOTS; s++) {
        const int ss = min(s, max(nslots - 1, 0));      // an XCD with one layer loads it twice: slot 1 is never used
#pragma unroll
        for (int k = 0; k < PD_UNITS; k++) {
            const size_t base = ((size_t)cu * PD_SLOTS * PD_UNITS + ss * PD_UNITS + k);
            wlo[s][k] = P0.wimg[(base * 2 + 0) * PD_THREADS + tid0];
            whi[s][k] = P0.wimg[(base * 2 + 1) * PD_THREADS + tid0];
        }
#pragma unroll
        for (int k = 0; k < PD_UNITS; k++) wsl[(s * PD_UNITS + k) * PD_THREADS + tid0] = P0.simg[((size_t)cu * PD_SLOTS * PD_UNITS + ss * PD_UNITS + k) * PD_THREADS + tid0];
    }
#pragma unroll
    for (int i = 0; i < PD_ULM * 2; i++) lmw[i * PD_THREADS + tid0] = P0.lmimg[((size_t)cu * PD_ULM * 2 + i) * PD_THREADS + tid0];
#pragma unroll
    for (int k = 0; k < PD_ULM; k++) lms[k * PD_THREADS + tid0] = P0.lmsimg[((size_t)cu * PD_ULM + k) * PD_THREADS + tid0];
    __syncthreads();
    PD_STAMP(1);

    int token = P0.token0;
    const int n_steps = P0.n_steps;
    for (int step = 0; step <= n_steps; step++) {
        // Values the whole step derives its addresses from are made opaque once per step: the loop would otherwise have every
        // LDS / granule / cache address of every phase hoisted in front of it and kept live (hundreds of registers -- and 154
        // of the 256 hold weights)
        // ... and the kernel arguments are read again through an opaque pointer where a step needs them, instead of all of them
        // (and everything derived from them) sitting in scalar registers across the loop: the scalar file spilled 250 values
        // into vector lanes
        const __attribute__((address_space(4))) PdParams *pk = (const __attribute__((address_space(4))) PdParams *)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(pk));
        const __attribute__((address_space(4))) PdParams &P = *pk;
        int tid = tid0, idx = idx0;
        unsigned xcd = xcd0;
        asm volatile("" : "+v"(tid), "+s"(idx), "+s"(xcd));
        // (roles, row ranges and layer numbers follow from idx / xcd with a few scalar operations per step; hoisted in front of
        //  the loop they were two hundred scalar values spilled into vector lanes)
        const int cu = (int)xcd * PD_CUS + idx;
        const int nslots = pd_nslots(L, (int)xcd), first = pd_first(L, (int)xcd);
        const bool embed_xcd = first == 0 && nslots > 0;     // this XCD owns layer 0: it turns the argmax into the next x
        const int lm_rows = pd_lm_rows(P.V), lm_r0 = cu * lm_rows, lm_n = max(0, min(lm_rows, P.V - lm_r0));
        // ... and so are the weight registers, in place (no instruction): their int8 -> float conversions are loop-invariant and
        // would otherwise be computed once in front of the loop -- 4608 floats per lane, spilled
#pragma unroll
        for (int s = 0; s < PD_SLOTS; s++) {
#pragma unroll
            for (int k = 0; k < PD_UNITS; k++) {
                asm volatile("" : "+v"(wlo[s][k].x), "+v"(wlo[s][k].y), "+v"(wlo[s][k].z), "+v"(wlo[s][k].w));
                asm volatile("" : "+v"(whi[s][k].x), "+v"(whi[s][k].y), "+v"(whi[s][k].z), "+v"(whi[s][k].w));
            }
        }
        int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
        int e0 = tid, e1 = tid + PD_THREADS;
        bool v0 = e0 < D, v1 = e1 < D;
        int c0 = v0 ? e0 : 0, c1 = v1 ? e1 : 0;
        // at every phase boundary the lane's index values are made opaque and derived again: nothing computed from them in
        // one phase (LDS / granule addresses) is shared with, or kept live into, a later one
#define PD_RELANE() do { asm volatile("" : "+v"(tid)); wave = __builtin_amdgcn_readfirstlane(tid >> 6); lane = tid & 63; e0 = tid; e1 = tid + PD_THREADS; \
                         v0 = e0 < D; v1 = e1 < D; c0 = v0 ? e0 : 0; c1 = v1 ? e1 : 0; } while (0)
        const int pos = P.pos0 + step;
        // The host rows of the previous step (resident session): acknowledged -- they bypass the caches, so waiting for the stores is
        // enough; a system-scope release fence would also write the XCD's L2 back, 17 us per step measured -- and counted on a
        // word of this launch's census block.  NOT in front of the unit's argmax granule: the acknowledgement crosses PCIe while
        // the granules travel and the token is picked; the doorman looks at the count before it tells the host.
#define PD_ACK_HOST_ROWS() do { if (P.session && P.host_logits && step > 0) { \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
            __syncthreads(); \
            if (tid == 0) __hip_atomic_fetch_add(P.census + 10, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } } while (0)
        // developer stamps (tools/persist_stamps.py): step 2 on XCD 1 -- unit 0 (a head of slot 0), unit H (a worker of slot 0)
#ifdef NL_PD_STAMPS
#define PD_ST(base, i) do { if (step == 2 && xcd == 1u && tid == 0 && P.dbg) P.dbg[(base) + (i)] = wall_clock64(); } while (0)
#else
#define PD_ST(base, i) do { } while (0)       // (the stamps' conditions cost scalar registers the production build does not have)
#endif
        const unsigned tag = P.tag_base + (unsigned)step + 1u;
        float xa = 0.f, xb = 0.f;       // this thread's two elements of the residual stream entering the next layer
        if (embed_xcd) {
            if (step > 0) {
                // ---- the 256 (max, index) pairs of the previous step -> token (go/main.go:400-408) ----
                float best = -INFINITY;
                int bidx = 0x7fffffff;
                if (tid < PD_GRID) {
                    const __amdgpu_buffer_rsrc_t r = pd_rsrc(P.gam, PD_GRID * 16u);
                    pd_u32x4 g;
                    for (int spins = 0;; spins++) {
                        g = __builtin_amdgcn_raw_buffer_load_b128(r, tid * 16, 0, 16);
                        if (__all(g.x == tag - 1u)) break;
                        if (spins >= P.spin_limit || ((spins & 63) == 63 && __hip_atomic_load(P.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                            pd_give_up(Q, 128u, tid);
                            break;
                        }
                        // (no s_sleep: see pd_gather)
                    }
                    best = __uint_as_float(g.y); bidx = (int)g.z;
                    pd_wave_argmax(best, bidx);          // (lane = compute unit = ascending vocabulary rows)
                    if (lane == 0) { bv[wave] = best; bi[wave] = bidx; }
                }
                __syncthreads();
                PD_RELANE();
                if (misc[2]) return;
                best = bv[0]; bidx = bi[0];
#pragma unroll
                for (int w = 1; w < PD_GRID / 64; w++)
                    if (bv[w] > best || (bv[w] == best && bi[w] < bidx)) { best = bv[w]; bidx = bi[w]; }
                token = bidx == 0x7fffffff ? 0 : bidx;      // all-NaN logits: the reference's loop never leaves index 0
                PD_ACK_HOST_ROWS();
                if (idx == 0 && tid == 0) {
                    P.ids_out[step - 1] = token;
                    if (P.session) {
                        if (P.host_logits) {      // every unit's rows of this step have been acknowledged
                            for (int spins = 0; __hip_atomic_load(P.census + 10, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)PD_GRID * (unsigned)step; spins++)
                                if (spins >= P.spin_limit) { pd_give_up(Q, 2048u, tid); break; }
                        }
                        // (a plain system-scope store: a release here would write this XCD's whole L2 back first)
                        __hip_atomic_store(P.host_done, ((pd_u64)(unsigned)step << 32) | (unsigned)token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                }
                __syncthreads();                              // bv / bi are reused by the LM head
                PD_RELANE();
            }
            if (step == P.n_steps) break;
            if (P.session && step > 0) {
                // ---- resident session: the caller's token for this step (or the end of the session) ----
                if (tid == 0) {
                    if (idx == 0) {
                        const long long t0 = wall_clock64();
                        unsigned tk = 0xffffffffu;
                        bool failed = false;
                        for (;;) {
                            const pd_u64 m = __hip_atomic_load(P.mbox, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            if ((unsigned)m == 0xffffffffu) break;                                           // quit
                            if ((unsigned)(m >> 32) == (unsigned)step) { tk = (unsigned)m; break; }
                            if (__hip_atomic_load(P.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { failed = true; break; }
                            if (wall_clock64() - t0 > P.idle_ticks) break;                                   // the caller went away
                            __builtin_amdgcn_s_sleep(4);
                        }
                        if (failed) misc[2] = 1;
                        else if (tk == 0xffffffffu) { const unsigned old = atomicOr(P.status, 256u); *P.host_status = old | 256u; misc[2] = 1; }
                        else pd_publish<false>(P.gtok, tag, __uint_as_float(tk));
                    }
                    if (!misc[2]) {
                        const __amdgpu_buffer_rsrc_t r = pd_rsrc(P.gtok, 8u);
                        pd_u64 g;
                        for (int spins = 0;; spins++) {
                            g = pd_ld8(r, 0);
                            if ((unsigned)(g >> 32) == tag) break;
                            if (spins >= P.spin_limit || ((spins & 63) == 63 && __hip_atomic_load(P.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                                pd_give_up(Q, 512u, tid);
                                break;
                            }
                            __builtin_amdgcn_s_sleep(1);
                        }
                        misc[3] = (int)(unsigned)g;
                    }
                }
                __syncthreads();
                PD_RELANE();
                if (misc[2]) return;
                token = misc[3];
                __syncthreads();
                PD_RELANE();
            }
            // ---- embedding row (go/model.go:389-446) ----
            xa = embed_value(P.embd_raw, WT_Q8_0, D, token, c0);
            xb = embed_value(P.embd_raw, WT_Q8_0, D, token, c1);
        } else {
            PD_ACK_HOST_ROWS();
            if (step == P.n_steps) break;
        }

        // =================================== this XCD's layers ===================================
#pragma unroll
        for (int s = 0; s < PD_SLOTS; s++) {
            if (s >= nslots) break;
            const int layer = first + s;
            const bool head = pd_is_head(s, idx, H);
            const bool next_here = s + 1 < nslots;           // the next layer lives on this XCD
            const int sb = s == 0 ? 8 : 63;                  // stamp base (slot 0 of XCD 1, unit 0: a head)
            (void)sb;
            if (s == 0 && idx == 0) PD_ST(sb, 0);
            // Attention roles of the slot: a head's OWNER (part 0: it also ropes and stores k | v) and, from the second pass on, up
            // to PD_PARTS - 1 HELPERS among the XCD's units that are not heads in this slot: part p takes the passes
            // nch - 1 - p, nch - 1 - p - PD_PARTS, ... (the last pass -- the one that holds this position -- is the owner's)
            const int nch = pos / 128 + 1, nparts = min(PD_PARTS, nch);
            int hidx = pd_head_index(s, idx, H), apart = 0;
            if (!head) {
                const int j = s == 0 ? idx - H : idx;             // the slot's non-heads, in order
                apart = 1 + j / H;
                hidx = j - (apart - 1) * H;
            }
            const bool arole = head || apart < nparts;          // (a helper beyond the passes there are has nothing to do)
            const int chf = nch - 1 - apart;                    // this unit's first (= highest) pass
            float *const kc = P.kcache + (long long)layer * P.kv_layer_stride + (long long)(arole ? hidx / GQ : 0) * P.seq_len * 64;
            float *const vc = P.vcache + (long long)layer * P.kv_layer_stride + (long long)(arole ? hidx / GQ : 0) * P.seq_len * 64;
            const __amdgpu_buffer_rsrc_t kr_ = pd_rsrc(kc, (unsigned)P.seq_len * 256u), vr_ = pd_rsrc(vc, (unsigned)P.seq_len * 256u);
            const int kr = lane >> 2, kq = lane & 3, vg = lane >> 4, vcl = lane & 15;
            float4 kreg[4], vreg[4];
            // (norm weights are requested before the polls they would otherwise wait behind)
            const float ga0 = P.norms[(size_t)(layer * 2) * D + c0], ga1 = P.norms[(size_t)(layer * 2) * D + c1];
            // ---- x of this layer (go/model.go:517): every unit of the XCD projects its rows of Q | K | V and adds residuals ----
            if (layer > 0) {
                float xv[2];
                pd_gather<2>(P.gx + (size_t)layer * D, D, tag, xv, Q, 1u, tid);
                xa = xv[0]; xb = xv[1];
            }
            if (s == 0 && idx == 0) PD_ST(sb, 1);
            if (v0) xraw[e0] = xa;
            if (v1) xraw[e1] = xb;
            pd_limbs(xa * ga0, v0, e0, xl, xs, lane);
            if (wave * 64 + PD_THREADS < D) pd_limbs(xb * ga1, v1, e1, xl, xs, lane);      // (wave-uniform: the wavefronts that hold second elements)
            {
                const float inv = pd_inv_rms(xa, xb, v0, v1, lane, wave, dred, D, P.eps);   // (its barrier also publishes the digit image)
                if (misc[2]) return;
                if (s == 0 && idx == 0) PD_ST(sb, 2);
                // ---- this unit's rows of [Q; K; V] (go/model.go:517-523): 3 D / 32 rows ----
                float ec = 0.f, es = 0.f;      // (heads: requested before the products, the rotation below must not wait for them)
                if (arole && tid < 128) { ec = P.rope_cos[pos * 32 + (tid & 31)]; es = P.rope_sin[pos * 32 + (tid & 31)]; }
                pd_units<0, PD_UQ, NB>(wlo[s], whi[s], wsl + s * PD_UNITS * PD_THREADS, reinterpret_cast<const uint4 *>(xl), xs, QRP * NB, part, tid);
                if (s == 0 && idx == 0) PD_ST(sb, 3);
                if (arole) {  // the cache rows of the unit's first pass are requested behind the dot products (rows beyond the pass repeat
                              // its last row; the row of this position comes from LDS): they arrive during the row sums and the exchange
                    const int t0 = chf * 128, lim = min(128, pos + 1 - t0);
                    const unsigned krow = (unsigned)(t0 + min(wave * 16 + kr, lim - 1)) * 16u + (unsigned)kq * 4u;
#pragma unroll
                    for (int kk = 0; kk < 4; kk++) {
                        kreg[kk] = pd_ld16f(kr_, (krow + kk) * 16u);
                        vreg[kk] = pd_ld16f(vr_, (unsigned)((t0 + min(wave * 16 + 4 * vg + kk, lim - 1)) * 16 + vcl) * 16u);
                    }
                }
                __syncthreads();
                PD_RELANE();
                if (tid < 4 * QR) {       // four lanes per row; the row leaves for the head that owns it (before RoPE)
                    const float v = pd_rowsum<NB, 4>(part + (tid >> 2) * NBP, tid & 3) * inv;
                    if (!(tid & 3)) pd_publish<false>(P.gqkv + (size_t)layer * 3 * D + idx * QR + (tid >> 2), tag, v);
                }
                if (arole) {
                    // ---- the head's q | k | v of this position: 192 granules (a helper: q only); RoPE (go/model.go:449-477: pairs
                    //      (i, i + 32) = lanes l, l ^ 32 of a wavefront), KV store by the owner (go/model.go:552-554) ----
                    if (tid < (head ? 192 : 64)) {
                        const int sect = tid >> 6, e = tid & 63;
                        const __amdgpu_buffer_rsrc_t r = pd_rsrc(P.gqkv + (size_t)layer * 3 * D, 3u * D * 8u);
                        const unsigned off = (unsigned)((sect == 0 ? hidx * 64 : sect == 1 ? D + (hidx / GQ) * 64 : D + KVD + (hidx / GQ) * 64) + e) * 8u;
                        pd_u64 g;
                        for (int spins = 0;; spins++) {
                            g = pd_ld8(r, off);
                            if (__all((unsigned)(g >> 32) == tag)) break;
                            if (spins >= Q.spin_limit || ((spins & 63) == 63 && __hip_atomic_load(Q.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                                pd_give_up(Q, 32u, tid);
                                break;
                            }
                            // (no s_sleep: see pd_gather)
                        }
                        const float v = __uint_as_float((unsigned)g);
                        float o = v;
                        if (sect < 2) {
                            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
                            const float x0 = __uint_as_float(sw[0]), x1 = __uint_as_float(sw[1]);      // elements i and i + 32 of the head, in every lane
                            const int up = e >> 5;
                            if (!P.rope_conj) o = up ? (x0 * es + x1 * ec) : (x0 * ec - x1 * es);
                            else o = up ? (-x0 * es + x1 * ec) : (x0 * ec + x1 * es);
                        }
                        qs[tid] = o;
                        // (the first query head of a kv group stores the group's K / V row: every head of the group computed the same one)
                        if (sect == 1 && head && hidx % GQ == 0) kc[(long long)pos * 64 + e] = o;
                        if (sect == 2 && head && hidx % GQ == 0) vc[(long long)pos * 64 + e] = o;
                    }
                    __syncthreads();
                    PD_RELANE();
                    if (misc[2]) return;
                // ---- softmax attention over positions 0 .. pos (go/model.go:557-587), 128 positions per pass: nl_block.h's
                    //      one-barrier pass (every wavefront reduces its 16 positions to one (max, sum, sum p v) partial) ----
                    if (s == 0 && idx == 0) PD_ST(sb, 4);
                    int nci = 0;            // chunk records of this unit
                    for (int ch = chf; ch >= 0; ch -= PD_PARTS, nci++) {
                        const int t0 = ch * 128, n = min(128, pos + 1 - t0);
                        if (ch != chf) {    // (a later pass fetches its rows at its start: a prefetch a pass ahead costs 16 registers this kernel lacks)
#pragma unroll
                            for (int kk = 0; kk < 4; kk++) {
                                kreg[kk] = pd_ld16f(kr_, (unsigned)((t0 + min(wave * 16 + kr, n - 1)) * 16 + kq * 4 + kk) * 16u);
                                vreg[kk] = pd_ld16f(vr_, (unsigned)((t0 + min(wave * 16 + 4 * vg + kk, n - 1)) * 16 + vcl) * 16u);
                            }
                        }
                        const int krow = wave * 16 + kr;
                        const bool kcurrow = t0 + krow == pos;
                        float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll
                        for (int kk = 0; kk < 4; kk++) {
                            const float4 q4 = *reinterpret_cast<const float4 *>(qs + kq * 16 + kk * 4);
                            const float4 kc4 = *reinterpret_cast<const float4 *>(kcur + kq * 16 + kk * 4);
                            const float4 k4 = kcurrow ? kc4 : kreg[kk];
                            d0 = fmaf(q4.x, k4.x, d0); d1 = fmaf(q4.y, k4.y, d1); d2 = fmaf(q4.z, k4.z, d2); d3 = fmaf(q4.w, k4.w, d3);
                        }
                        const float sv = krow < n ? quad_sum((d0 + d1) + (d2 + d3)) * P.scale : -INFINITY;
                        const float mw = wave_max_f32(sv);
                        const float p = krow < n ? pd_exp(sv - (mw == -INFINITY ? 0.f : mw)) : 0.f;
                        const float lw = wave_sum_f32(kq == 0 ? p : 0.f);
                        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                        const float4 vc4 = *reinterpret_cast<const float4 *>(vcur + vcl * 4);
                        // the probabilities of this wavefront's 16 rows meet through LDS (one write, one 16-byte read: the lane's V rows
                        // 4 vg .. 4 vg + 3 are consecutive; the LDS serves a wavefront's operations in order)
                        if (kq == 0) wpart[wave * 68 + 48 + kr] = p;       // (columns 48 .. 63 of the wavefront's partial row: rewritten below)
                        const float4 pw4 = *reinterpret_cast<const float4 *>(wpart + wave * 68 + 48 + 4 * vg);
                        const float pw[4] = {pw4.x, pw4.y, pw4.z, pw4.w};
#pragma unroll
                        for (int kk = 0; kk < 4; kk++) {
                            const bool vcurrow = t0 + wave * 16 + 4 * vg + kk == pos;
                            const float4 v4 = vcurrow ? vc4 : vreg[kk];
                            o.x = fmaf(pw[kk], v4.x, o.x); o.y = fmaf(pw[kk], v4.y, o.y); o.z = fmaf(pw[kk], v4.z, o.z); o.w = fmaf(pw[kk], v4.w, o.w);
                        }
                        o.x = pd_rows4_sum(o.x); o.y = pd_rows4_sum(o.y); o.z = pd_rows4_sum(o.z); o.w = pd_rows4_sum(o.w);
                        if (lane < 16) *reinterpret_cast<float4 *>(wpart + wave * 68 + 4 + vcl * 4) = o;
                        if (lane == 0) { wpart[wave * 68] = mw; wpart[wave * 68 + 1] = lw; }
                        __syncthreads();
                        PD_RELANE();
                        if (wave == 0) {
                            const float mwv = lane < 8 ? wpart[min(lane, 7) * 68] : -INFINITY;
                            const float M = wave_max_f32(mwv);
                            const float wgt = (lane < 8 && mwv != -INFINITY) ? __expf(mwv - M) : 0.f;
                            const float Ls = wave_sum_f32(lane < 8 ? wgt * wpart[min(lane, 7) * 68 + 1] : 0.f);
                            float ov = 0.f;
#pragma unroll
                            for (int w = 0; w < 8; w++) ov = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(wgt), w)), wpart[w * 68 + 4 + lane], ov);
                            chunk[nci * 66 + 2 + lane] = ov;
                            if (lane == 0) { chunk[nci * 66] = M; chunk[nci * 66 + 1] = Ls; }
                        }
                        if (ch - PD_PARTS >= 0) __syncthreads();
                    }
                    // ---- the owner collects the helpers' partials (each the merge of that helper's passes) behind its own records ----
                    if (head && nparts > 1) {
                        const int np1 = nparts - 1;
                        if (tid < np1 * 66) {
                            const __amdgpu_buffer_rsrc_t r = pd_rsrc(P.gpart + ((size_t)layer * H + hidx) * (PD_PARTS - 1) * 66, (unsigned)((PD_PARTS - 1) * 66) * 8u);
                            pd_u64 g;
                            for (int spins = 0;; spins++) {
                                g = pd_ld8(r, (unsigned)tid * 8u);
                                if (__all((unsigned)(g >> 32) == tag || tid >= np1 * 66)) break;
                                if (spins >= Q.spin_limit || ((spins & 63) == 63 && __hip_atomic_load(Q.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                                    pd_give_up(Q, 4096u, tid);
                                    break;
                                }
                            }
                            chunk[nci * 66 + tid] = __uint_as_float((unsigned)g);       // (records nci .. nci + np1 - 1: the helpers' layout is the records')
                        }
                        nci += np1;
                    }
                    __syncthreads();
                    PD_RELANE();
                    if (misc[2]) return;
                    if (s == 0 && idx == 0) PD_ST(sb, 5);
                    if (tid < 64) {
                        // the records merged like position splits (fixed order: own passes from the highest down, then the helpers by part)
                        float M = chunk[0];
                        for (int c = 1; c < nci; c++) M = fmaxf(M, chunk[c * 66]);
                        float v = 0.f, Ls = 0.f;
                        if (nci == 1) { v = chunk[2 + tid]; Ls = chunk[1]; }
                        else
                            for (int c = 0; c < nci; c++) {
                                const float w = pd_exp(chunk[c * 66] - M);
                                Ls += w * chunk[c * 66 + 1];
                                v += w * chunk[c * 66 + 2 + tid];
                            }
                        if (head) pd_publish<false>(P.go + (size_t)layer * D + hidx * 64 + tid, tag, v * (1.0f / Ls));      // -> every unit of this XCD
                        else {      // a helper: its record for the owner
                            pd_u64 *dst = P.gpart + (((size_t)layer * H + hidx) * (PD_PARTS - 1) + (apart - 1)) * 66;
                            pd_publish<false>(dst + 2 + tid, tag, v);
                            if (tid == 0) { pd_publish<false>(dst, tag, M); pd_publish<false>(dst + 1, tag, Ls); }
                        }
                    }
                    if (s == 0 && idx == 0) PD_ST(sb, 6);
                }
            }
            {
                // =============================== every unit: WO, gate / up, down rows ===============================
                constexpr int nr = WR, nrp = WRP, rg = GR, rgp = GRP;
                const int r0 = idx * WR, g0 = idx * GR;
                // ---- o of every head -> LDS; WO rows + residual (go/model.go:590-594) ----
                {
                    float ov[2];
                    pd_gather<2>(P.go + (size_t)layer * D, D, tag, ov, Q, 2u, tid);
                    pd_limbs(ov[0], v0, e0, xl, xs, lane);
                    if (wave * 64 + PD_THREADS < D) pd_limbs(ov[1], v1, e1, xl, xs, lane);
                }
                __syncthreads();
                PD_RELANE();
                if (s == 0 && idx == 0) PD_ST(sb + 16, 2);
                if (misc[2]) return;
                pd_units<PD_UQ, PD_UW, NB>(wlo[s], whi[s], wsl + s * PD_UNITS * PD_THREADS, reinterpret_cast<const uint4 *>(xl), xs, nrp * NB, part, tid);
                __syncthreads();
                PD_RELANE();
                if (s == 0 && idx == 0) PD_ST(sb + 16, 3);
                if (tid < 4 * nr) {    // four lanes per row
                    const float v = pd_rowsum<NB, 4>(part + (tid >> 2) * NBP, tid & 3);
                    if (!(tid & 3)) pd_publish<false>(P.gxp + (size_t)layer * D + r0 + (tid >> 2), tag, xraw[r0 + (tid >> 2)] + v);
                }
                if (s == 0 && idx == 0) PD_ST(sb + 16, 4);
                // ---- x' of every worker; RMSNorm; gate and up rows, SiLU(gate) * up (go/model.go:597-606) ----
                const float gf0 = P.norms[(size_t)(layer * 2 + 1) * D + c0], gf1 = P.norms[(size_t)(layer * 2 + 1) * D + c1];
                float xp[2];
                pd_gather<2>(P.gxp + (size_t)layer * D, D, tag, xp, Q, 4u, tid);
                if (s == 0 && idx == 0) PD_ST(sb + 16, 5);
                // (no barrier: a wavefront is here only when ALL of x' has arrived -- this unit's rows too, published by lanes that read
                //  xraw and, through their quad, the row's block products first; the digit image of o was released by the barrier
                //  behind the WO units)
                PD_RELANE();
                if (v0) xraw[e0] = xp[0];
                if (v1) xraw[e1] = xp[1];
                pd_limbs(xp[0] * gf0, v0, e0, xl, xs, lane);
                if (wave * 64 + PD_THREADS < D) pd_limbs(xp[1] * gf1, v1, e1, xl, xs, lane);
                const float inv2 = pd_inv_rms(xp[0], xp[1], v0, v1, lane, wave, dred, D, P.eps);
                if (misc[2]) return;
                if (s == 0 && idx == 0) PD_ST(sb + 16, 6);
                pd_units<PD_UQ + PD_UW, PD_UG, NB>(wlo[s], whi[s], wsl + s * PD_UNITS * PD_THREADS, reinterpret_cast<const uint4 *>(xl), xs, 2 * rgp * NB, part, tid);
                __syncthreads();
                PD_RELANE();
                if (s == 0 && idx == 0) PD_ST(sb + 16, 7);
                // row sums and SiLU(gate) * up without a round trip through LDS: a quad = the gate row and the up row of one index,
                // two lanes each
                if ((tid >> 2) < rg) {
                    const int row = tid >> 2, mat = (tid >> 1) & 1;
                    const float own = pd_rowsum<NB, 2>(part + (mat * rgp + row) * NBP, tid & 1) * inv2;
                    const float other = dpp_f32<DPP_QUAD_XOR2>(own);
                    if (!(tid & 3)) {
                        const float g = own, u = other;
                        const float ex = pd_exp(-g);
                        pd_publish<false>(P.gh + (size_t)layer * I + g0 + row, tag, (g / (1.0f + ex)) * u);
                    }
                }
                if (s == 0 && idx == 0) PD_ST(sb + 16, 8);
                // ---- h of every worker -> LDS; down rows + residual (go/model.go:609-612) ----
                {
                    float hv[3];
                    pd_gather<3>(P.gh + (size_t)layer * I, I, tag, hv, Q, 8u, tid);
#pragma unroll
                    for (int k = 0; k < 3; k++)
                        if (wave * 64 + k * PD_THREADS < I) pd_limbs(hv[k], tid + k * PD_THREADS < I, tid + k * PD_THREADS, hl, hs, lane);
                }
                __syncthreads();
                PD_RELANE();
                if (s == 0 && idx == 0) PD_ST(sb + 16, 9);
                if (misc[2]) return;
                pd_units<PD_UQ + PD_UW + PD_UG, PD_UD, NBI>(wlo[s], whi[s], wsl + s * PD_UNITS * PD_THREADS, reinterpret_cast<const uint4 *>(hl), hs, nrp * NBI, part, tid);
                __syncthreads();
                PD_RELANE();
                if (s == 0 && idx == 0) PD_ST(sb + 16, 10);
                if (tid < 4 * nr) {      // four lanes per row, a quarter of the blocks each, joined in lane order
                    const int row = tid >> 2, q = tid & 3;
                    const float tot = pd_rowsum<NBI, 4>(part + row * NBIP, q);
                    if (q == 0) {
                        const float xo = xraw[r0 + row] + tot;
                        pd_u64 *dst = P.gx + (size_t)(layer + 1) * D + r0 + row;
                        if (next_here) pd_publish<false>(dst, tag, xo); else pd_publish<true>(dst, tag, xo);
                    }
                }
            }
            if (s == 0 && idx == 0) PD_ST(sb + 16, 11);
            // (no barrier between slots: whatever comes next starts with a gather of values that depend on this unit's published
            //  rows, i.e. on every LDS read of this slot)
            PD_RELANE();
        }

        // =================================== LM head (go/model.go:616-619), every compute unit ===================================
        {
            if (idx == 0) PD_ST(40, 0);
            const float go0 = P.norms[(size_t)(L * 2) * D + c0], go1 = P.norms[(size_t)(L * 2) * D + c1];
            float xv[2];
            pd_gather<2>(P.gx + (size_t)L * D, D, tag, xv, Q, 16u, tid);
            if (idx == 0) PD_ST(40, 1);
            pd_limbs(xv[0] * go0, v0, e0, xl, xs, lane);
            if (wave * 64 + PD_THREADS < D) pd_limbs(xv[1] * go1, v1, e1, xl, xs, lane);
            const float inv = pd_inv_rms(xv[0], xv[1], v0, v1, lane, wave, dred, D, P.eps);
            if (misc[2]) return;
            const int total = pd_pad4(lm_rows) * NB;
            pd_units_impl<PD_ULM, NB>([&](int k, uint4 &l, uint4 &h, unsigned &d) {
                l = lmw[(k * 2 + 0) * PD_THREADS + tid]; h = lmw[(k * 2 + 1) * PD_THREADS + tid]; d = lms[k * PD_THREADS + tid]; },
                reinterpret_cast<const uint4 *>(xl), xs, total, part, tid);
            __syncthreads();
            PD_RELANE();
            if (idx == 0) PD_ST(40, 2);
            float best = -INFINITY;
            int bidx = 0x7fffffff;
            if (tid < 256) {       // two lanes per row (both hold the row's logit)
                const int row = tid >> 1;
                const float lg = pd_rowsum<NB, 2>(part + min(row, 127) * NBP, tid & 1) * inv;
                if (row < lm_n) {
                    if (!(tid & 1)) {
                        P.logits[lm_r0 + row] = lg;
                        if (P.host_logits && (P.session || step == P.n_steps - 1))      // per-call Forward: no DMA behind the launch
                            __hip_atomic_store(P.host_logits + lm_r0 + row, lg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    if (lg > best) { best = lg; bidx = lm_r0 + row; }
                }
                pd_wave_argmax(best, bidx);              // (lane pairs = ascending rows)
                if (lane == 0) { bv[wave] = best; bi[wave] = bidx; }
            }
            __syncthreads();
            PD_RELANE();
            if (tid == 0) {
                best = bv[0]; bidx = bi[0];
#pragma unroll
                for (int w = 1; w < 4; w++)
                    if (bv[w] > best || (bv[w] == best && bi[w] < bidx)) { best = bv[w]; bidx = bi[w]; }
                const pd_u32x4 g = {tag, __float_as_uint(best), (unsigned)bidx, 0u};
                __builtin_amdgcn_raw_buffer_store_b128(g, pd_rsrc(P.gam, PD_GRID * 16u), cu * 16, 0, 16);     // write-through: layer 0's XCD reads it
            }
            __syncthreads();
            PD_RELANE();
        }
        if (idx == 0) PD_ST(40, 3);
        if (step == 0) PD_STAMP(2);
    }
    if (P0.dbg && cu == 0 && tid0 == 0) P0.dbg[3] = wall_clock64();
#undef PD_STAMP
#undef PD_ST
#undef PD_RELANE
#undef PD_ACK_HOST_ROWS
}

}  // namespace nl
