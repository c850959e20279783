// nl_p2p.h -- one-shot all-reduce / argmax exchange between the tensor-parallel ranks of one node.
//
// Where go/model.go:590-594 and :609-612 add WO*xb2 and Wdown*hb into the residual stream, a tensor-parallel
// rank holds only a partial [D] vector.  The 16 KiB all-reduce that completes it is latency, not bandwidth:
// 80 of them per token for the 7.9B tier.  Instead of a ring, every rank PUSHES its partial straight into a
// receive slot in every peer's memory (xGMI is point to point: one hop) and each rank sums the G slots itself
// in fixed rank order (deterministic, and bitwise what the in-process group of nl_group_forward computes).  The
// sum is done by the lane that produced the row, in the tail of the producing GEMV (EPI_P2P, nl_kernels.h): the
// peers run the same launch at the same time, so the wait is the wire's latency, and no reduce launch exists.
//
// Transport: 8-byte granules {tag, float bits}, each written by ONE system-scope store and polled with
// system-scope loads -- the data is its own flag, so no fence or separate flag orders anything
// (cdna_hip_programming.md Guideline 16 form R2, here across devices).  The receive area is allocated
// uncached and mapped into the peers with hipIpc* (one process per GPU).  tag = (forward counter << 8) | seam,
// the counter living on each device and advancing once per Forward, so a captured graph needs no per-launch
// argument.  Slots are double-buffered by seam parity: a rank can only be one seam ahead of its slowest peer.
// Every poll is bounded; a timeout sets a status word that the host turns into NL_ERR_COMM.
#pragma once
#include "nl_kernels.h"

namespace nl {

typedef unsigned long long u64;

__device__ __forceinline__ u64 granule_load(const u64 *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void granule_store(u64 *p, unsigned tag, unsigned bits) {
    __hip_atomic_store(p, ((u64)tag << 32) | bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct P2PArgmaxParams {
    ArgmaxParams A;        // this rank's partial maxima from its LM-head slice (indices local to the slice)
    EmbedParams E;         // E.x != nullptr: also embed the chosen token (next step of a multi-step graph)
    u64 *dst[8];           // peer r's argmax slots, already offset to this rank's entry
    const u64 *slots;      // this rank's receive entries: [G][4] granules {value, index, fused-launch status, -}
    unsigned *fstatus;     // this rank's fused-launch status word (nl_block.h; null without a fused plan): pushed with the
    unsigned *host_fstatus;//   pair, and set here (device word + its host-visible copy) when ANY rank reports a give-up, so
                           //   that every rank of the group retires its fused plan and redoes the call in the same step
    int G, row0;           // row0: global vocabulary index of this rank's first row
    unsigned seam;
    unsigned *epoch;       // advanced when E.x is set (the embedded token opens the next Forward)
    unsigned *status;
    long long timeout_ticks;
};

// Greedy argmax across the vocabulary shards (go/main.go:400-408: strict '>' => lowest index wins ties): local
// argmax of this rank's slice, one {value, index} pair pushed to every rank, every rank picks the same winner.
// The sender's LM-head kernel has completed before this kernel starts and a system-scope fence precedes the
// push, so a rank that sees the pair also sees that sender's logits slice in its gathered logits buffer.
__global__ void __launch_bounds__(1024) p2p_argmax_kernel(P2PArgmaxParams P) {
    __shared__ float bv[16];
    __shared__ int bi[16];
    __shared__ int tok;
    const int tid = threadIdx.x;
    float best = -INFINITY;
    int idx = 0x7fffffff;
    for (int i = tid; i < P.A.npart; i += blockDim.x) {
        float v = P.A.part_val[i];
        int vi = P.A.part_idx[i];
        if (v > best || (v == best && vi < idx)) { best = v; idx = vi; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(best, o);
        int oi = __shfl_xor(idx, o);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    if ((tid & 63) == 0) { bv[tid >> 6] = best; bi[tid >> 6] = idx; }
    __syncthreads();
    const unsigned tag = (__hip_atomic_load(P.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << 8) | P.seam;
    if (tid == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); w++)
            if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
        const int gidx = idx == 0x7fffffff ? 0x7fffffff : idx + P.row0;
        const unsigned fs = P.fstatus ? __hip_atomic_load(P.fstatus, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        __threadfence_system();
        for (int r = 0; r < P.G; r++) {
            granule_store(P.dst[r], tag, __float_as_uint(best));
            granule_store(P.dst[r] + 1, tag, (unsigned)gidx);
            granule_store(P.dst[r] + 2, tag, fs);
        }
    }
    // lanes 0..G-1 of wave 0 each wait for one rank's pair
    if (tid < 64) {
        const int r = min(tid, P.G - 1);
        const bool dead = __hip_atomic_load(P.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        u64 gv, gi, gs;
        const long long t0 = wall_clock64();
        bool got = true;
        for (;;) {
            gv = granule_load(P.slots + 4 * r);
            gi = granule_load(P.slots + 4 * r + 1);
            gs = granule_load(P.slots + 4 * r + 2);
            const bool ok = (unsigned)(gv >> 32) == tag && (unsigned)(gi >> 32) == tag && (unsigned)(gs >> 32) == tag;
            if (__all(ok)) break;
            if (dead || wall_clock64() - t0 > P.timeout_ticks) {
                if (tid == 0) atomicOr(P.status, 2u);
                got = false;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        float v = tid < P.G ? __uint_as_float((unsigned)gv) : -INFINITY;
        int vi = tid < P.G ? (int)(unsigned)gi : 0x7fffffff;
        unsigned fst = (tid < P.G && got) ? (unsigned)gs : 0u;
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) {   // G <= 8
            float ov = __shfl_xor(v, o);
            int oi = __shfl_xor(vi, o);
            if (ov > v || (ov == v && oi < vi)) { v = ov; vi = oi; }
            fst |= __shfl_xor(fst, o);
        }
        if (tid == 0 && fst && P.fstatus) { atomicOr(P.fstatus, fst); *P.host_fstatus = fst; }
        if (tid == 0) {
            if (vi == 0x7fffffff) vi = 0;   // all-NaN logits: the reference's loop never leaves index 0
            tok = vi;
            *P.A.result = vi;
            if (P.A.ctl[CTL_CHAIN]) {
                int step = P.A.ctl[CTL_STEP];
                P.A.ids[step] = vi;
                P.A.ctl[CTL_STEP] = step + 1;
                P.A.ctl[CTL_TOKEN] = vi;
                P.A.ctl[CTL_POS] = P.A.ctl[CTL_POS] + 1;
            }
            if (P.E.x) *P.epoch = *P.epoch + 1;
        }
    }
    if (!P.E.x) return;
    __syncthreads();
    const int token = tok;
    const int gr = P.E.gamma_row ? P.E.gamma_row[token] : -1;
    for (int i = tid; i < P.E.dim; i += blockDim.x) {
        float v = embed_value(P.E.table, P.E.wtype, P.E.dim, token, i);
        if (gr >= 0) v += P.E.gamma_val[(long long)gr * P.E.dim + i];
        P.E.x[i] = v;
    }
}

}  // namespace nl
