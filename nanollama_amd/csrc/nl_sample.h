// nl_sample.h -- on-device sampling: the non-greedy branch of Engine.Generate (go/main.go:174-195) without the
// per-token read-back of V logits.  Mirrors, in this order:
//   repetition penalty, in place, once per OCCURRENCE of a token in the recent window   go/main.go:177-187
//   temp <= 0                      -> argmax (lowest index wins ties)                      :297-299, :400-408
//   top_p < 1  -> sampleTopP: p_i = f32(exp(f64((l_i - max) / temp))), normalise, sort by p descending,
//                 cumulative sum until >= top_p, r = u * cumsum, first j with r <= cdf_j      :346-398
//   otherwise  -> sampleTopK: the top_k largest logits (earlier index wins ties), probabilities relative to
//                 the largest, r = u * sum, first i with r <= cdf_i                           :294-343
//   recent window append / drop-oldest                                                        :197-200
// The uniform u comes from the host's generator (one float32 per step, uploaded up front): Go's math/rand
// stream is not reproducible from another language anyway, the algorithm around it is what is mirrored.
// No library sort is involved (round 5): top-k finds its k-th largest logit by a 2-bit-per-pass selection on the keys' bits,
// collects the candidates above it (equal ones by ascending index, as the Go insertion list keeps them) and orders the <= 1024
// of them with a bitonic network in LDS (samp_topk_kernel); its sums run sequentially over the <= top_k terms exactly as the Go
// loop's.  top-p needs no order at all:
//
// Round 4: top-p needs NO sort (vocabularies up to 131072).  Both questions the Go loop asks of the sorted list -- where
// does the cumulative probability reach top_p, and where does it reach r = u * cumsum -- are weighted rank selections, and
// samp_select_radix_kernel answers them by radix selection on the bits of p (three histogram levels of 11 + 11 + 10 bits,
// each bucket holding the SUM of its candidates' weights) inside one workgroup that keeps the candidates in registers.
// Weights are exact integers, W_i = floor(p_i * 2^45), so sums do not depend on the order of addition (LDS atomics stay
// deterministic) and every cumulative value is the exact sum of the float32 p_i (to 2^-45 each), not a float32 chain:
//   cut  = first j (p descending, equal p by ascending id) with CDF_j >= ceil(top_p * TOTAL)
//   pick = first j with CDF_j >= max(1, ceil(u * CDF_cut))
// -- the Go conditions `cum >= topP` and `r <= cdf` on the unnormalised, exactly summed weights.  tests/sampling_mirror.py
// (device_top_p) restates it with Python integers and the device is held to it exactly; against the literal Go chain the
// picks are equal except for a u within float32 rounding of a cdf boundary, as before.
#pragma once
#include "nl_kernels.h"

namespace nl {

constexpr int SAMP_THREADS = 1024;

struct SampleParams {
    float *logits;
    int vocab;
    float temp, top_p;
    int top_k;
    float rep_penalty;
    int *recent;          // [rep_window] oldest -> newest
    int *recent_n;
    int rep_window;
    const float *uniforms;
    int *ctl, *ids;
    float *keys_in;       // [1024 * chunk] unnormalised probabilities of the top-p selection
    float *scal;          // (spare scalars)
    float *pmax;          // [nblocks] workgroup maxima of the penalised logits (samp_penalty_kernel)
    int nblocks;
    int radix;            // top-p by radix selection (samp_select_radix_kernel): keys_in = p and the level-1 histogram h1g
    unsigned long long *h1g;   // [2048] sums of candidate weights per level-1 bucket: zeroed by samp_penalty_kernel, filled by
                               // samp_prob_hist_kernel (all compute units), read by the selecting workgroup
    int nblocks_pen;      // workgroups of samp_penalty_kernel (pmax entries)
    int embed;            // the select launch also writes the picked token's embedding row (emb): the next Forward then starts
    EmbedParams emb;      //   at its first layer, as the chained greedy graph does with argmax_embed_kernel
};

// the embedding lookup that opens the next Forward (embed_kernel), by the workgroup that picked the token
__device__ __forceinline__ void samp_embed_tail(const SampleParams &P, int token) {
    if (!P.embed) return;
    const EmbedParams &E = P.emb;
    if (E.epoch && threadIdx.x == 0) *E.epoch = *E.epoch + 1;            // the forward counter, as embed_kernel advances it
    const int gr = E.gamma_row ? E.gamma_row[token] : -1;
    for (int i = threadIdx.x; i < E.dim; i += blockDim.x) {
        float v = embed_value(E.table, E.wtype, E.dim, token, i);
        if (gr >= 0) v += E.gamma_val[(long long)gr * E.dim + i];
        E.x[i] = v;
    }
}

// device scratch of one sampler (one vocabulary)
struct SampScratch {
    float *keys_in = nullptr, *scal = nullptr, *pmax = nullptr, *uniforms = nullptr;
    unsigned long long *h1g = nullptr;
    int *recent = nullptr, *recent_n = nullptr;
};

// vocab / 256 workgroups: repetition penalty on the workgroup's own 256 logits (every workgroup reads the window and
// counts the occurrences of its own ids: the penalty of a token depends on its own logit and its count only, and x / p,
// x * p keep the sign, so "once per occurrence in window order" is "count times"), then the workgroup's maximum
__global__ void __launch_bounds__(256) samp_penalty_kernel(SampleParams P) {
    __shared__ float red[4];
    __shared__ int win[SAMP_THREADS];
    const int tid = threadIdx.x, i = blockIdx.x * 256 + tid;
    const int n = sload_i32(P.recent_n);
    float v = P.logits[min(i, P.vocab - 1)];
    if (P.radix)      // the level-1 histogram of this step's selection (filled by the next launch)
        for (int j = i; j < 2048; j += (int)gridDim.x * 256) P.h1g[j] = 0;
    if (P.rep_penalty > 1.0f && n > 0) {
        for (int s = tid; s < n; s += 256) win[s] = P.recent[s];
        __syncthreads();
        int count = 0;
        for (int s = 0; s < n; s++) count += win[s] == i ? 1 : 0;
        if (count && i < P.vocab) {
            for (int c = 0; c < count; c++) v = v > 0.f ? v / P.rep_penalty : v * P.rep_penalty;
            P.logits[i] = v;
        }
    }
    float m = i < P.vocab ? v : -INFINITY;
    m = wave_max_f32(m);
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) P.pmax[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// The top-p path's probabilities p_i = f32(exp(f64((l_i - max) / temp))), plus level 1 of the selection's histograms, built HERE by
// every compute unit instead of by the one selecting workgroup (3.8 us of its 20): a candidate (p >= 2^-45, weight >= 1) goes
// to bucket (bits(p) - bits(2^-45)) >> 18 -- 32 buckets per binary exponent, 1440 in all -- with its exact integer weight;
// a workgroup sums its 1024 candidates in LDS and flushes the non-empty buckets with 64-bit device-scope atomics (integer sums:
// the order does not matter; 32 workgroups x ~150 buckets cost 0.2 us, tools/atomic_probe.hip).
constexpr unsigned SAMP_KEY_MIN = 82u << 23;                 // bits of 2^-45: the smallest p of weight 1
constexpr int SAMP_S1 = 18, SAMP_S2 = 7;                     // level 1: bits 28..18 of key - SAMP_KEY_MIN, level 2: 17..7, level 3: 6..0
__device__ __forceinline__ unsigned long long samp_weight_of(unsigned k) {   // floor(p 2^45) = (m 2^22) >> (127 - e), k >= SAMP_KEY_MIN
    const unsigned long long m = (unsigned long long)((k & 0x7fffffu) | 0x800000u) << 22;
    return m >> (127u - (k >> 23));
}
__global__ void __launch_bounds__(SAMP_THREADS) samp_prob_hist_kernel(SampleParams P) {
    __shared__ float red[SAMP_THREADS / 64];
    __shared__ unsigned long long hl[2048];
    const int tid = threadIdx.x, i = blockIdx.x * SAMP_THREADS + tid;
    const float l = P.logits[min(i, P.vocab - 1)];
    hl[tid] = 0; hl[tid + 1024] = 0;
    float gmax = tid < P.nblocks_pen ? P.pmax[tid] : -INFINITY;          // (<= 512 workgroup maxima)
    gmax = wave_max_f32(gmax);
    if ((tid & 63) == 0) red[tid >> 6] = gmax;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < SAMP_THREADS / 64; w++) gmax = fmaxf(gmax, red[w]);
    if (i < P.vocab) {
        const float key = (float)exp((double)((l - gmax) / P.temp));
        P.keys_in[i] = key;
        const unsigned kb = __float_as_uint(key);
        if (kb >= SAMP_KEY_MIN) atomicAdd(&hl[(kb - SAMP_KEY_MIN) >> SAMP_S1], samp_weight_of(kb));
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const unsigned long long v = hl[tid + q * 1024];
        if (v) atomicAdd(&P.h1g[tid + q * 1024], v);
    }
}

// every thread of the workgroup visits the V 32-bit words of an L2-resident array: coalesced 16-byte agent-scope loads (through the
// L2 only: 60-110 GB/s per compute unit against 37 for plain loads, tools/ingest_probe.hip), eight in flight per lane
__device__ __forceinline__ uint4 samp_ld_l2(const unsigned *base, unsigned word_off) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(base), 0, -1, 0x00020000);
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const v4u t = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(word_off * 4u), 0, 16 /* sc1 */);
    return make_uint4(t.x, t.y, t.z, t.w);
}
template <class F>
__device__ __forceinline__ void samp_each_key(const unsigned *kin, int V, F f) {
    const int nq = (V + 3) >> 2, tid = threadIdx.x;
    for (int q0 = 0; q0 < nq; q0 += 8 * SAMP_THREADS) {
        uint4 v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = samp_ld_l2(kin, (unsigned)min(q0 + j * SAMP_THREADS + tid, nq - 1) * 4u);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int q = q0 + j * SAMP_THREADS + tid, i = q * 4;
            if (q < nq) {
                f(v[j].x, i);
                if (i + 1 < V) f(v[j].y, i + 1);
                if (i + 2 < V) f(v[j].z, i + 2);
                if (i + 3 < V) f(v[j].w, i + 3);
            }
        }
    }
}

// the end of every selecting launch: embedding row of the pick (opens the next Forward), recent window append / drop-oldest
// (go/main.go:197-200), decode state.  Called by all SAMP_THREADS threads of the one workgroup.
__device__ __forceinline__ void samp_finish(const SampleParams &P, int pick, int step) {
    const int tid = threadIdx.x;
    samp_embed_tail(P, pick);
    const int n = *P.recent_n;
    int shifted = 0;
    if (P.rep_window > 0 && n >= P.rep_window && tid + 1 < n) shifted = P.recent[tid + 1];   // rep_window <= 1024
    __syncthreads();
    if (P.rep_window > 0) {
        if (n >= P.rep_window) {
            if (tid + 1 < n) P.recent[tid] = shifted;
            if (tid == 0) P.recent[n - 1] = pick;
        } else if (tid == 0) {
            P.recent[n] = pick;
            *P.recent_n = n + 1;
        }
    }
    if (tid == 0) {
        P.ids[step] = pick;
        P.ctl[CTL_STEP] = step + 1;
        P.ctl[CTL_TOKEN] = pick;
        P.ctl[CTL_POS] = P.ctl[CTL_POS] + 1;
    }
}

// temp <= 0 with a repetition penalty: argmax over the penalised logits, lowest index wins ties (go/main.go:297-299, :400-408)
__global__ void __launch_bounds__(SAMP_THREADS) samp_argmax_select_kernel(SampleParams P) {
    __shared__ float bval[SAMP_THREADS / 64];
    __shared__ int bidx[SAMP_THREADS / 64];
    __shared__ int s_pick;
    const int tid = threadIdx.x, V = P.vocab;
    const int step = P.ctl[CTL_STEP];
    float best = -INFINITY;
    int idx = 0x7fffffff;
    for (int i = tid; i < V; i += SAMP_THREADS) {
        const float v = P.logits[i];
        if (v > best || idx == 0x7fffffff) { best = v; idx = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o);
        const int oi = __shfl_xor(idx, o);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    if ((tid & 63) == 0) { bval[tid >> 6] = best; bidx[tid >> 6] = idx; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < SAMP_THREADS / 64; w++)
            if (bval[w] > best || (bval[w] == best && bidx[w] < idx)) { best = bval[w]; idx = bidx[w]; }
        s_pick = idx == 0x7fffffff ? 0 : idx;
    }
    __syncthreads();
    samp_finish(P, s_pick, step);
}

// ---- sampleTopK (go/main.go:294-343) without a sort of the vocabulary ----------------------------------------------------------
// The Go loop keeps an insertion list of the top_k largest logits (strict '>': of equal logits the earlier index stays ahead) and
// walks it.  Here, one workgroup:
//   1. the logits become order-preserving 32-bit keys (a NaN never enters the Go list: lowest key);
//   2. the k-th largest key T is found two bits at a time, sixteen passes: every lane counts, among its keys that match the
//      bits fixed so far, those whose next digit is >= 1, >= 2, = 3 (three counters in ONE 64-bit sum, 21 bits each: <= 131072
//      keys), the counts meet by DPP + LDS, and the digit where the cumulative count from the top reaches what is still needed
//      is fixed -- no atomics, nothing data-dependent in the control flow.  Vocabularies up to 32768 keep their keys in
//      registers; larger ones stream them out of L2 each pass (agent-scope 16-byte loads, samp_each_key);
//   3. candidates = every key > T and, of the keys == T, the first `need` in index order (two block scans: ranks of the equal
//      keys, then list positions) as (key, ~index) pairs in LDS;
//   4. a bitonic network orders the <= 1024 pairs descending (equal keys by ascending index) -- only next_pow2(top_k) of them;
//   5. probabilities, their sum and the cdf walk exactly as the Go loop: one lane, left to right.
// top_k <= 1024 (check_sample_params; the reference's default is 50).
typedef unsigned long long samp_u64_t;
__device__ __forceinline__ unsigned samp_key_of(float v) {
    const unsigned b = __float_as_uint(v);
    if (v != v) return 0u;
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float samp_val_of(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

// sum of a 64-bit value over the workgroup, result in every thread (two barriers; scr: 16 entries)
__device__ __forceinline__ samp_u64_t samp_block_sum_u64(samp_u64_t v, samp_u64_t *scr) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)v, o), hi = __shfl_xor((unsigned)(v >> 32), o);
        v += ((samp_u64_t)hi << 32) | lo;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scr[threadIdx.x >> 6] = v;
    __syncthreads();
    samp_u64_t t = 0;
#pragma unroll
    for (int w = 0; w < SAMP_THREADS / 64; w++) t += scr[w];
    return t;
}
// exclusive prefix of a per-thread count over the workgroup (thread order) and the total
__device__ __forceinline__ unsigned samp_block_excl_scan(unsigned c, unsigned *wtot, unsigned &total) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    unsigned incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    __syncthreads();
    if (lane == 63) wtot[wv] = incl;
    __syncthreads();
    unsigned pre = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SAMP_THREADS / 64; w++) { const unsigned t = wtot[w]; if (w < wv) pre += t; tot += t; }
    total = tot;
    return pre + incl - c;
}

template <bool STREAM>     // false: vocab <= 32768, 32 keys per lane in registers; true: keys streamed out of L2 every pass
__global__ void __launch_bounds__(SAMP_THREADS) samp_topk_kernel(SampleParams P) {
    constexpr int C = 32;
    __shared__ samp_u64_t list[SAMP_THREADS];
    __shared__ samp_u64_t scr[SAMP_THREADS / 64];
    __shared__ unsigned wtot[SAMP_THREADS / 64];
    __shared__ float prob[SAMP_THREADS];
    __shared__ int s_pick;
    const int tid = threadIdx.x, V = P.vocab;
    const int step = P.ctl[CTL_STEP];
    const float u = P.uniforms[step];
    const int K = min(min(P.top_k, V), SAMP_THREADS);
    const unsigned *lg = reinterpret_cast<const unsigned *>(P.logits);
    unsigned key[STREAM ? 1 : C];
    const int lo = tid * C;
    if (!STREAM) {
#pragma unroll
        for (int k = 0; k < C; k++) key[k] = lo + k < V ? samp_key_of(P.logits[min(lo + k, V - 1)]) : 0u;     // (past the vocabulary: the lowest key, never needed: K <= V)
    }
    list[tid] = 0;
    // ---- 2. the k-th largest key, two bits per pass ----
    unsigned prefix = 0, need = (unsigned)K;
    for (int sft = 30; sft >= 0; sft -= 2) {
        samp_u64_t cnt = 0;
        auto tally = [&](unsigned k) {
            const bool m = sft == 30 || (k >> (sft + 2)) == prefix;
            const unsigned d = (k >> sft) & 3u;
            cnt += m ? ((samp_u64_t)(d >= 1u) | ((samp_u64_t)(d >= 2u) << 21) | ((samp_u64_t)(d == 3u) << 42)) : 0ull;
        };
        if (STREAM) samp_each_key(lg, V, [&](unsigned raw, int) { tally(samp_key_of(__uint_as_float(raw))); });
        else {
#pragma unroll
            for (int k = 0; k < C; k++) if (lo + k < V) tally(key[k]);
        }
        const samp_u64_t tot = samp_block_sum_u64(cnt, scr);
        const unsigned c1 = (unsigned)(tot & 0x1fffffu), c2 = (unsigned)((tot >> 21) & 0x1fffffu), c3 = (unsigned)(tot >> 42);
        unsigned d;
        if (c3 >= need) d = 3u;
        else if (c2 >= need) { d = 2u; need -= c3; }
        else if (c1 >= need) { d = 1u; need -= c2; }
        else { d = 0u; need -= c1; }
        prefix = (prefix << 2) | d;
    }
    const unsigned T = prefix;        // need = how many of the keys == T belong to the top K (>= 1)
    // ---- 3. candidates -> list, by two scans ----
    unsigned n_eq = 0, n_gt = 0;
    if (STREAM) samp_each_key(lg, V, [&](unsigned raw, int) { const unsigned k = samp_key_of(__uint_as_float(raw)); n_eq += k == T; n_gt += k > T; });
    else {
#pragma unroll
        for (int k = 0; k < C; k++) if (lo + k < V) { n_eq += key[k] == T; n_gt += key[k] > T; }
    }
    // (streamed keys visit a lane in index order within each of its 16-byte loads but the lanes interleave: ranks of equal keys are
    //  taken in the lane-major order below for the register form, and by index through a second counting pass for the streamed one)
    unsigned tot_eq, tot_take;
    unsigned eq_before = samp_block_excl_scan(n_eq, wtot, tot_eq);
    if (!STREAM) {
        const unsigned take_eq = eq_before >= need ? 0u : min(n_eq, need - eq_before);
        unsigned posn = samp_block_excl_scan(n_gt + take_eq, wtot, tot_take);
        unsigned eq_seen = 0;
#pragma unroll
        for (int k = 0; k < C; k++) {
            if (lo + k < V) {
                const bool gt = key[k] > T, eq = key[k] == T;
                const bool take = gt || (eq && eq_before + eq_seen < need);
                eq_seen += eq;
                if (take) { list[min(posn, (unsigned)SAMP_THREADS - 1)] = ((samp_u64_t)key[k] << 32) | (0xffffffffu - (unsigned)(lo + k)); posn++; }
            }
        }
    } else {
        // streamed: lane order is not index order, so the equal keys are ranked by INDEX: a key == T at index i is taken when fewer
        // than `need` equal keys have a smaller index -- counted exactly by a pass per candidate would be quadratic; instead the
        // equal keys' indices go through the same 2-bit selection (the need-th smallest index among them), below
        unsigned iprefix = 0, ineed = need;         // the ineed-th SMALLEST index among keys == T: select on ~index, largest first
        for (int sft = 30; sft >= 0; sft -= 2) {
            samp_u64_t cnt = 0;
            samp_each_key(lg, V, [&](unsigned raw, int i) {
                if (samp_key_of(__uint_as_float(raw)) != T) return;
                const unsigned k = 0xffffffffu - (unsigned)i;
                const bool m = sft == 30 || (k >> (sft + 2)) == iprefix;
                const unsigned d = (k >> sft) & 3u;
                cnt += m ? ((samp_u64_t)(d >= 1u) | ((samp_u64_t)(d >= 2u) << 21) | ((samp_u64_t)(d == 3u) << 42)) : 0ull;
            });
            const samp_u64_t tot = samp_block_sum_u64(cnt, scr);
            const unsigned c1 = (unsigned)(tot & 0x1fffffu), c2 = (unsigned)((tot >> 21) & 0x1fffffu), c3 = (unsigned)(tot >> 42);
            unsigned d;
            if (c3 >= ineed) d = 3u;
            else if (c2 >= ineed) { d = 2u; ineed -= c3; }
            else if (c1 >= ineed) { d = 1u; ineed -= c2; }
            else { d = 0u; ineed -= c1; }
            iprefix = (iprefix << 2) | d;
        }
        const unsigned last_idx = 0xffffffffu - iprefix;      // equal keys with index <= last_idx are taken
        unsigned n_take = 0;
        samp_each_key(lg, V, [&](unsigned raw, int i) { const unsigned k = samp_key_of(__uint_as_float(raw)); n_take += k > T || (k == T && (unsigned)i <= last_idx); });
        unsigned posn = samp_block_excl_scan(n_take, wtot, tot_take);
        samp_each_key(lg, V, [&](unsigned raw, int i) {
            const unsigned k = samp_key_of(__uint_as_float(raw));
            if (k > T || (k == T && (unsigned)i <= last_idx)) { list[min(posn, (unsigned)SAMP_THREADS - 1)] = ((samp_u64_t)k << 32) | (0xffffffffu - (unsigned)i); posn++; }
        });
    }
    __syncthreads();
    // ---- 4. bitonic network, descending, over the first P2 = next_pow2(K) entries (the rest are 0 = below every candidate) ----
    int P2 = 1;
    while (P2 < K) P2 <<= 1;
    for (int size = 2; size <= P2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (tid < (P2 >> 1)) {
                const int i = ((tid & ~(stride - 1)) << 1) | (tid & (stride - 1)), j = i | stride;
                const bool desc = (i & size) == 0;
                const samp_u64_t a = list[i], b = list[j];
                if ((a < b) == desc) { list[i] = b; list[j] = a; }
            }
            __syncthreads();
        }
    }
    // ---- 5. the Go loop (go/main.go:325-342): probabilities relative to the largest, sequential sums on one lane ----
    const float v0 = samp_val_of((unsigned)(list[0] >> 32));
    if (tid < K) prob[tid] = (float)exp((double)((samp_val_of((unsigned)(list[tid] >> 32)) - v0) / P.temp));
    __syncthreads();
    if (tid == 0) {
        float sum = 0.f;
        for (int i = 0; i < K; i++) sum += prob[i];
        const float r = u * sum;
        float cdf = 0.f;
        int sel = 0;
        for (int i = 0; i < K; i++) {
            cdf += prob[i];
            if (r <= cdf) { sel = i; break; }
        }
        s_pick = (int)(0xffffffffu - (unsigned)list[sel]);
    }
    __syncthreads();
    samp_finish(P, s_pick, step);
}

// ---- top-p without a sort: weighted radix selection in one workgroup -----------------------------------------------
typedef unsigned long long samp_u64;

// exact ceil(f * x) for a float32 0 <= f <= 1 (f = m * 2^(e - 150), m < 2^24; m * x < 2^88)
__device__ __forceinline__ samp_u64 samp_ceil_mul(float f, samp_u64 x) {
    const unsigned bits = __float_as_uint(f);
    int e = (int)((bits >> 23) & 0xffu);
    unsigned m = bits & 0x7fffffu;
    if (e == 0) { if (m == 0) return 0; e = 1; } else m |= 0x800000u;
    const int s = 150 - e;                                  // >= 23 for f <= 1
    const unsigned __int128 prod = (unsigned __int128)m * x;
    if (s >= 100) return prod ? 1 : 0;
    return (samp_u64)((prod + ((((unsigned __int128)1) << s) - 1)) >> s);
}

// weight of a candidate: floor(p * 2^45) (p <= 1: at most 2^45; 131072 of them stay below 2^63), from the bits of p --
// p = m * 2^(e - 150), so the weight is m shifted by e - 105 (integer shifts: a float64 product costs ten times as much
// on the one compute unit that runs the selection)
__device__ __forceinline__ samp_u64 samp_weight(unsigned key) {
    const int e = (int)((key >> 23) & 0xffu);
    const samp_u64 m = (samp_u64)((key & 0x7fffffu) | 0x800000u);
    const int sh = e - 105;                                  // (e = 0, zero / denormal p: far below 2^-45, weight 0)
    return sh >= 0 ? m << (sh & 63) : (sh > -24 ? m >> ((-sh) & 63) : 0ull);
}

__device__ __forceinline__ samp_u64 samp_shfl_up_u64(samp_u64 v, int d) {
    const unsigned lo = __shfl_up((unsigned)v, d), hi = __shfl_up((unsigned)(v >> 32), d);
    return ((samp_u64)hi << 32) | lo;
}

// inclusive prefix sum over the 64 lanes on DPP (four in-row shifts, two cross-row broadcasts; hipcc lowers __shfl_up to
// ds_bpermute, ~130 cycles a hop and twelve hops for a 64-bit scan)
template <int CTRL, int ROWMASK, bool BOUND>
__device__ __forceinline__ samp_u64 samp_dpp_u64(samp_u64 v) {
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, ROWMASK, 0xF, BOUND);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, ROWMASK, 0xF, BOUND);
    return ((samp_u64)(unsigned)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ samp_u64 samp_wave_scan_u64(samp_u64 v) {
    v += samp_dpp_u64<0x111, 0xF, true>(v);                 // row_shr:1 (lanes shifted in from outside the row: 0)
    v += samp_dpp_u64<0x112, 0xF, true>(v);
    v += samp_dpp_u64<0x114, 0xF, true>(v);
    v += samp_dpp_u64<0x118, 0xF, true>(v);
    v += samp_dpp_u64<DPP_ROW_BCAST15, 0xA, false>(v);      // rows 1, 3 += lane 15 of rows 0, 2
    v += samp_dpp_u64<DPP_ROW_BCAST31, 0xC, false>(v);      // rows 2, 3 += lane 31
    return v;
}

// Descending weighted selection over the 2048 buckets of h (bucket 2047 first): the bucket in which the running sum first
// reaches x = x_of(total) (1 <= x <= total), and what is left of x inside it.  The whole workgroup calls it; four
// wavefronts (one per SIMD) do the work, eight buckets per lane -- the selection runs on ONE compute unit and is bound by
// its vector issue slots, so work that a quarter of the wavefronts can do is not spread over all sixteen.  scr: 2 x 8 words
// of LDS used alternately (par), so that no barrier is needed behind the result read.
struct SampSel { int bucket; samp_u64 rem, x, total; };
template <typename XF>
__device__ __forceinline__ SampSel samp_select_bucket(const samp_u64 *h, XF x_of, samp_u64 *scr, int &par) {
    const int tid = threadIdx.x, ln = tid & 63, wv = tid >> 6;
    samp_u64 *wtot = scr + par * 8, *res = scr + par * 8 + 4;
    par ^= 1;
    samp_u64 v[8], inc = 0;
    if (tid < 256) {
        const uint4 *src = reinterpret_cast<const uint4 *>(h + 2040 - 8 * tid);      // buckets 2040 - 8 tid .. 2047 - 8 tid
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint4 t = src[3 - q];
            v[2 * q] = ((samp_u64)t.w << 32) | t.z;         // descending: the higher bucket of the pair first
            v[2 * q + 1] = ((samp_u64)t.y << 32) | t.x;
        }
        samp_u64 local = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) local += v[q];
        inc = samp_wave_scan_u64(local);
        if (ln == 63) wtot[wv] = inc;
        inc -= local;                                        // exclusive, within the wavefront
    }
    __syncthreads();
    if (tid < 256) {
        const samp_u64 t0 = wtot[0], t1 = wtot[1], t2 = wtot[2], t3 = wtot[3];
        const samp_u64 total = (t0 + t1) + (t2 + t3);
        const samp_u64 x = x_of(total);
        samp_u64 run = inc + (wv > 0 ? t0 : 0) + (wv > 1 ? t1 : 0) + (wv > 2 ? t2 : 0);
#pragma unroll
        for (int q = 0; q < 8; q++) {
            if (run < x && x <= run + v[q]) { res[0] = (samp_u64)(2047 - 8 * tid - q); res[1] = x - run; res[2] = x; res[3] = total; }
            run += v[q];
        }
    }
    __syncthreads();
    return SampSel{(int)res[0], res[1], res[2], res[3]};
}

// 1 workgroup: C candidates per thread in registers (1024 * C >= vocab), the cut, the pick, the decode state.
// Candidates = the entries of weight >= 1 (p >= 2^-45); keys are taken relative to bits(2^-45).  Level 1 of the histograms
// (2048 buckets of 2^18 key values: 32 per binary exponent) arrives ready-made from samp_prob_hist_kernel; a selection scans it,
// copies the candidates of the crossing bucket into an LDS list (one compare per candidate) and runs levels 2 (11 bits) and 3
// (7 bits) on that list -- or, when the bucket holds more than the list (thousands of candidates with one p), on the registers.
// (Until the histogram moved out, this workgroup also took the range of the keys and built level 1 relative to the smallest
// candidate -- a flat distribution's keys share three exponents and LDS atomics on one address are served one at a time:
// 0.6 + 3.8 us of the launch's 20.)
#ifdef NL_SAMP_STAMPS   // developer build (tools/samp_probe.hip): shader-clock stamps of thread 0
__device__ unsigned long long g_samp_stamps[32];
#define SAMP_STAMP(i) do { if (threadIdx.x == 0) g_samp_stamps[i] = wall_clock64(); } while (0)
#else
#define SAMP_STAMP(i) do {} while (0)
#endif
constexpr int SAMP_LIST_CAP = 8192;
template <int C>
__global__ void __launch_bounds__(SAMP_THREADS) samp_select_radix_kernel(SampleParams P) {
    __shared__ samp_u64 hw[2048];
    __shared__ unsigned list[SAMP_LIST_CAP];
    __shared__ samp_u64 scr[16];
    __shared__ unsigned wcnt[SAMP_THREADS / 64];
    __shared__ unsigned list_n;
    __shared__ int s_pick;
    const int tid = threadIdx.x, V = P.vocab, lo = tid * C;
    const int step = P.ctl[CTL_STEP];
    const float u = P.uniforms[step];
    unsigned key[C];                                        // bits of p >= 0: unsigned order = float order
    const unsigned *kin = reinterpret_cast<const unsigned *>(P.keys_in);
    SAMP_STAMP(0);
#pragma unroll
    for (int k = 0; k < C; k += 4) {                        // (keys_in holds 1024 * C entries; the tail past vocab is masked)
        const uint4 t4 = *reinterpret_cast<const uint4 *>(kin + lo + k);
        key[k] = lo + k < V ? t4.x : 0u; key[k + 1] = lo + k + 1 < V ? t4.y : 0u;
        key[k + 2] = lo + k + 2 < V ? t4.z : 0u; key[k + 3] = lo + k + 3 < V ? t4.w : 0u;
    }
    if (tid == 0) s_pick = 0x7fffffff;
    int par = 0;
    SAMP_STAMP(1);
    // level 1 comes from samp_prob_hist_kernel (global memory, built by every compute unit); keys relative to SAMP_KEY_MIN
    constexpr unsigned kmin = SAMP_KEY_MIN;
    constexpr int s1 = SAMP_S1, s2 = SAMP_S2;
    const samp_u64 *const h1 = P.h1g;
    auto weight_of = [](unsigned k) { return samp_weight_of(k); };
    SAMP_STAMP(2);
    SAMP_STAMP(3);
    [[maybe_unused]] int stamp_base = 4;
    // one weighted selection: up to three levels; returns the relative key of the crossing candidates and what is left of x
    // among them
    auto select = [&](auto x_of, unsigned &dout, samp_u64 &rem, samp_u64 &x) {
        const SampSel a = samp_select_bucket(h1, x_of, scr, par);
        unsigned prefix = (unsigned)a.bucket;               // = d >> s1 of the crossing candidates
        rem = a.rem;
        x = a.x;
        SAMP_STAMP(stamp_base + 0);
        // the candidates of that bucket -> list (relative keys).  (Counting first and reserving one run per thread was
        // measured slower: the second walk over the registers costs more than the atomics it saves)
        if (tid == 0) list_n = 0;
        hw[tid] = 0; hw[tid + 1024] = 0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < C; k++) {
            const unsigned d = key[k] - kmin;               // (a non-candidate wraps to >= 2^32 - 2^30: its d >> s1 is no bucket)
            if ((d >> s1) == prefix) {
                const unsigned at = atomicAdd(&list_n, 1u);
                if (at < (unsigned)SAMP_LIST_CAP) list[at] = d;
            }
        }
        __syncthreads();
        SAMP_STAMP(stamp_base + 1);
        const unsigned n = list_n;
        const bool listed = n <= (unsigned)SAMP_LIST_CAP;   // (else: the lower levels walk the registers again)
        auto same = [](samp_u64 r) { return [r](samp_u64) { return r; }; };
        {
            const unsigned mask = (1u << (s1 - s2)) - 1u;
            if (listed) {
                for (unsigned i = tid; i < n; i += SAMP_THREADS) { const unsigned d = list[i]; atomicAdd(&hw[(d >> s2) & mask], weight_of(d + kmin)); }
            } else {
#pragma unroll
                for (int k = 0; k < C; k++) {
                    unsigned kk = key[k];
                    asm volatile("" : "+v"(kk));
                    const unsigned d = kk - kmin;
                    if ((d >> s1) == prefix) atomicAdd(&hw[(d >> s2) & mask], weight_of(kk));
                }
            }
            __syncthreads();
            const SampSel b = samp_select_bucket(hw, same(rem), scr, par);
            prefix = (prefix << (s1 - s2)) | (unsigned)b.bucket;
            rem = b.rem;
            SAMP_STAMP(stamp_base + 2);
        }
        if (s2 > 0) {
            hw[tid] = 0; hw[tid + 1024] = 0;
            __syncthreads();
            const unsigned mask = (1u << s2) - 1u;
            if (listed) {
                for (unsigned i = tid; i < n; i += SAMP_THREADS) { const unsigned d = list[i]; if ((d >> s2) == prefix) atomicAdd(&hw[d & mask], weight_of(d + kmin)); }
            } else {
#pragma unroll
                for (int k = 0; k < C; k++) {
                    unsigned kk = key[k];
                    asm volatile("" : "+v"(kk));
                    const unsigned d = kk - kmin;
                    if ((d >> s2) == prefix) atomicAdd(&hw[d & mask], weight_of(kk));
                }
            }
            __syncthreads();
            const SampSel c = samp_select_bucket(hw, same(rem), scr, par);
            prefix = (prefix << s2) | (unsigned)c.bucket;
            rem = c.rem;
            SAMP_STAMP(stamp_base + 3);
        }
        dout = prefix;
    };
    // ties: as many of the equal candidates as x needs (one, unless several candidates share the crossing p)
    auto tie_rank = [](samp_u64 rem, samp_u64 w) { return rem <= w ? (samp_u64)1 : (rem + w - 1) / w; };
    unsigned dc, dp;
    samp_u64 rem, xcut, xr;
    const float top_p = P.top_p;
    select([top_p](samp_u64 total) { const samp_u64 x = samp_ceil_mul(top_p, total); return x < 1 ? (samp_u64)1 : x; }, dc, rem, xcut);
    stamp_base = 8;
    const samp_u64 wc = weight_of(dc + kmin);
    const samp_u64 cum = xcut - rem + tie_rank(rem, wc) * wc;            // CDF of the cut candidate
    select([u, cum](samp_u64) { const samp_u64 x = samp_ceil_mul(u, cum); return x < 1 ? (samp_u64)1 : x; }, dp, rem, xr);
    SAMP_STAMP(12);
    const unsigned rank = (unsigned)tie_rank(rem, weight_of(dp + kmin));   // 1-based among the candidates with this p, by id
    // the rank-th candidate with relative key dp in ascending id (thread chunks are contiguous in id)
    const unsigned kpick = dp + kmin;
    if (rank == 1) {
        // (the usual case: the first candidate with this p = the smallest id)
        int first = 0x7fffffff;
#pragma unroll
        for (int k = C - 1; k >= 0; k--) first = key[k] == kpick ? lo + k : first;
        if (first != 0x7fffffff) atomicMin(&s_pick, first);
    } else {
        unsigned cnt = 0;
#pragma unroll
        for (int k = 0; k < C; k++) cnt += key[k] == kpick ? 1u : 0u;
        unsigned inc = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned o = __shfl_up(inc, d);
            if ((tid & 63) >= d) inc += o;
        }
        if ((tid & 63) == 63) wcnt[tid >> 6] = inc;
        __syncthreads();
        unsigned pre = inc - cnt;
        for (int w = 0; w < (tid >> 6); w++) pre += wcnt[w];
        if (pre < rank && rank <= pre + cnt) {
            unsigned seen = pre;
#pragma unroll
            for (int k = 0; k < C; k++) {
                if (key[k] == kpick) { seen++; if (seen == rank) s_pick = lo + k; }
            }
        }
    }
    __syncthreads();
    SAMP_STAMP(13);
    const int pick = s_pick == 0x7fffffff ? 0 : s_pick;
    samp_embed_tail(P, pick);
    // recent window: append, drop the oldest when full (go/main.go:197-200)
    const int n = *P.recent_n;
    int shifted = 0;
    if (P.rep_window > 0 && n >= P.rep_window && tid + 1 < n) shifted = P.recent[tid + 1];   // rep_window <= 1024
    __syncthreads();
    if (P.rep_window > 0) {
        if (n >= P.rep_window) {
            if (tid + 1 < n) P.recent[tid] = shifted;
            if (tid == 0) P.recent[n - 1] = pick;
        } else if (tid == 0) {
            P.recent[n] = pick;
            *P.recent_n = n + 1;
        }
    }
    if (tid == 0) {
        P.ids[step] = pick;
        P.ctl[CTL_STEP] = step + 1;
        P.ctl[CTL_TOKEN] = pick;
        P.ctl[CTL_POS] = P.ctl[CTL_POS] + 1;
    }
    SAMP_STAMP(14);
}

// ---- the same selection for vocabularies of 32769 .. 131072 candidates (the 7.9B tier): they do not fit the registers of one
// workgroup, so every pass STREAMS the keys out of L2 -- coalesced 16-byte loads through the L2 only (agent scope: plain loads
// of an L2-resident array reach 37 GB/s per compute unit, these 60-110, tools/ingest_probe.hip), eight in flight per lane.
// Passes over the keys: range, level-1 histogram, one list pass per selection, the pick's id -- five instead of a 2 x 12-launch
// device-wide sort; everything else is samp_select_radix_kernel's.
__global__ void __launch_bounds__(SAMP_THREADS) samp_select_radix_stream_kernel(SampleParams P) {
    __shared__ samp_u64 hw[2048];
    __shared__ unsigned list[SAMP_LIST_CAP];
    __shared__ samp_u64 scr[16];
    __shared__ unsigned wcnt[SAMP_THREADS / 64];
    __shared__ unsigned list_n;
    __shared__ int s_pick;
    const int tid = threadIdx.x, V = P.vocab;
    const int step = P.ctl[CTL_STEP];
    const float u = P.uniforms[step];
    const unsigned *kin = reinterpret_cast<const unsigned *>(P.keys_in);
    if (tid == 0) s_pick = 0x7fffffff;
    int par = 0;
    constexpr unsigned kmin = SAMP_KEY_MIN;                  // level 1 comes from samp_prob_hist_kernel, keys relative to SAMP_KEY_MIN
    constexpr int s1 = SAMP_S1, s2 = SAMP_S2;
    const samp_u64 *const h1 = P.h1g;
    auto weight_of = [](unsigned k) { return samp_weight_of(k); };
    __syncthreads();
    auto select = [&](auto x_of, unsigned &dout, samp_u64 &rem, samp_u64 &x) {
        const SampSel a = samp_select_bucket(h1, x_of, scr, par);
        unsigned prefix = (unsigned)a.bucket;
        rem = a.rem;
        x = a.x;
        if (tid == 0) list_n = 0;
        hw[tid] = 0; hw[tid + 1024] = 0;
        __syncthreads();
        samp_each_key(kin, V, [&](unsigned key, int) {
            const unsigned d = key - kmin;                  // (a non-candidate wraps to >= 2^32 - 2^30: its d >> s1 is no bucket)
            if ((d >> s1) == prefix) {
                const unsigned at = atomicAdd(&list_n, 1u);
                if (at < (unsigned)SAMP_LIST_CAP) list[at] = d;
            }
        });
        __syncthreads();
        const unsigned n = list_n;
        const bool listed = n <= (unsigned)SAMP_LIST_CAP;
        auto same = [](samp_u64 r) { return [r](samp_u64) { return r; }; };
        {
            const unsigned mask = (1u << (s1 - s2)) - 1u;
            if (listed) {
                for (unsigned i = tid; i < n; i += SAMP_THREADS) { const unsigned d = list[i]; atomicAdd(&hw[(d >> s2) & mask], weight_of(d + kmin)); }
            } else {
                samp_each_key(kin, V, [&](unsigned key, int) {
                    const unsigned d = key - kmin;
                    if ((d >> s1) == prefix) atomicAdd(&hw[(d >> s2) & mask], weight_of(key));
                });
            }
            __syncthreads();
            const SampSel b = samp_select_bucket(hw, same(rem), scr, par);
            prefix = (prefix << (s1 - s2)) | (unsigned)b.bucket;
            rem = b.rem;
        }
        if (s2 > 0) {
            hw[tid] = 0; hw[tid + 1024] = 0;
            __syncthreads();
            const unsigned mask = (1u << s2) - 1u;
            if (listed) {
                for (unsigned i = tid; i < n; i += SAMP_THREADS) { const unsigned d = list[i]; if ((d >> s2) == prefix) atomicAdd(&hw[d & mask], weight_of(d + kmin)); }
            } else {
                samp_each_key(kin, V, [&](unsigned key, int) {
                    const unsigned d = key - kmin;
                    if ((d >> s2) == prefix) atomicAdd(&hw[d & mask], weight_of(key));
                });
            }
            __syncthreads();
            const SampSel c = samp_select_bucket(hw, same(rem), scr, par);
            prefix = (prefix << s2) | (unsigned)c.bucket;
            rem = c.rem;
        }
        dout = prefix;
    };
    auto tie_rank = [](samp_u64 rem, samp_u64 w) { return rem <= w ? (samp_u64)1 : (rem + w - 1) / w; };
    unsigned dc, dp;
    samp_u64 rem, xcut, xr;
    const float top_p = P.top_p;
    select([top_p](samp_u64 total) { const samp_u64 x = samp_ceil_mul(top_p, total); return x < 1 ? (samp_u64)1 : x; }, dc, rem, xcut);
    const samp_u64 wc = weight_of(dc + kmin);
    const samp_u64 cum = xcut - rem + tie_rank(rem, wc) * wc;
    select([u, cum](samp_u64) { const samp_u64 x = samp_ceil_mul(u, cum); return x < 1 ? (samp_u64)1 : x; }, dp, rem, xr);
    const unsigned rank = (unsigned)tie_rank(rem, weight_of(dp + kmin));
    const unsigned kpick = dp + kmin;
    if (rank == 1) {
        int first = 0x7fffffff;
        samp_each_key(kin, V, [&](unsigned key, int i) { if (key == kpick) first = min(first, i); });
        if (first != 0x7fffffff) atomicMin(&s_pick, first);
    } else {
        // ties: the rank-th candidate with this p in ascending id -- contiguous ids per thread (uncoalesced reads: the rare path)
        const int C = (V + SAMP_THREADS - 1) / SAMP_THREADS, lo = tid * C;
        unsigned cnt = 0;
        for (int k = 0; k < C; k++) cnt += (lo + k < V && kin[lo + k] == kpick) ? 1u : 0u;
        unsigned inc = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned o = __shfl_up(inc, d);
            if ((tid & 63) >= d) inc += o;
        }
        if ((tid & 63) == 63) wcnt[tid >> 6] = inc;
        __syncthreads();
        unsigned pre = inc - cnt;
        for (int w = 0; w < (tid >> 6); w++) pre += wcnt[w];
        if (pre < rank && rank <= pre + cnt) {
            unsigned seen = pre;
            for (int k = 0; k < C; k++)
                if (lo + k < V && kin[lo + k] == kpick) { seen++; if (seen == rank) s_pick = lo + k; }
        }
    }
    __syncthreads();
    const int pick = s_pick == 0x7fffffff ? 0 : s_pick;
    samp_embed_tail(P, pick);
    const int n = *P.recent_n;
    int shifted = 0;
    if (P.rep_window > 0 && n >= P.rep_window && tid + 1 < n) shifted = P.recent[tid + 1];   // rep_window <= 1024
    __syncthreads();
    if (P.rep_window > 0) {
        if (n >= P.rep_window) {
            if (tid + 1 < n) P.recent[tid] = shifted;
            if (tid == 0) P.recent[n - 1] = pick;
        } else if (tid == 0) {
            P.recent[n] = pick;
            *P.recent_n = n + 1;
        }
    }
    if (tid == 0) {
        P.ids[step] = pick;
        P.ctl[CTL_STEP] = step + 1;
        P.ctl[CTL_TOKEN] = pick;
        P.ctl[CTL_POS] = P.ctl[CTL_POS] + 1;
    }
}

}  // namespace nl
