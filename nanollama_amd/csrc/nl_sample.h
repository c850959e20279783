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
// The descending order is a stable radix sort (rocPRIM, ties keep ascending index -- Go's sort.Slice leaves the
// order of equal probabilities unspecified).  Sums: the top-k path adds its <= top_k terms sequentially exactly as
// the Go loop does; the top-p path adds V terms in fixed chunks (1024 contiguous chunks, each summed left to right,
// chunk totals accumulated left to right) -- deterministic, but not the Go loop's single left-to-right chain, so a
// u within ~1e-7 of a cdf boundary can select the neighbouring candidate.
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
    float *keys_in, *keys_out;
    int *idx_in, *idx_out;
    float *partial;       // [nblocks of samp_prob_kernel]
    float *scal;          // [0] = max logit after the penalty
    int nblocks;
};

// device scratch of one sampler (one vocabulary)
struct SampScratch {
    float *keys_in = nullptr, *keys_out = nullptr, *partial = nullptr, *scal = nullptr, *uniforms = nullptr;
    int *idx_in = nullptr, *idx_out = nullptr, *recent = nullptr, *recent_n = nullptr;
    void *sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
};

// 1 workgroup: repetition penalty, then the maximum logit
__global__ void __launch_bounds__(SAMP_THREADS) samp_penalty_kernel(SampleParams P) {
    __shared__ float red[SAMP_THREADS / 64];
    __shared__ int win[SAMP_THREADS];
    const int tid = threadIdx.x, n = *P.recent_n;
    // the maximum scan's loads do not depend on the penalty: issue the first batch now
    float m = -INFINITY;
    if (P.rep_penalty > 1.0f && n > 0) {
        if (tid < n) win[tid] = P.recent[tid];
        __syncthreads();
        if (tid < n) {
            const int tok = win[tid];
            if (tok >= 0 && tok < P.vocab) {
                bool first = true;
                for (int s = 0; s < tid; s++) first = first && win[s] != tok;
                if (first) {
                    int count = 0;
                    for (int s = tid; s < n; s++) count += win[s] == tok;
                    float v = P.logits[tok];
                    for (int c = 0; c < count; c++) v = v > 0.f ? v / P.rep_penalty : v * P.rep_penalty;
                    P.logits[tok] = v;
                }
            }
        }
    }
    __syncthreads();   // (orders the penalty stores before the scan below; same workgroup)
    for (int i0 = tid; i0 < P.vocab; i0 += 8 * SAMP_THREADS) {   // 8 independent loads per round (clamped, masked)
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = P.logits[min(i0 + k * SAMP_THREADS, P.vocab - 1)];
#pragma unroll
        for (int k = 0; k < 8; k++) m = i0 + k * SAMP_THREADS < P.vocab ? fmaxf(m, v[k]) : m;
    }
    m = wave_max_f32(m);
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < SAMP_THREADS / 64; w++) m = fmaxf(m, red[w]);
        P.scal[0] = m;
    }
}

// sort keys: top-p mode = unnormalised probabilities (+ per-workgroup partial sums), top-k mode = the logits
__global__ void __launch_bounds__(256) samp_prob_kernel(SampleParams P) {
    __shared__ float red[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool top_p_mode = P.top_p < 1.0f;
    float key = 0.f;
    if (i < P.vocab) {
        const float l = P.logits[i];
        key = top_p_mode ? (float)exp((double)((l - P.scal[0]) / P.temp)) : l;
        P.keys_in[i] = key;
        P.idx_in[i] = i;
    } else if (top_p_mode) {
        key = 0.f;
    }
    if (top_p_mode) {
        float s = i < P.vocab ? key : 0.f;
        s = wave_sum_f32(s);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) P.partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

// 1 workgroup: pick the token from the sorted candidates and advance the decode state
template <int C>   // chunk length per lane: a multiple of 32 with 1024 * C >= vocab (keys_out is zero-padded to 1024 * C)
__global__ void __launch_bounds__(SAMP_THREADS) samp_select_kernel(SampleParams P) {
    __shared__ float chunk[SAMP_THREADS];
    __shared__ float bval[SAMP_THREADS / 64];
    __shared__ int bidx[SAMP_THREADS / 64];
    __shared__ float wtot[SAMP_THREADS / 64];
    __shared__ int s_cut, s_pick, s_pick_pos;
    __shared__ float s_cum, s_inv;
    const int tid = threadIdx.x, V = P.vocab;
    const int step = P.ctl[CTL_STEP];
    const float u = P.uniforms[step];
    if (tid == 0) { s_cut = 0x7fffffff; s_pick = -1; s_pick_pos = 0x7fffffff; s_cum = 0.f; }
    int pick = 0;
    if (P.temp <= 0.f) {
        // argmax over the penalised logits, lowest index wins ties
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
    } else if (P.top_p < 1.0f) {
        if (tid == 0) {
            float sum = 0.f;
            for (int b = 0; b < P.nblocks; b++) sum += P.partial[b];
            s_inv = 1.0f / sum;
        }
        __syncthreads();
        const float inv = s_inv;
        // cum(i) = (prefix of the earlier wavefronts' totals + prefix of the earlier lanes' chunk totals) + the
        // left-to-right sum inside this lane's contiguous chunk: every chain is a fixed left-to-right order
        const int lo = tid * C, hi = min(lo + C, V);
        float q[C];                                        // this lane's chunk, normalised; one round of loads
#pragma unroll
        for (int k = 0; k < C; k += 4) {
            const float4 t4 = *reinterpret_cast<const float4 *>(P.keys_out + lo + k);
            q[k] = t4.x * inv; q[k + 1] = t4.y * inv; q[k + 2] = t4.z * inv; q[k + 3] = t4.w * inv;
        }
        float local = 0.f;
#pragma unroll
        for (int k = 0; k < C; k++) local += q[k];         // entries past vocab are +0.0f
        chunk[tid] = local;
        __syncthreads();
        const int wv = tid >> 6, ln = tid & 63;
        float within = 0.f;
        for (int t = wv * 64; t < tid; t++) within += chunk[t];
        if (ln == 63) wtot[wv] = within + local;
        __syncthreads();
        float pre = 0.f;
        for (int w = 0; w < wv; w++) pre += wtot[w];
        pre += within;
        // (1) the cut: the smallest i with cum(i) >= top_p.  Every lane scans its own chunk and the minimum wins, so
        //     a one-ulp disagreement between a chunk's last cum and the next chunk's prefix cannot lose the cut.
        {
            float part = 0.f;
            int found = 0x7fffffff;
#pragma unroll
            for (int k = 0; k < C; k++) {
                part += q[k];
                if (found == 0x7fffffff && lo + k < hi && pre + part >= P.top_p) found = lo + k;
            }
            if (found != 0x7fffffff) atomicMin(&s_cut, found);
        }
        __syncthreads();
        const int cut = s_cut;
        if (cut < 0x7fffffff) {
            if (lo <= cut && cut < hi) {
                float part = 0.f;
#pragma unroll
                for (int k = 0; k < C; k++) part += lo + k <= cut ? q[k] : 0.f;
                s_cum = pre + part;
            }
            __syncthreads();
            // (2) the smallest j <= cut with r <= cum(j)
            const float r = u * s_cum;
            float part = 0.f;
            int found = 0x7fffffff;
#pragma unroll
            for (int k = 0; k < C; k++) {
                part += q[k];
                if (found == 0x7fffffff && lo + k < hi && lo + k <= cut && r <= pre + part) found = lo + k;
            }
            if (found != 0x7fffffff) atomicMin(&s_pick_pos, found);
        }
        __syncthreads();
        if (tid == 0) s_pick = P.idx_out[s_pick_pos < 0x7fffffff ? s_pick_pos : 0];
    } else {
        // top-k: the Go loop (go/main.go:325-342).  The <= 1024 candidate probabilities are computed one per lane
        // (each is a pure function of its own logit), the two sums run left to right on lane 0 exactly as in Go.
        const int K = min(P.top_k, V);
        const float v0 = P.keys_out[0];
        if (K <= SAMP_THREADS) {
            if (tid < K) chunk[tid] = (float)exp((double)((P.keys_out[tid] - v0) / P.temp));
            __syncthreads();
            if (tid == 0) {
                float sum = 0.f;
                for (int i = 0; i < K; i++) sum += chunk[i];
                const float r = u * sum;
                float cdf = 0.f;
                int sel = 0;
                for (int i = 0; i < K; i++) {
                    cdf += chunk[i];
                    if (r <= cdf) { sel = i; break; }
                }
                s_pick = P.idx_out[sel];
            }
        } else if (tid == 0) {
            float sum = 0.f;
            for (int i = 0; i < K; i++) sum += (float)exp((double)((P.keys_out[i] - v0) / P.temp));
            const float r = u * sum;
            float cdf = 0.f;
            int sel = 0;
            for (int i = 0; i < K; i++) {
                cdf += (float)exp((double)((P.keys_out[i] - v0) / P.temp));
                if (r <= cdf) { sel = i; break; }
            }
            s_pick = P.idx_out[sel];
        }
    }
    __syncthreads();
    pick = s_pick;
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
}

}  // namespace nl
