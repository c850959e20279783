// nl_batch.h -- element-wise kernels of the multi-token step (batched decode streams / prompt prefill).
// The heavy lifting is qgemm_kernel (nl_qgemm.h) and the per-token attention kernel (nl_kernels.h, batched
// over blockIdx.z); everything here is one workgroup per token.
#pragma once
#include "nl_kernels.h"

namespace nl {

struct BEmbedParams {
    const uint8_t *table;
    int wtype, dim;
    const int *tokens;
    float *x;  // [N][dim]
    const int *gamma_row;
    const float *gamma_val;
};

__global__ void bembed_kernel(BEmbedParams P) {
    const int token = P.tokens[blockIdx.x];
    float *x = P.x + (long long)blockIdx.x * P.dim;
    const int gr = P.gamma_row ? P.gamma_row[token] : -1;
    for (int i = threadIdx.x; i < P.dim; i += blockDim.x) {
        float v = embed_value(P.table, P.wtype, P.dim, token, i);
        if (gr >= 0) v += P.gamma_val[(long long)gr * P.dim + i];
        x[i] = v;
    }
}

// RMSNormInto go/quant.go:597-607, one workgroup per token
__global__ void brmsnorm_kernel(const float *x, const float *w, float eps, float *out, int n) {
    __shared__ double dred[4];
    const float *xr = x + (long long)blockIdx.x * n;
    float *o = out + (long long)blockIdx.x * n;
    double ss = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) ss += (double)xr[i] * (double)xr[i];
    ss = wave_sum_f64(ss);
    if ((threadIdx.x & 63) == 0) dred[threadIdx.x >> 6] = ss;
    __syncthreads();
    double tot = 0.0;
    for (int k = 0; k < (int)(blockDim.x >> 6); k++) tot += dred[k];
    float inv = (float)(1.0 / sqrt(tot / (double)n + (double)eps));
    for (int i = threadIdx.x; i < n; i += blockDim.x) o[i] = (xr[i] * inv) * w[i];
}

struct BRopeParams {
    const float *qkv;       // [N][R], rows in the packed (ROWMAP_HEADPERM) tile order of the QKV matrix
    int R, head_dim, n_q_heads, n_kv_heads, seq_len, rope_conj, qk_norm;
    float eps;
    const int *pos, *stream;
    const float *rope_cos, *rope_sin;
    float *q;               // [N][n_q_heads*hd] natural order
    float *kcache, *vcache; // this layer, stream 0
    long long kv_stream_stride;
    const float *bias_q, *bias_k, *bias_v;  // optional, natural (head, element) order
};

// RoPE (go/model.go:449-477) + optional QK-norm (:542-549) + KV store (:552-554) for one token per workgroup
__global__ void brope_kv_kernel(BRopeParams P) {
    extern __shared__ float vals[];  // [R] in natural (head, element) order after RoPE
    const int item = blockIdx.x, hd = P.head_dim, half = hd >> 1, tph = hd / 16;
    const int pos = P.pos[item];
    const float *src = P.qkv + (long long)item * P.R;
    for (int rho = threadIdx.x; rho < P.R; rho += blockDim.x) {
        const int tile = rho / TR, r = rho % TR;
        const int head = tile / tph, j = tile % tph;
        const int i = j * 8 + (r & 7), e = i + (r >> 3) * half;
        float v = src[rho], outv = v;
        float partner = src[rho ^ 8];
        if (P.bias_q) {
            const int ep = i + ((r ^ 8) >> 3) * half;   // the partner's element index
            if (head < P.n_q_heads) { v += P.bias_q[head * hd + e]; partner += P.bias_q[head * hd + ep]; }
            else if (head < P.n_q_heads + P.n_kv_heads) {
                v += P.bias_k[(head - P.n_q_heads) * hd + e]; partner += P.bias_k[(head - P.n_q_heads) * hd + ep];
            } else v += P.bias_v[(head - P.n_q_heads - P.n_kv_heads) * hd + e];
            outv = v;
        }
        if (head < P.n_q_heads + P.n_kv_heads) {
            float c = P.rope_cos[pos * half + i], s = P.rope_sin[pos * half + i];
            float x0 = (r < 8) ? v : partner, x1 = (r < 8) ? partner : v;
            if (!P.rope_conj) outv = (r < 8) ? (x0 * c - x1 * s) : (x0 * s + x1 * c);
            else outv = (r < 8) ? (x0 * c + x1 * s) : (-x0 * s + x1 * c);
        }
        vals[head * hd + e] = outv;
    }
    __syncthreads();
    if (P.qk_norm) {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
        for (int head = wave; head < P.n_q_heads + P.n_kv_heads; head += nw) {
            float *vec = vals + head * hd;
            double ss = 0.0;
            for (int i = lane; i < hd; i += 64) ss += (double)vec[i] * (double)vec[i];
            ss = wave_sum_f64(ss);
            float inv = (float)(1.0 / sqrt(ss / (double)hd + (double)P.eps));
            for (int i = lane; i < hd; i += 64) vec[i] = vec[i] * inv;
        }
        __syncthreads();
    }
    const long long soff = (long long)P.stream[item] * P.kv_stream_stride;
    const int nq = P.n_q_heads * hd, nk = P.n_kv_heads * hd;
    for (int i = threadIdx.x; i < P.R; i += blockDim.x) {
        float v = vals[i];
        if (i < nq) {
            P.q[(long long)item * nq + i] = v;
        } else if (i < nq + nk) {
            int kvh = (i - nq) / hd, e = (i - nq) % hd;
            P.kcache[soff + ((long long)kvh * P.seq_len + pos) * hd + e] = v;
        } else {
            int kvh = (i - nq - nk) / hd, e = (i - nq - nk) % hd;
            P.vcache[soff + ((long long)kvh * P.seq_len + pos) * hd + e] = v;
        }
    }
}

struct BMergeParams {
    const float *part_o, *part_ml;  // [N][heads][nsplit_max][hd] / [..][2]
    const int *pos;
    int heads, nsplit_max, head_dim;
    float *out;  // [N][heads*hd]
};

// online-softmax merge of the position splits (same arithmetic as the decode GEMV's PRO_ATTN prologue)
__global__ void battn_merge_kernel(BMergeParams P) {
    const int item = blockIdx.x, hd = P.head_dim;
    const int ns = P.pos[item] / ATT_CH + 1;
    const long long pbase = (long long)item * P.heads * P.nsplit_max;
    for (int i = threadIdx.x; i < P.heads * hd; i += blockDim.x) {
        const int h = i / hd, d = i - h * hd;
        const float *ml = P.part_ml + (pbase + (long long)h * P.nsplit_max) * 2;
        const float *po = P.part_o + (pbase + (long long)h * P.nsplit_max) * hd + d;
        float r;
        if (ns == 1) {
            r = po[0] * (1.0f / ml[1]);
        } else {
            float M = ml[0];
            for (int c = 1; c < ns; c++) M = fmaxf(M, ml[2 * c]);
            float L = 0.f, o = 0.f;
            for (int c = 0; c < ns; c++) {
                float w = (float)exp((double)(ml[2 * c] - M));
                L += w * ml[2 * c + 1];
                o += w * po[(long long)c * hd];
            }
            r = o * (1.0f / L);
        }
        P.out[(long long)item * P.heads * hd + i] = r;
    }
}

// SiLU(gate) * up, go/quant.go:629-631 + go/model.go:604-606
__global__ void bswiglu_kernel(const float *g, const float *u, float *out, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float v = g[i];
        float ex = (float)exp((double)(-v));
        out[i] = (v / (1.0f + ex)) * u[i];
    }
}

// argmax per token (go/main.go:400-408), one workgroup per token
__global__ void __launch_bounds__(1024) bargmax_kernel(const float *logits, int n, int *ids) {
    __shared__ float bv[16];
    __shared__ int bi[16];
    const float *lg = logits + (long long)blockIdx.x * n;
    const int tid = threadIdx.x;
    float best = -INFINITY;
    int idx = 0x7fffffff;
    for (int i = tid; i < n; i += blockDim.x) {
        float v = lg[i];
        if (v > best || idx == 0x7fffffff) { best = v; idx = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(best, o);
        int oi = __shfl_xor(idx, o);
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
