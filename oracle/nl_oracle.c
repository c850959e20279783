/*
 * nl_oracle.c -- CPU restatement of the nanollama Go inference hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under nanollama_amd/ (the product) may
 * import, link, call or execute this file.  It is used by tests/, by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg as the checker /
 * reported CPU baseline, never as the thing measured or shipped.
 *
 * It restates, in plain C, the arithmetic of the reference's Go engine
 * (ariannamethod/nanollama, go/quant.go + go/model.go + go/gguf.go +
 * go/main.go).  Every function cites the reference file:line it follows.
 * The Go toolchain is absent from the build image, so the reference binary
 * itself cannot be built; this restatement is PINNED instead against golden
 * logits produced in the build container by the reference's own importable
 * Python (scripts/export_gguf.py writer + quantisers, nanollama/llama.py
 * model) -- see tests/golden/make_goldens.py and tests/test_oracle_golden.py.
 *
 * Arithmetic discipline (must be compiled with -O2 -ffp-contract=off):
 *   - Go on amd64 never fuses a*b+c, so neither do we (no FMA contraction).
 *   - float32 accumulations are sequential in exactly the Go loop order.
 *   - float64 is used exactly where the Go code converts to float64
 *     (RMSNorm sum of squares, exp() in Softmax/SiLU, RoPE tables).
 *   - libm differences: Go's math.Exp/Pow/Cos/Sin are its own routines, glibc's
 *     are ours; both are <1ulp in float64 and every use is rounded to float32
 *     right after, so results agree except on astronomically rare ties.
 *
 * Threading follows go/quant.go:49-72: rows are split into numWorkers
 * contiguous chunks, serial when rows < 4*numWorkers; each out[i] is produced
 * by exactly one worker in a fixed order, so results do not depend on the
 * thread count.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NLO_API __attribute__((visibility("default")))

/* ggml tensor types, go/gguf.go:43-57 */
enum {
    GGML_F32 = 0, GGML_F16 = 1, GGML_Q4_0 = 2, GGML_Q5_0 = 6, GGML_Q8_0 = 8,
    GGML_Q4_K = 12, GGML_Q6_K = 14
};

static int g_workers = 1;

NLO_API void nlo_set_threads(int n) { g_workers = n < 1 ? 1 : n; }
NLO_API int nlo_get_threads(void) { return g_workers; }

/* ---------------------------------------------------------------- fp16 --- */

/* go/gguf.go:603-636 -- the LUT construction, evaluated per call. */
NLO_API float nlo_half2float(uint16_t h) {
    uint32_t sign = (h >> 15) & 1u, exp = (h >> 10) & 0x1Fu, mant = h & 0x3FFu, f;
    if (exp == 0) {
        if (mant == 0) {
            f = sign << 31;
        } else {
            uint32_t e = 1;
            while ((mant & 0x400u) == 0) { mant <<= 1; e--; }
            mant &= 0x3FFu;
            f = (sign << 31) | ((e + 127u - 15u) << 23) | (mant << 13);
        }
    } else if (exp == 0x1F) {
        f = (sign << 31) | 0x7F800000u | (mant << 13);
    } else {
        f = (sign << 31) | ((exp - 15u + 127u) << 23) | (mant << 13);
    }
    float out;
    memcpy(&out, &f, 4);
    return out;
}

static float h2f_lut[65536];
static int h2f_ready = 0;
static void h2f_init(void) {
    if (h2f_ready) return;
    for (int h = 0; h < 65536; h++) h2f_lut[h] = nlo_half2float((uint16_t)h);
    h2f_ready = 1;
}
static inline float h2f(const uint8_t *p) { return h2f_lut[(uint16_t)(p[0] | (p[1] << 8))]; }

/* block sizes, go/gguf.go:239-272 */
NLO_API int nlo_block_bytes(uint32_t t) {
    switch (t) {
    case GGML_F32: return 4;
    case GGML_F16: return 2;
    case GGML_Q4_0: return 18;
    case GGML_Q8_0: return 34;
    case GGML_Q5_0: return 22;
    case GGML_Q6_K: return 210;
    case GGML_Q4_K: return 144;
    default: return 0;
    }
}
NLO_API int nlo_block_elems(uint32_t t) {
    switch (t) {
    case GGML_F32: case GGML_F16: return 1;
    case GGML_Q4_K: case GGML_Q6_K: return 256;
    default: return 32;
    }
}

/* ------------------------------------------------- block dequantisers --- */

/* go/quant.go:22-31 */
static void dequant_q4_0_block(const uint8_t *b, float *out) {
    float d = h2f(b);
    for (int j = 0; j < 16; j++) {
        uint8_t v = b[2 + j];
        int v0 = (int)(v & 0x0F) - 8, v1 = (int)(v >> 4) - 8;
        out[j] = (float)v0 * d;
        out[j + 16] = (float)v1 * d;
    }
}
/* go/quant.go:103-108 */
static void dequant_q8_0_block(const uint8_t *b, float *out) {
    float d = h2f(b);
    for (int j = 0; j < 32; j++) out[j] = (float)(int8_t)b[2 + j] * d;
}
/* go/quant.go:405-420 */
static void dequant_q5_0_block(const uint8_t *b, float *out) {
    float d = h2f(b);
    uint32_t qh = (uint32_t)b[2] | ((uint32_t)b[3] << 8) | ((uint32_t)b[4] << 16) | ((uint32_t)b[5] << 24);
    const uint8_t *qs = b + 6;
    for (int j = 0; j < 16; j++) {
        int lo = qs[j] & 0x0F, hi = qs[j] >> 4;
        int q0 = lo | (int)(((qh >> j) & 1u) << 4);
        int q1 = hi | (int)(((qh >> (j + 16)) & 1u) << 4);
        out[j] = (float)(q0 - 16) * d;
        out[j + 16] = (float)(q1 - 16) * d;
    }
}
/* go/quant.go:285-294 */
static void scale_min_k4(int j, const uint8_t *s, uint8_t *sc, uint8_t *m) {
    if (j < 4) { *sc = s[j] & 63; *m = s[j + 4] & 63; }
    else {
        *sc = (uint8_t)((s[j + 4] & 0x0F) | ((s[j - 4] >> 6) << 4));
        *m = (uint8_t)((s[j + 4] >> 4) | ((s[j] >> 6) << 4));
    }
}
/* go/quant.go:296-323 */
static void dequant_q4_k_block(const uint8_t *b, float *out) {
    float d = h2f(b), dmin = h2f(b + 2);
    const uint8_t *scales = b + 4, *qs = b + 16;
    int is = 0, oi = 0, qi = 0;
    for (int j = 0; j < 256; j += 64) {
        uint8_t sc0, m0, sc1, m1v;
        scale_min_k4(is, scales, &sc0, &m0);
        float d1 = d * (float)sc0, m1 = dmin * (float)m0;
        scale_min_k4(is + 1, scales, &sc1, &m1v);
        float d2 = d * (float)sc1, m2 = dmin * (float)m1v;
        for (int l = 0; l < 32; l++) out[oi + l] = d1 * (float)(qs[qi + l] & 0x0F) - m1;
        for (int l = 0; l < 32; l++) out[oi + 32 + l] = d2 * (float)(qs[qi + l] >> 4) - m2;
        qi += 32; oi += 64; is += 2;
    }
}
/* go/quant.go:174-208 (one 256-element super block) */
static void dequant_q6_k_block(const uint8_t *b, float *out) {
    const uint8_t *ql = b, *qh = b + 128;
    const int8_t *scales = (const int8_t *)(b + 192);
    float d = h2f(b + 208);
    for (int n128 = 0; n128 < 2; n128++) {
        const uint8_t *qlP = ql + n128 * 64, *qhP = qh + n128 * 32;
        const int8_t *scP = scales + n128 * 8;
        float *y = out + n128 * 128;
        for (int l = 0; l < 32; l++) {
            int is = l / 16;
            int q1 = (qlP[l] & 0x0F) | (((qhP[l] >> 0) & 3) << 4);
            int q2 = (qlP[l + 32] & 0x0F) | (((qhP[l] >> 2) & 3) << 4);
            int q3 = (qlP[l] >> 4) | (((qhP[l] >> 4) & 3) << 4);
            int q4 = (qlP[l + 32] >> 4) | (((qhP[l] >> 6) & 3) << 4);
            y[l + 0] = d * (float)scP[is + 0] * (float)(q1 - 32);
            y[l + 32] = d * (float)scP[is + 2] * (float)(q2 - 32);
            y[l + 64] = d * (float)scP[is + 4] * (float)(q3 - 32);
            y[l + 96] = d * (float)scP[is + 6] * (float)(q4 - 32);
        }
    }
}

/* Dequantise n elements of a tensor: go/quant.go:34-42,110-118,174-208,325-333,
 * 422-430 and the F32/F16 arms of getF32Tensor go/model.go:268-303. */
NLO_API int nlo_dequant(float *out, const void *data, uint32_t type, int64_t n) {
    h2f_init();
    const uint8_t *p = (const uint8_t *)data;
    switch (type) {
    case GGML_F32: memcpy(out, p, (size_t)n * 4); return 0;
    case GGML_F16: for (int64_t i = 0; i < n; i++) out[i] = h2f(p + 2 * i); return 0;
    case GGML_Q4_0: for (int64_t i = 0; i < n / 32; i++) dequant_q4_0_block(p + 18 * i, out + 32 * i); return 0;
    case GGML_Q8_0: for (int64_t i = 0; i < n / 32; i++) dequant_q8_0_block(p + 34 * i, out + 32 * i); return 0;
    case GGML_Q5_0: for (int64_t i = 0; i < n / 32; i++) dequant_q5_0_block(p + 22 * i, out + 32 * i); return 0;
    case GGML_Q4_K: for (int64_t i = 0; i < n / 256; i++) dequant_q4_k_block(p + 144 * i, out + 256 * i); return 0;
    case GGML_Q6_K: for (int64_t i = 0; i < n / 256; i++) dequant_q6_k_block(p + 210 * i, out + 256 * i); return 0;
    default: return -1;
    }
}

/* ------------------------------------------------------------ matmuls --- */

/* go/quant.go:74-94 */
static void mm_q4_0_range(float *out, const uint8_t *w, const float *x, int s, int e, int bpr) {
    int64_t bytesPerRow = (int64_t)bpr * 18;
    for (int i = s; i < e; i++) {
        const uint8_t *row = w + (int64_t)i * bytesPerRow;
        float sum = 0.0f;
        for (int b = 0; b < bpr; b++) {
            const uint8_t *blk = row + b * 18;
            float d = h2f(blk);
            const float *xb = x + b * 32;
            float dot = 0.0f;
            for (int j = 0; j < 16; j++) {
                uint8_t bv = blk[2 + j];
                float v0 = (float)((int)(bv & 0x0F) - 8);
                float v1 = (float)((int)(bv >> 4) - 8);
                float p0 = v0 * xb[j];
                float p1 = v1 * xb[j + 16];
                float t = p0 + p1;
                dot = dot + t;
            }
            float sd = dot * d;
            sum = sum + sd;
        }
        out[i] = sum;
    }
}
/* go/quant.go:149-165 */
static void mm_q8_0_range(float *out, const uint8_t *w, const float *x, int s, int e, int bpr) {
    int64_t bytesPerRow = (int64_t)bpr * 34;
    for (int i = s; i < e; i++) {
        const uint8_t *row = w + (int64_t)i * bytesPerRow;
        float sum = 0.0f;
        for (int b = 0; b < bpr; b++) {
            const uint8_t *blk = row + b * 34;
            float d = h2f(blk);
            const float *xb = x + b * 32;
            float dot = 0.0f;
            for (int j = 0; j < 32; j++) {
                float p = (float)(int8_t)blk[2 + j] * xb[j];
                dot = dot + p;
            }
            float sd = dot * d;
            sum = sum + sd;
        }
        out[i] = sum;
    }
}
/* go/quant.go:461-484 */
static void mm_q5_0_range(float *out, const uint8_t *w, const float *x, int s, int e, int bpr) {
    int64_t bytesPerRow = (int64_t)bpr * 22;
    for (int r = s; r < e; r++) {
        const uint8_t *row = w + (int64_t)r * bytesPerRow;
        float sum = 0.0f;
        for (int b = 0; b < bpr; b++) {
            const uint8_t *blk = row + b * 22;
            float d = h2f(blk);
            uint32_t qh = (uint32_t)blk[2] | ((uint32_t)blk[3] << 8) | ((uint32_t)blk[4] << 16) | ((uint32_t)blk[5] << 24);
            const uint8_t *qs = blk + 6;
            const float *xb = x + b * 32;
            for (int j = 0; j < 16; j++) {
                int lo = qs[j] & 0x0F, hi = qs[j] >> 4;
                int q0 = lo | (int)(((qh >> j) & 1u) << 4);
                int q1 = hi | (int)(((qh >> (j + 16)) & 1u) << 4);
                float a = (float)(q0 - 16) * d; a = a * xb[j]; sum = sum + a;
                float c = (float)(q1 - 16) * d; c = c * xb[j + 16]; sum = sum + c;
            }
        }
        out[r] = sum;
    }
}
/* go/quant.go:364-396 */
static void mm_q4_k_range(float *out, const uint8_t *w, const float *x, int s, int e, int bpr) {
    int64_t bytesPerRow = (int64_t)bpr * 144;
    for (int r = s; r < e; r++) {
        const uint8_t *row = w + (int64_t)r * bytesPerRow;
        float sum = 0.0f;
        for (int b = 0; b < bpr; b++) {
            const uint8_t *blk = row + b * 144;
            float d = h2f(blk), dmin = h2f(blk + 2);
            const uint8_t *scales = blk + 4, *qs = blk + 16;
            const float *xb = x + b * 256;
            int is = 0, qi = 0;
            for (int j = 0; j < 256; j += 64) {
                uint8_t sc0, m0, sc1, m1v;
                scale_min_k4(is, scales, &sc0, &m0);
                float d1 = d * (float)sc0, m1 = dmin * (float)m0;
                scale_min_k4(is + 1, scales, &sc1, &m1v);
                float d2 = d * (float)sc1, m2 = dmin * (float)m1v;
                for (int l = 0; l < 32; l++) {
                    float t = d1 * (float)(qs[qi + l] & 0x0F); t = t - m1; t = t * xb[j + l]; sum = sum + t;
                }
                for (int l = 0; l < 32; l++) {
                    float t = d2 * (float)(qs[qi + l] >> 4); t = t - m2; t = t * xb[j + 32 + l]; sum = sum + t;
                }
                qi += 32; is += 2;
            }
        }
        out[r] = sum;
    }
}
/* go/quant.go:239-276 */
static void mm_q6_k_range(float *out, const uint8_t *w, const float *x, int s, int e, int bpr) {
    int64_t bytesPerRow = (int64_t)bpr * 210;
    for (int r = s; r < e; r++) {
        const uint8_t *row = w + (int64_t)r * bytesPerRow;
        float sum = 0.0f;
        for (int b = 0; b < bpr; b++) {
            const uint8_t *blk = row + b * 210;
            const uint8_t *ql = blk, *qh = blk + 128;
            const int8_t *scales = (const int8_t *)(blk + 192);
            float d = h2f(blk + 208);
            const float *xb = x + b * 256;
            for (int n128 = 0; n128 < 2; n128++) {
                const uint8_t *qlP = ql + n128 * 64, *qhP = qh + n128 * 32;
                const int8_t *scP = scales + n128 * 8;
                const float *xx = xb + n128 * 128;
                for (int l = 0; l < 32; l++) {
                    int is = l / 16;
                    int q1 = (qlP[l] & 0x0F) | (((qhP[l] >> 0) & 3) << 4);
                    int q2 = (qlP[l + 32] & 0x0F) | (((qhP[l] >> 2) & 3) << 4);
                    int q3 = (qlP[l] >> 4) | (((qhP[l] >> 4) & 3) << 4);
                    int q4 = (qlP[l + 32] >> 4) | (((qhP[l] >> 6) & 3) << 4);
                    float s0 = d * (float)scP[is + 0], s2 = d * (float)scP[is + 2];
                    float s4 = d * (float)scP[is + 4], s6 = d * (float)scP[is + 6];
                    float t;
                    t = s0 * (float)(q1 - 32); t = t * xx[l + 0]; sum = sum + t;
                    t = s2 * (float)(q2 - 32); t = t * xx[l + 32]; sum = sum + t;
                    t = s4 * (float)(q3 - 32); t = t * xx[l + 64]; sum = sum + t;
                    t = s6 * (float)(q4 - 32); t = t * xx[l + 96]; sum = sum + t;
                }
            }
        }
        out[r] = sum;
    }
}
/* go/quant.go:516-525 */
static void mm_f32_range(float *out, const float *w, const float *x, int s, int e, int cols) {
    for (int i = s; i < e; i++) {
        float sum = 0.0f;
        const float *row = w + (int64_t)i * cols;
        for (int j = 0; j < cols; j++) { float p = row[j] * x[j]; sum = sum + p; }
        out[i] = sum;
    }
}
/* go/quant.go:553-563 */
static void mm_f16_range(float *out, const uint8_t *w, const float *x, int s, int e, int cols) {
    for (int i = s; i < e; i++) {
        float sum = 0.0f;
        const uint8_t *row = w + (int64_t)i * cols * 2;
        for (int j = 0; j < cols; j++) { float p = h2f(row + 2 * j) * x[j]; sum = sum + p; }
        out[i] = sum;
    }
}

static int mm_range(float *out, const void *w, uint32_t type, const float *x, int s, int e, int cols) {
    const uint8_t *p = (const uint8_t *)w;
    switch (type) {
    case GGML_Q4_0: mm_q4_0_range(out, p, x, s, e, cols / 32); return 0;
    case GGML_Q8_0: mm_q8_0_range(out, p, x, s, e, cols / 32); return 0;
    case GGML_Q5_0: mm_q5_0_range(out, p, x, s, e, cols / 32); return 0;
    case GGML_Q4_K: mm_q4_k_range(out, p, x, s, e, cols / 256); return 0;
    case GGML_Q6_K: mm_q6_k_range(out, p, x, s, e, cols / 256); return 0;
    case GGML_F16: mm_f16_range(out, p, x, s, e, cols); return 0;
    case GGML_F32: mm_f32_range(out, (const float *)w, x, s, e, cols); return 0;
    default: return -1;
    }
}

/* matmulDispatch go/model.go:361-386 + the goroutine fan-out of every MatMul*
 * (go/quant.go:45-72 and siblings).  Unknown type: the reference prints a
 * WARNING and leaves out[] stale (go/model.go:383-385); we return -1 too. */
NLO_API int nlo_matmul(float *out, const void *w, uint32_t type, const float *x, int rows, int cols) {
    h2f_init();
    if (nlo_block_bytes(type) == 0) {
        fprintf(stderr, "[oracle] WARNING: unsupported matmul type %u for %dx%d\n", type, rows, cols);
        return -1;
    }
    int nw = g_workers;
    if (rows < nw * 4 || nw == 1) return mm_range(out, w, type, x, 0, rows, cols);
    int chunk = (rows + nw - 1) / nw;
#ifdef _OPENMP
#pragma omp parallel for schedule(static, 1) num_threads(nw)
#endif
    for (int k = 0; k < nw; k++) {
        int s = k * chunk, e = s + chunk;
        if (e > rows) e = rows;
        if (s < e) mm_range(out, w, type, x, s, e, cols);
    }
    return 0;
}

/* ------------------------------------------------------ math utilities --- */

static float rms_inv(const float *x, int n, float eps) {
    double ss = 0.0;
    for (int i = 0; i < n; i++) ss += (double)x[i] * (double)x[i];
    return (float)(1.0 / sqrt(ss / (double)n + (double)eps));
}
/* go/quant.go:570-580 */
NLO_API void nlo_rmsnorm(float *x, const float *w, int n, float eps) {
    float inv = rms_inv(x, n, eps);
    for (int i = 0; i < n; i++) { float t = x[i] * inv; x[i] = t * w[i]; }
}
/* go/quant.go:584-594 */
NLO_API void nlo_rmsnorm_bare(float *x, int n, float eps) {
    float inv = rms_inv(x, n, eps);
    for (int i = 0; i < n; i++) x[i] = x[i] * inv;
}
/* go/quant.go:597-607 */
NLO_API void nlo_rmsnorm_into(float *out, const float *x, const float *w, int n, float eps) {
    float inv = rms_inv(x, n, eps);
    for (int i = 0; i < n; i++) { float t = x[i] * inv; out[i] = t * w[i]; }
}
/* go/quant.go:610-626 */
NLO_API void nlo_softmax(float *x, int n) {
    float max = x[0];
    for (int i = 1; i < n; i++) if (x[i] > max) max = x[i];
    float sum = 0.0f;
    for (int i = 0; i < n; i++) {
        float t = x[i] - max;
        x[i] = (float)exp((double)t);
        sum = sum + x[i];
    }
    float inv = 1.0f / sum;
    for (int i = 0; i < n; i++) x[i] = x[i] * inv;
}
/* go/quant.go:629-631 */
NLO_API float nlo_silu(float x) {
    float e = (float)exp((double)(-x));
    float den = 1.0f + e;
    return x / den;
}
/* go/main.go:400-408 -- strict '>' so the lowest index wins ties */
NLO_API int nlo_argmax(const float *logits, int n) {
    int best = 0;
    for (int i = 1; i < n; i++) if (logits[i] > logits[best]) best = i;
    return best;
}

/* ---------------------------------------------------------------- model --- */

typedef struct {
    int32_t n_layers, dim, n_heads, n_kv_heads, head_dim, interm, vocab, seq_len;
    float eps, rope_theta;
    int32_t qk_norm, rope_conjugate;
} nlo_config;

typedef struct { const void *w; uint32_t type; } nlo_mat;

typedef struct {
    float *attn_norm, *ffn_norm;
    nlo_mat wq, wk, wv, wo, wgate, wup, wdown;
    float *bq, *bk, *bv, *bo; /* optional biases, go/model.go:244-247 */
} nlo_layer;

typedef struct nlo_model {
    nlo_config cfg;
    nlo_mat token_embd, output;
    float *output_norm;
    nlo_layer *layers;
    /* LlamaState go/model.go:93-118 */
    float *x, *xb, *xb2, *hb, *hb2, *q, *k, *v, *att, *logits;
    float *key_cache, *value_cache, *cos_cache, *sin_cache, *emb_buf;
    /* GammaEssence go/gamma.go:22-35 (values already float32) */
    int32_t *gamma_idx; float *gamma_val; int gamma_n;
    int finalized;
    char err[256];
} nlo_model;

NLO_API nlo_model *nlo_create(const nlo_config *c) {
    h2f_init();
    nlo_model *m = (nlo_model *)calloc(1, sizeof(nlo_model));
    m->cfg = *c;
    /* go/model.go:140-148 */
    if (m->cfg.head_dim == 0 && m->cfg.n_heads > 0) m->cfg.head_dim = m->cfg.dim / m->cfg.n_heads;
    if (m->cfg.seq_len > 2048) m->cfg.seq_len = 2048;
    m->layers = (nlo_layer *)calloc((size_t)m->cfg.n_layers, sizeof(nlo_layer));
    return m;
}

NLO_API const char *nlo_last_error(nlo_model *m) { return m->err; }

static float *to_f32(const void *data, uint32_t type, int n) {
    float *out = (float *)malloc((size_t)n * 4);
    if (nlo_dequant(out, data, type, n) != 0) { free(out); return NULL; }
    return out;
}

/* loadWeights go/model.go:177-265: norms (and biases) are converted to float32
 * (getF32Tensor :268-303), matrices keep their raw bytes + type.  `data` for
 * matrices is aliased, not copied: the caller keeps it alive. */
NLO_API int nlo_set_tensor(nlo_model *m, const char *name, uint32_t type, const void *data) {
    const nlo_config *c = &m->cfg;
    int kvdim = c->n_kv_heads * c->head_dim;
    if (!strcmp(name, "token_embd.weight")) { m->token_embd.w = data; m->token_embd.type = type; return 0; }
    if (!strcmp(name, "output.weight")) { m->output.w = data; m->output.type = type; return 0; }
    if (!strcmp(name, "output_norm.weight")) { m->output_norm = to_f32(data, type, c->dim); return m->output_norm ? 0 : -1; }
    int li = -1, off = 0;
    if (sscanf(name, "blk.%d.%n", &li, &off) < 1 || li < 0 || li >= c->n_layers || off == 0) {
        snprintf(m->err, sizeof m->err, "unknown tensor %s", name);
        return -2;
    }
    nlo_layer *l = &m->layers[li];
    const char *s = name + off;
    nlo_mat mat = { data, type };
    if (!strcmp(s, "attn_norm.weight")) { l->attn_norm = to_f32(data, type, c->dim); return l->attn_norm ? 0 : -1; }
    if (!strcmp(s, "ffn_norm.weight")) { l->ffn_norm = to_f32(data, type, c->dim); return l->ffn_norm ? 0 : -1; }
    if (!strcmp(s, "attn_q.weight")) { l->wq = mat; return 0; }
    if (!strcmp(s, "attn_k.weight")) { l->wk = mat; return 0; }
    if (!strcmp(s, "attn_v.weight")) { l->wv = mat; return 0; }
    if (!strcmp(s, "attn_output.weight")) { l->wo = mat; return 0; }
    if (!strcmp(s, "ffn_gate.weight")) { l->wgate = mat; return 0; }
    if (!strcmp(s, "ffn_up.weight")) { l->wup = mat; return 0; }
    if (!strcmp(s, "ffn_down.weight")) { l->wdown = mat; return 0; }
    if (!strcmp(s, "attn_q.bias")) { l->bq = to_f32(data, type, c->n_heads * c->head_dim); return 0; }
    if (!strcmp(s, "attn_k.bias")) { l->bk = to_f32(data, type, kvdim); return 0; }
    if (!strcmp(s, "attn_v.bias")) { l->bv = to_f32(data, type, kvdim); return 0; }
    if (!strcmp(s, "attn_output.bias")) { l->bo = to_f32(data, type, c->dim); return 0; }
    snprintf(m->err, sizeof m->err, "unknown tensor %s", name);
    return -2;
}

/* allocState go/model.go:324-343 + precomputeRoPE :346-358 + the tied
 * embedding fallback :195-201. */
NLO_API int nlo_finalize(nlo_model *m) {
    const nlo_config *c = &m->cfg;
    if (!m->token_embd.w) { snprintf(m->err, sizeof m->err, "token_embd.weight: tensor not found"); return -1; }
    if (!m->output_norm) { snprintf(m->err, sizeof m->err, "output_norm.weight: tensor not found"); return -1; }
    if (!m->output.w) m->output = m->token_embd;
    for (int i = 0; i < c->n_layers; i++) {
        nlo_layer *l = &m->layers[i];
        if (!l->attn_norm || !l->ffn_norm || !l->wq.w || !l->wk.w || !l->wv.w || !l->wo.w ||
            !l->wgate.w || !l->wup.w || !l->wdown.w) {
            snprintf(m->err, sizeof m->err, "layer %d: tensor not found", i);
            return -1;
        }
    }
    int kvdim = c->n_kv_heads * c->head_dim, half = c->head_dim / 2;
    size_t kvn = (size_t)c->n_layers * c->seq_len * kvdim;
    m->x = calloc(c->dim, 4); m->xb = calloc(c->dim, 4); m->xb2 = calloc(c->dim, 4);
    m->hb = calloc(c->interm, 4); m->hb2 = calloc(c->interm, 4);
    m->q = calloc((size_t)c->n_heads * c->head_dim, 4); m->k = calloc(kvdim, 4); m->v = calloc(kvdim, 4);
    m->att = calloc((size_t)c->n_heads * c->seq_len, 4);
    m->logits = calloc(c->vocab, 4);
    m->key_cache = calloc(kvn, 4); m->value_cache = calloc(kvn, 4);
    m->cos_cache = calloc((size_t)c->seq_len * half, 4); m->sin_cache = calloc((size_t)c->seq_len * half, 4);
    m->emb_buf = calloc(c->dim, 4);
    double theta = (double)c->rope_theta;
    for (int pos = 0; pos < c->seq_len; pos++)
        for (int i = 0; i < half; i++) {
            double freq = 1.0 / pow(theta, (double)(2 * i) / (double)c->head_dim);
            double angle = (double)pos * freq;
            m->cos_cache[pos * half + i] = (float)cos(angle);
            m->sin_cache[pos * half + i] = (float)sin(angle);
        }
    m->finalized = 1;
    return 0;
}

NLO_API void nlo_destroy(nlo_model *m) {
    if (!m) return;
    for (int i = 0; i < m->cfg.n_layers; i++) {
        nlo_layer *l = &m->layers[i];
        free(l->attn_norm); free(l->ffn_norm); free(l->bq); free(l->bk); free(l->bv); free(l->bo);
    }
    free(m->layers); free(m->output_norm);
    free(m->x); free(m->xb); free(m->xb2); free(m->hb); free(m->hb2); free(m->q); free(m->k); free(m->v);
    free(m->att); free(m->logits); free(m->key_cache); free(m->value_cache);
    free(m->cos_cache); free(m->sin_cache); free(m->emb_buf);
    free(m->gamma_idx); free(m->gamma_val);
    free(m);
}

/* embedLookupInto go/model.go:389-446 */
NLO_API void nlo_embed_lookup(float *out, const void *data, uint32_t type, int token, int dim) {
    h2f_init();
    int be = nlo_block_elems(type), bb = nlo_block_bytes(type);
    if (bb == 0) { for (int i = 0; i < dim; i++) out[i] = 0.0f; return; }
    const uint8_t *row = (const uint8_t *)data + (int64_t)token * (dim / be) * bb;
    nlo_dequant(out, row, type, dim);
}

/* applyRoPE go/model.go:449-461 / applyRoPEConjugate :465-477 */
static void rope(float *vec, int pos, const nlo_model *m, int conj) {
    int half = m->cfg.head_dim / 2;
    const float *cc = m->cos_cache + (size_t)pos * half, *ss = m->sin_cache + (size_t)pos * half;
    for (int i = 0; i < half; i++) {
        float x0 = vec[i], x1 = vec[i + half], c = cc[i], si = ss[i];
        if (!conj) {
            float a = x0 * c, b = x1 * si; vec[i] = a - b;
            float d = x0 * si, e = x1 * c; vec[i + half] = d + e;
        } else {
            float a = x0 * c, b = x1 * si; vec[i] = a + b;
            float d = -x0 * si, e = x1 * c; vec[i + half] = d + e;
        }
    }
}

static void add_bias(float *out, const float *b, int n) {
    if (!b) return;
    for (int i = 0; i < n; i++) out[i] = out[i] + b[i];
}

/* Forward go/model.go:490-620 */
NLO_API void nlo_forward(nlo_model *m, int token, int pos) {
    const nlo_config *c = &m->cfg;
    int dim = c->dim, hd = c->head_dim, kvdim = c->n_kv_heads * hd;
    int group = c->n_heads / c->n_kv_heads;

    nlo_embed_lookup(m->emb_buf, m->token_embd.w, m->token_embd.type, token, dim);
    /* ApplyToEmbedding go/gamma.go:272-290 (go/model.go:503-505); the Go map keeps the LAST duplicate index */
    for (int gi = m->gamma_n - 1; gi >= 0; gi--)
        if (m->gamma_idx[gi] == token) {
            for (int i = 0; i < dim; i++) m->emb_buf[i] = m->emb_buf[i] + m->gamma_val[(size_t)gi * dim + i];
            break;
        }
    memcpy(m->x, m->emb_buf, (size_t)dim * 4);

    float attn_scale = (float)(1.0 / sqrt((double)hd));

    for (int layer = 0; layer < c->n_layers; layer++) {
        nlo_layer *l = &m->layers[layer];
        nlo_rmsnorm_into(m->xb, m->x, l->attn_norm, dim, c->eps);

        nlo_matmul(m->q, l->wq.w, l->wq.type, m->xb, c->n_heads * hd, dim);
        nlo_matmul(m->k, l->wk.w, l->wk.type, m->xb, kvdim, dim);
        nlo_matmul(m->v, l->wv.w, l->wv.type, m->xb, kvdim, dim);
        add_bias(m->q, l->bq, c->n_heads * hd);
        add_bias(m->k, l->bk, kvdim);
        add_bias(m->v, l->bv, kvdim);

        for (int h = 0; h < c->n_heads; h++) rope(m->q + h * hd, pos, m, c->rope_conjugate);
        for (int h = 0; h < c->n_kv_heads; h++) rope(m->k + h * hd, pos, m, c->rope_conjugate);

        if (c->qk_norm) {
            for (int h = 0; h < c->n_heads; h++) nlo_rmsnorm_bare(m->q + h * hd, hd, c->eps);
            for (int h = 0; h < c->n_kv_heads; h++) nlo_rmsnorm_bare(m->k + h * hd, hd, c->eps);
        }

        size_t loff = (size_t)layer * c->seq_len * kvdim;
        memcpy(m->key_cache + loff + (size_t)pos * kvdim, m->k, (size_t)kvdim * 4);
        memcpy(m->value_cache + loff + (size_t)pos * kvdim, m->v, (size_t)kvdim * 4);

        for (int h = 0; h < c->n_heads; h++) {
            int kvh = h / group;
            const float *qh = m->q + h * hd;
            float *att = m->att + (size_t)h * c->seq_len;
            for (int t = 0; t <= pos; t++) {
                const float *kp = m->key_cache + loff + (size_t)t * kvdim + kvh * hd;
                float dot = 0.0f;
                for (int d = 0; d < hd; d++) { float p = qh[d] * kp[d]; dot = dot + p; }
                att[t] = dot * attn_scale;
            }
            nlo_softmax(att, pos + 1);
            float *o = m->xb2 + h * hd;
            for (int d = 0; d < hd; d++) o[d] = 0.0f;
            for (int t = 0; t <= pos; t++) {
                float a = att[t];
                const float *vp = m->value_cache + loff + (size_t)t * kvdim + kvh * hd;
                for (int d = 0; d < hd; d++) { float p = a * vp[d]; o[d] = o[d] + p; }
            }
        }

        nlo_matmul(m->xb, l->wo.w, l->wo.type, m->xb2, dim, dim);
        add_bias(m->xb, l->bo, dim);
        for (int i = 0; i < dim; i++) m->x[i] = m->x[i] + m->xb[i];

        nlo_rmsnorm_into(m->xb, m->x, l->ffn_norm, dim, c->eps);
        nlo_matmul(m->hb, l->wgate.w, l->wgate.type, m->xb, c->interm, dim);
        nlo_matmul(m->hb2, l->wup.w, l->wup.type, m->xb, c->interm, dim);
        for (int i = 0; i < c->interm; i++) m->hb[i] = nlo_silu(m->hb[i]) * m->hb2[i];
        nlo_matmul(m->xb, l->wdown.w, l->wdown.type, m->hb, dim, c->interm);
        for (int i = 0; i < dim; i++) m->x[i] = m->x[i] + m->xb[i];
    }

    nlo_rmsnorm(m->x, m->output_norm, dim, c->eps);
    nlo_matmul(m->logits, m->output.w, m->output.type, m->x, c->vocab, dim);
}

/* Reset go/model.go:623-631 */
NLO_API void nlo_reset(nlo_model *m) {
    const nlo_config *c = &m->cfg;
    size_t kvn = (size_t)c->n_layers * c->seq_len * c->n_kv_heads * c->head_dim;
    memset(m->key_cache, 0, kvn * 4);
    memset(m->value_cache, 0, kvn * 4);
}

NLO_API void nlo_set_gamma(nlo_model *m, const int32_t *idx, int n, const float *values) {
    free(m->gamma_idx); free(m->gamma_val);
    m->gamma_idx = NULL; m->gamma_val = NULL; m->gamma_n = n;
    if (n <= 0) { m->gamma_n = 0; return; }
    m->gamma_idx = (int32_t *)malloc((size_t)n * 4);
    m->gamma_val = (float *)malloc((size_t)n * m->cfg.dim * 4);
    memcpy(m->gamma_idx, idx, (size_t)n * 4);
    memcpy(m->gamma_val, values, (size_t)n * m->cfg.dim * 4);
}

NLO_API float *nlo_logits(nlo_model *m) { return m->logits; }
NLO_API const nlo_config *nlo_get_config(nlo_model *m) { return &m->cfg; }
NLO_API float *nlo_state_buffer(nlo_model *m, const char *which) {
    if (!strcmp(which, "x")) return m->x;
    if (!strcmp(which, "q")) return m->q;
    if (!strcmp(which, "k")) return m->k;
    if (!strcmp(which, "v")) return m->v;
    if (!strcmp(which, "xb2")) return m->xb2;
    if (!strcmp(which, "hb")) return m->hb;
    if (!strcmp(which, "key_cache")) return m->key_cache;
    if (!strcmp(which, "value_cache")) return m->value_cache;
    if (!strcmp(which, "cos")) return m->cos_cache;
    if (!strcmp(which, "sin")) return m->sin_cache;
    return NULL;
}

/* Engine.Generate go/main.go:152-230 restricted to the greedy configuration
 * the parity runs use (--temp 0 --rep-penalty 1.0): Reset, token-at-a-time
 * prefill that stops at pos >= SeqLen-1 (:160-166), then argmax / EOS check /
 * Forward (:173-219).  Returns the number of ids written to out_ids (the
 * sampled ids, including a final EOS if one was drawn).  If logits_out is not
 * NULL it receives the logits each id was sampled from (n_out x vocab). */
NLO_API int nlo_generate_greedy(nlo_model *m, const int32_t *prompt, int n_prompt, int max_tokens,
                                int eos_id, int32_t *out_ids, float *logits_out) {
    const nlo_config *c = &m->cfg;
    nlo_reset(m);
    int pos = 0;
    for (int i = 0; i < n_prompt; i++) {
        nlo_forward(m, prompt[i], pos);
        pos++;
        if (pos >= c->seq_len - 1) break;
    }
    int n = 0;
    for (int i = 0; i < max_tokens; i++) {
        if (logits_out) memcpy(logits_out + (size_t)n * c->vocab, m->logits, (size_t)c->vocab * 4);
        int next = nlo_argmax(m->logits, c->vocab);
        out_ids[n++] = next;
        if (next == eos_id) break;
        nlo_forward(m, next, pos);
        pos++;
        if (pos >= c->seq_len) break;
    }
    return n;
}
