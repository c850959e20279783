// dgemm_bench.hip -- developer microbenchmark: the decode-batch GEMM (nl_dgemm.h) on goldie's four projection shapes at N tokens,
// every epilogue with real operands (RoPE tables, KV cache, residual, folded norm), random packed weights rotated through
// > 256 MB so that no launch finds its weights in the Infinity Cache.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -I nanollama_amd/csrc tools/dgemm_bench.hip -o /tmp/dgemm_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
#include "nl_dgemm.h"
using namespace nl;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void fill_u32(uint32_t *p, size_t n, uint32_t seed, uint32_t andm, uint32_t orm) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = (x & andm) | orm;
    }
}
__global__ void fill_f32(float *p, size_t n, float v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void fill_f64(double *p, size_t n, double v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void fill_tkv(long long *p, int n, long long stream_stride, long long pos_mul) { for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = i * stream_stride + i * pos_mul; }
__global__ void fill_iota(int *p, int n, int mul) { for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = i * mul; }

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 64, iters = argc > 2 ? atoi(argv[2]) : 200;
    const int D = 1536, I = 4096, H = 24, KV = 6, hd = 64, seq = 2048, R = (H + 2 * KV) * hd;
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    struct Mat { uint8_t *q; uint32_t *s; size_t qbytes, swords; int rows, cols, ntiles, npairs, copies; };
    auto mk = [&](int rows, int cols) {
        Mat m{}; m.rows = rows; m.cols = cols; m.ntiles = rows / 16; m.npairs = cols / 64;
        m.qbytes = (size_t)m.ntiles * m.npairs * 2 * TR * 16; m.swords = (size_t)m.ntiles * m.npairs * TR;
        m.copies = getenv("DG_COPIES") ? atoi(getenv("DG_COPIES")) : (int)std::max<size_t>(2, ((size_t)300 << 20) / m.qbytes);   // DG_COPIES=1: the same weights every launch (L2 / Infinity Cache warm)
        CK(hipMalloc(&m.q, m.qbytes * m.copies)); CK(hipMalloc(&m.s, m.swords * 4 * m.copies));
        fill_u32<<<2048, 256, 0, st>>>((uint32_t *)m.q, m.qbytes * m.copies / 4, 1, 0xffffffffu, 0);
        fill_u32<<<2048, 256, 0, st>>>(m.s, m.swords * m.copies, 2, 0x03ff03ffu, 0x20002000u);     // fp16 scales ~2^-7
        return m;
    };
    Mat qkv = mk(R, D), wo = mk(D, D), gate = mk(I, D), up = mk(I, D), down = mk(D, I);
    const int nt16 = ((N + 63) / 64) * 4;
    const size_t nxf = xfrag_uint4(I, N);
    uint4 *xf, *xf2; CK(hipMalloc(&xf, nxf * 16)); CK(hipMalloc(&xf2, nxf * 16));
    fill_u32<<<2048, 256, 0, st>>>((uint32_t *)xf, nxf * 4, 3, 0x03ff03ffu, 0x30003000u); // fp16 values ~0.1
    fill_u32<<<2048, 256, 0, st>>>((uint32_t *)xf2, nxf * 4, 4, 0x03ff03ffu, 0x30003000u);
    float *x, *qout, *kc, *vc, *cs, *sn, *nw, *sc1, *sc2, *tcs, *tsn; double *ssq; int *pos, *strm; long long *tkv;
    CK(hipMalloc(&x, (size_t)64 * D * 4)); CK(hipMalloc(&qout, (size_t)64 * H * hd * 4));
    const size_t kvs = (size_t)KV * seq * hd;
    CK(hipMalloc(&kc, kvs * 64 * 4)); CK(hipMalloc(&vc, kvs * 64 * 4));
    CK(hipMalloc(&cs, (size_t)seq * hd / 2 * 4)); CK(hipMalloc(&sn, (size_t)seq * hd / 2 * 4));
    CK(hipMalloc(&nw, (size_t)D * 4)); CK(hipMalloc(&sc1, 64 * 4)); CK(hipMalloc(&sc2, 64 * 4));
    CK(hipMalloc(&ssq, (size_t)64 * (D / 32) * 8)); CK(hipMalloc(&pos, 64 * 4)); CK(hipMalloc(&strm, 64 * 4));
    fill_f32<<<256, 256, 0, st>>>(x, (size_t)64 * D, 0.5f); fill_f32<<<256, 256, 0, st>>>(cs, (size_t)seq * hd / 2, 0.8f);
    fill_f32<<<256, 256, 0, st>>>(sn, (size_t)seq * hd / 2, 0.6f); fill_f32<<<8, 256, 0, st>>>(nw, D, 1.0f);
    fill_f32<<<1, 64, 0, st>>>(sc1, 64, 1.0f); fill_f32<<<1, 64, 0, st>>>(sc2, 64, 1.0f);
    fill_f64<<<16, 256, 0, st>>>(ssq, (size_t)64 * (D / 32), 8.0);
    fill_iota<<<1, 64, 0, st>>>(pos, 64, 3); fill_iota<<<1, 64, 0, st>>>(strm, 64, 1);
    CK(hipMalloc(&tcs, (size_t)64 * hd / 2 * 4)); CK(hipMalloc(&tsn, (size_t)64 * hd / 2 * 4)); CK(hipMalloc(&tkv, 64 * 8));
    fill_f32<<<8, 256, 0, st>>>(tcs, (size_t)64 * hd / 2, 0.8f); fill_f32<<<8, 256, 0, st>>>(tsn, (size_t)64 * hd / 2, 0.6f);
    fill_tkv<<<1, 64, 0, st>>>(tkv, 64, (long long)kvs, 3LL * hd);
    CK(hipStreamSynchronize(st));

    auto timeit = [&](const char *name, auto launch, double bytes, double flop) {
        float ms = 0;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(a, st));
            for (int i = 0; i < iters; i++) launch(i);
            CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
        }
        CK(hipGetLastError());
        const double us = ms * 1e3 / iters;
        printf("%-10s N=%-3d %8.2f us per launch (incl. boundary)  weights %6.2f TB/s  %6.1f TFLOP/s\n", name, N, us, bytes / us * 1e-6, flop / us * 1e-6);
    };
    auto base = [&](const Mat &m, int i) {
        QGemmParams P{};
        P.q = m.q + (size_t)(i % m.copies) * m.qbytes; P.s = m.s + (size_t)(i % m.copies) * m.swords;
        P.rows = m.rows; P.cols = m.cols; P.npairs = m.npairs; P.ntiles = m.ntiles; P.nt16 = nt16; P.n_tokens = N; P.ksplit = 1;
        return P;
    };
    auto stamps = [&](const char *name, int grid) {
#ifdef DG_STAMPS
        std::vector<long long> s_(64);
        CK(hipMemcpyFromSymbol(s_.data(), HIP_SYMBOL(g_dg_stamps), 64 * 8));
        printf("   %s: workgroup 9, wavefront 0, cycles since entry: issued", name);
        for (int i = 1; i < 40 && s_[i]; i++) printf(" %lld", s_[i] - s_[0]);
        if (!s_[40]) { printf("\n"); goto census; }
        printf(" | partials out + barrier %lld\n", s_[40] - s_[0]);
    census:
        std::vector<long long> c_(2 * 2048);
        CK(hipMemcpyFromSymbol(c_.data(), HIP_SYMBOL(g_dg_census), 2 * 2048 * 8));
        long long e0 = 1LL << 62, e1 = 0, x0 = 1LL << 62, x1 = 0; int nwg = 0;
        for (int i = 0; i < grid; i++) if (c_[2 * i] && c_[2 * i + 1]) { nwg++; e0 = std::min(e0, c_[2 * i]); e1 = std::max(e1, c_[2 * i]); x0 = std::min(x0, c_[2 * i + 1]); x1 = std::max(x1, c_[2 * i + 1]); }
        printf("   census: %d workgroups; entries spread %.2f us; first exit +%.2f us, last exit +%.2f us after the first entry; wg 9: entry +%.2f exit +%.2f\n",
               nwg, (e1 - e0) * 0.01, (x0 - e0) * 0.01, (x1 - e0) * 0.01, (c_[18] - e0) * 0.01, (c_[19] - e0) * 0.01);
        if (nwg > 256) {      // several rounds of workgroups: when they enter, how long one stays
            std::vector<double> ent, dur;
            for (int i = 0; i < grid; i++) if (c_[2 * i] && c_[2 * i + 1]) { ent.push_back((c_[2 * i] - e0) * 0.01); dur.push_back((c_[2 * i + 1] - c_[2 * i]) * 0.01); }
            std::sort(ent.begin(), ent.end()); std::sort(dur.begin(), dur.end());
            printf("   entries at (us):");
            for (int q = 0; q <= 8; q++) printf(" %.1f", ent[std::min(ent.size() - 1, ent.size() * q / 8)]);
            printf("; time in the kernel per workgroup: min %.2f median %.2f max %.2f us\n", dur.front(), dur[dur.size() / 2], dur.back());
        }
        std::vector<long long> z_(64, 0);
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_dg_stamps), z_.data(), 64 * 8));
#endif
    };
    const QGemmParams::NormIn nin{ssq, D / 32, D, 1e-5f, sc1, sc2};
    timeit("qkv+rope", [&](int i) {
        QGemmParams P = base(qkv, i);
        P.xf = xf; P.ldo = R; P.nrm_in = nin;
        P.rope = QGemmParams::Rope{pos, strm, cs, sn, qout, kc, vc, (long long)kvs, nullptr, nullptr, nullptr, hd, H, KV, seq, 0, tcs, tsn, tkv};
        CK(dg_launch_rope(P, st));
    }, (double)qkv.qbytes * 18 / 16, 2.0 * R * D * N);
    stamps("qkv", (int)(dg_grid(qkv.ntiles, 3, N).x * dg_grid(qkv.ntiles, 3, N).y));
    timeit("wo+norm", [&](int i) {
        QGemmParams P = base(wo, i);
        P.xf = xf; P.out = x; P.ldo = D; P.resid = x;
        P.nrm_out = QGemmParams::NormOut{nw, xf2, ssq, sc1};
        CK(dg_launch_plain(P, st));
    }, (double)wo.qbytes * 18 / 16, 2.0 * D * D * N);
    stamps("wo", (int)(dg_grid(wo.ntiles, 2, N).x * dg_grid(wo.ntiles, 2, N).y));
    timeit("gate|up", [&](int i) {
        QGemmParams P = base(gate, i);
        P.q1 = up.q + (size_t)(i % up.copies) * up.qbytes; P.s1 = up.s + (size_t)(i % up.copies) * up.swords;
        P.xf = xf; P.ldo = I; P.nrm_in = nin; P.xf_out = xf2; P.out_q4 = 1;
        CK(dg_launch_swiglu(P, st));
    }, (double)gate.qbytes * 2 * 18 / 16, 2.0 * 2 * I * D * N);
    stamps("gate|up", (int)(dg_grid(gate.ntiles, 4, N).x * dg_grid(gate.ntiles, 4, N).y));
    timeit("down+norm", [&](int i) {
        QGemmParams P = base(down, i);
        P.xf = xf2; P.out = x; P.ldo = D; P.resid = x;
        P.nrm_out = QGemmParams::NormOut{nw, xf, ssq, sc1};
        CK(dg_launch_plain(P, st));
    }, (double)down.qbytes * 18 / 16, 2.0 * D * I * N);
    stamps("down", (int)(dg_grid(down.ntiles, 2, N).x * dg_grid(down.ntiles, 2, N).y));
    {   // the LM head of the batch: 32000 rows, dghead_kernel (one resident workgroup per compute unit walks its row groups)
        const int V = 32000;
        Mat head = mk(V, D);
        float *logits; uint2 *cand;
        CK(hipMalloc(&logits, (size_t)64 * V * 4)); CK(hipMalloc(&cand, (size_t)64 * (V / 16) * 8));
        CK(hipStreamSynchronize(st));
        timeit("lm head", [&](int i) {
            QGemmParams P = base(head, i);
            P.xf = xf; P.out = logits; P.ldo = V; P.nrm_in = nin; P.part1 = reinterpret_cast<float *>(cand);
            CK(dg_launch_head(P, st));
        }, (double)head.qbytes * 18 / 16, 2.0 * V * D * N);
        stamps("lm head", 256);
        timeit("down+head", [&](int i) {      // as in a step: the head behind the producer of its fragments (subtract down+norm above)
            QGemmParams Q = base(down, i);
            Q.xf = xf2; Q.out = x; Q.ldo = D; Q.resid = x;
            Q.nrm_out = QGemmParams::NormOut{nw, xf, ssq, sc1};
            CK(dg_launch_plain(Q, st));
            QGemmParams P = base(head, i);
            P.xf = xf; P.out = logits; P.ldo = V; P.nrm_in = nin; P.part1 = reinterpret_cast<float *>(cand);
            CK(dg_launch_head(P, st));
        }, (double)head.qbytes * 18 / 16, 2.0 * V * D * N);
        {   // same products as the plain launch (no norm on either side)
            float *ref; CK(hipMalloc(&ref, (size_t)64 * V * 4));
            CK(hipMemsetAsync(ref, 0, (size_t)64 * V * 4, st)); CK(hipMemsetAsync(logits, 0, (size_t)64 * V * 4, st));
            QGemmParams P = base(head, 0);
            P.xf = xf; P.out = ref; P.ldo = V;
            CK(dg_launch_plain(P, st));
            P.out = logits; P.part1 = reinterpret_cast<float *>(cand);
            fill_f64<<<16, 256, 0, st>>>(ssq, (size_t)64 * (D / 32), 8.0);      // (the producer launches above left their own sums)
            if (getenv("DG_HEAD_NORM")) P.nrm_in = nin;       // ssq 8.0 x 48 partials, dim 1536: inv = 1 / sqrt(0.25 + 1e-5)
            CK(dg_launch_head(P, st));
            CK(hipStreamSynchronize(st));
            std::vector<float> a_((size_t)N * V), b_((size_t)N * V);
            CK(hipMemcpy(a_.data(), ref, a_.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b_.data(), logits, b_.size() * 4, hipMemcpyDeviceToHost));
            double worst = 0; size_t nbad = 0, first = (size_t)-1;
            if (getenv("DG_HEAD_NORM")) for (auto &v : a_) v *= (float)(1.0 / sqrt(384.0 / 1536.0 + (double)1e-5f));
            for (size_t i = 0; i < a_.size(); i++) { const double d = fabs((double)a_[i] - b_[i]); if (!(d <= 1e-3 * (1 + fabs(a_[i])))) { nbad++; if (first == (size_t)-1) first = i; } if (d > worst) worst = d; }
            printf("   lm head against the plain launch: max |diff| %.3g, %zu of %zu off", worst, nbad, a_.size());
            if (nbad) printf(" (first: token %zu row %zu: %g vs %g)", first / V, first % V, b_[first], a_[first]);
            printf("\n");
        }
    }
    return 0;
}
