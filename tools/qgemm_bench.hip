// qgemm_bench.hip -- developer microbenchmark of the multi-token MFMA kernel (nl_qgemm.h) on random packed
// weights: time per launch for the GEMM shapes of a tier at N tokens, cycling through enough weight copies
// that every launch streams from HBM.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize
//        -I nanollama_amd/csrc tools/qgemm_bench.hip -o /tmp/qgemm_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "nl_qgemm.h"
using namespace nl;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void fill_u32(uint32_t *p, size_t n, uint32_t seed, uint32_t andm, uint32_t orm) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = (x & andm) | orm;
    }
}

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 64;
    const int force_ks = argc > 2 ? atoi(argv[2]) : 0;
    const int only = argc > 3 ? atoi(argv[3]) : -1;   // shape index, -1 = all
    const int iters = argc > 4 ? atoi(argv[4]) : 200;
    struct Shape { const char *name; int rows, cols; };
    std::vector<Shape> shapes = {{"goldie qkv", 2304, 1536}, {"goldie wo", 1536, 1536}, {"goldie gate", 4096, 1536},
                                 {"goldie down", 1536, 4096}, {"mini gate", 2048, 768}, {"mini down", 768, 2048},
                                 {"big gate", 11008, 4096}, {"mini qkv", 1152, 768}, {"mini wo", 768, 768}};
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (size_t si = 0; si < shapes.size(); si++) {
        if (only >= 0 && (int)si != only) continue;
        auto &sh = shapes[si];
        const int ntiles = sh.rows / 16, npairs = sh.cols / 64;
        const size_t qbytes = (size_t)ntiles * npairs * 2 * TR * 16, swords = (size_t)ntiles * npairs * TR;
        const int copies = (int)std::max<size_t>(2, ((size_t)600 << 20) / qbytes);
        uint8_t *q; uint32_t *s; uint4 *xf; float *out, *part;
        CK(hipMalloc(&q, qbytes * copies)); CK(hipMalloc(&s, swords * 4 * copies));
        const size_t nxf = xfrag_uint4(sh.cols, N);
        CK(hipMalloc(&xf, nxf * 16)); CK(hipMalloc(&out, (size_t)N * sh.rows * 4)); CK(hipMalloc(&part, (size_t)16 * N * sh.rows * 4));
        fill_u32<<<2048, 256, 0, st>>>((uint32_t *)q, qbytes * copies / 4, 1, 0xffffffffu, 0);
        fill_u32<<<2048, 256, 0, st>>>(s, swords * copies, 2, 0x03ff03ffu, 0x20002000u);     // fp16 scales ~2^-7
        fill_u32<<<2048, 256, 0, st>>>((uint32_t *)xf, nxf * 4, 3, 0x03ff03ffu, 0x30003000u); // fp16 values ~0.1
        const int row_groups = (ntiles + QG_WAVES * QG_RT - 1) / (QG_WAVES * QG_RT), tok_tiles = (N + QG_TOK - 1) / QG_TOK;
        const int nchunks = (sh.cols / 32 + QG_KC - 1) / QG_KC;
        int ks = 1;
        while (row_groups * tok_tiles * ks < 128 && ks * 2 <= nchunks && ks < 16) ks *= 2;
        if (force_ks) ks = force_ks;
        QGemmParams P{};
        P.rows = sh.rows; P.cols = sh.cols; P.npairs = npairs; P.ntiles = ntiles; P.xf = xf; P.nt16 = tok_tiles * 4;
        P.n_tokens = N; P.out = out; P.ldo = sh.rows; P.ksplit = ks; P.part = part;
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(a, st));
            for (int i = 0; i < iters; i++) {
                P.q = q + (size_t)(i % copies) * qbytes; P.s = s + (size_t)(i % copies) * swords;
                hipLaunchKernelGGL((qgemm_kernel<WT_Q4_0, QG_WAVES, QG_RT, QG_EPI_PLAIN>), dim3(row_groups, tok_tiles, ks), dim3(QG_WAVES * 64), 0, st, P);
                if (ks > 1) {
                    const long long count = (long long)N * sh.rows;
                    hipLaunchKernelGGL(qgemm_sum_kernel, dim3((unsigned)std::min<long long>((count + 255) / 256, 2048)), dim3(256), 0, st,
                                       part, ks, count, (const float *)nullptr, out, (const float *)nullptr, sh.rows);
                }
            }
            CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
        }
        const double us = ms * 1e3 / iters, wb = (double)qbytes + swords * 4;
        printf("%-12s %5dx%-5d N=%-4d grid %3dx%dx%-2d  %8.2f us/launch  weights %.2f TB/s  %.1f TFLOP/s\n", sh.name, sh.rows, sh.cols, N,
               row_groups, tok_tiles, ks, us, wb / us * 1e-6, 2.0 * sh.rows * sh.cols * N / us * 1e-6);
        CK(hipFree(q)); CK(hipFree(s)); CK(hipFree(xf)); CK(hipFree(out)); CK(hipFree(part));
    }
    return 0;
}
