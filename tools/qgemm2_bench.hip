// qgemm2_bench.hip -- developer microbenchmark: the weights-through-LDS multi-token GEMM (nl_qgemm2.h) next to
// qgemm_kernel (nl_qgemm.h) on random packed weights, same shapes as tools/qgemm_bench.hip, plus max |difference|.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -I nanollama_amd/csrc tools/qgemm2_bench.hip -o /tmp/qgemm2_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
#include "nl_qgemm2.h"
using namespace nl;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void fill_u32(uint32_t *p, size_t n, uint32_t seed, uint32_t andm, uint32_t orm) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = (x & andm) | orm;
    }
}

static const char *g_only = nullptr;   // argv[3]: run only the shapes whose name contains this
template <int WT>
int run(int N, int iters) {
    constexpr int CPP = WFrag<WT>::CPP;
    struct Shape { const char *name; int rows, cols; };
    std::vector<Shape> shapes = {{"goldie gate", 4096, 1536}, {"goldie down", 1536, 4096}, {"mini gate", 2048, 768}, {"mini down", 768, 2048},
                                 {"big gate", 11008, 4096}, {"mini qkv", 1152, 768}, {"mini wo", 768, 768}};
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (auto &sh : shapes) {
        if (g_only && !strstr(sh.name, g_only)) continue;
        const int ntiles = sh.rows / 16, npairs = sh.cols / 64;
        const size_t qbytes = (size_t)ntiles * npairs * CPP * TR * 16, swords = (size_t)ntiles * npairs * TR;
        const int copies = (int)std::max<size_t>(2, ((size_t)600 << 20) / qbytes);
        uint8_t *q; uint32_t *s; uint4 *xf; float *out, *out2;
        CK(hipMalloc(&q, qbytes * copies)); CK(hipMalloc(&s, swords * 4 * copies));
        const size_t nxf = xfrag_uint4(sh.cols, N);
        CK(hipMalloc(&xf, nxf * 16)); CK(hipMalloc(&out, (size_t)N * sh.rows * 4)); CK(hipMalloc(&out2, (size_t)N * sh.rows * 4));
        fill_u32<<<2048, 256, 0, st>>>((uint32_t *)q, qbytes * copies / 4, 1, 0xffffffffu, 0);
        fill_u32<<<2048, 256, 0, st>>>(s, swords * copies, 2, 0x03ff03ffu, 0x20002000u);     // fp16 scales ~2^-7
        fill_u32<<<2048, 256, 0, st>>>((uint32_t *)xf, nxf * 4, 3, 0x03ff03ffu, 0x30003000u); // fp16 values ~0.1
        const int tok_tiles = (N + QG_TOK - 1) / QG_TOK;
        QGemmParams P{};
        P.rows = sh.rows; P.cols = sh.cols; P.npairs = npairs; P.ntiles = ntiles; P.xf = xf; P.nt16 = tok_tiles * 4;
        P.n_tokens = N; P.ldo = sh.rows; P.ksplit = 1;
        const dim3 g1((ntiles + QG_WAVES - 1) / QG_WAVES, tok_tiles, 1);
        float ms1 = 0;
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(a, st));
            for (int i = 0; i < iters; i++) {
                P.q = q + (size_t)(i % copies) * qbytes; P.s = s + (size_t)(i % copies) * swords; P.out = out;
                hipLaunchKernelGGL((qgemm_kernel<WT, QG_WAVES, 1, QG_EPI_PLAIN>), g1, dim3(QG_WAVES * 64), 0, st, P);
            }
            CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms1, a, b));
        }
        const double fl = 2.0 * sh.rows * sh.cols * N;
        printf("%-12s %5dx%-5d N=%-4d  qgemm %8.2f us %6.1f TF |", sh.name, sh.rows, sh.cols, N, ms1 * 1e3 / iters, fl / (ms1 * 1e3 / iters) * 1e-6);
        std::vector<float> h1((size_t)N * sh.rows), h2((size_t)N * sh.rows);
        CK(hipMemcpy(h1.data(), out, h1.size() * 4, hipMemcpyDeviceToHost));
        auto variant = [&](const char *tag, auto kern, int RT, int NTW, int WAVES) {
            const dim3 g2((sh.rows + RT * 16 - 1) / (RT * 16), (N + WAVES * NTW * 16 - 1) / (WAVES * NTW * 16), 1);
            float ms2 = 0;
            for (int rep = 0; rep < 2; rep++) {
                CK(hipEventRecord(a, st));
                for (int i = 0; i < iters; i++) {
                    P.q = q + (size_t)(i % copies) * qbytes; P.s = s + (size_t)(i % copies) * swords; P.out = out2;
                    hipLaunchKernelGGL(kern, g2, dim3(WAVES * 64), 0, st, P);
                }
                CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms2, a, b));
            }
            CK(hipGetLastError());
            CK(hipMemcpy(h2.data(), out2, h2.size() * 4, hipMemcpyDeviceToHost));
            double worst = 0;
            for (size_t i = 0; i < h1.size(); i++) worst = std::max(worst, (double)std::fabs(h1[i] - h2[i]));
            printf(" %s %7.2f us x%.2f%s |", tag, ms2 * 1e3 / iters, ms1 / ms2, worst == 0 ? "" : " DIFF");
        };
#ifdef QG2_STAMPS
#ifdef QG2_STAMP_414
        variant("4/1/4", qgemm2_kernel<WT, 4, 1, 4>, 4, 1, 4);
#else
        variant("4/2/8", qgemm2_kernel<WT, 4, 2, 8>, 4, 2, 8);
#endif
#else
        variant("3/1/4", qgemm2_kernel<WT, 3, 1, 4>, 3, 1, 4);
        variant("3/2/4", qgemm2_kernel<WT, 3, 2, 4>, 3, 2, 4);
        variant("6/1/4", qgemm2_kernel<WT, 6, 1, 4>, 6, 1, 4);
        variant("2/1/4", qgemm2_kernel<WT, 2, 1, 4>, 2, 1, 4);
        variant("4/1/2", qgemm2_kernel<WT, 4, 1, 2>, 4, 1, 2);
        variant("8/1/4", qgemm2_kernel<WT, 8, 1, 4>, 8, 1, 4);
        variant("16/1/4", qgemm2_kernel<WT, 16, 1, 4>, 16, 1, 4);
        variant("16/1/2", qgemm2_kernel<WT, 16, 1, 2>, 16, 1, 2);
        variant("8/2/2", qgemm2_kernel<WT, 8, 2, 2>, 8, 2, 2);
        variant("8/1/8", qgemm2_kernel<WT, 8, 1, 8>, 8, 1, 8);
        variant("4/1/4", qgemm2_kernel<WT, 4, 1, 4>, 4, 1, 4);
        variant("2/2/8", qgemm2_kernel<WT, 2, 2, 8>, 2, 2, 8);
        variant("4/2/8", qgemm2_kernel<WT, 4, 2, 8>, 4, 2, 8);
        variant("4/2/4", qgemm2_kernel<WT, 4, 2, 4>, 4, 2, 4);
        variant("4/1/8", qgemm2_kernel<WT, 4, 1, 8>, 4, 1, 8);
#endif
        printf("\n");
#ifdef QG2_STAMPS
        {   // phase stamps of one wavefront of the LAST variant run above (cycles since its first stamp)
            std::vector<long long> st_(4096);
            CK(hipMemcpyFromSymbol(st_.data(), HIP_SYMBOL(g_qg2_stamps), 4096 * 8));
            const int nch = (sh.cols / 32 + 3) / 4;
            printf("   stamps: start 0");
            for (int c = 0; c < nch; c++) {
                printf("\n   chunk %2d: loads issued %6lld | blocks", c, st_[1 + c * 8] - st_[0]);
                for (int b2 = 0; b2 < 4; b2++) printf(" %6lld", st_[2 + c * 8 + b2] - st_[0]);
                printf(" | staged %6lld | barrier %6lld", st_[6 + c * 8] - st_[0], st_[7 + c * 8] - st_[0]);
            }
            printf("\n   epilogue start %6lld end %6lld\n", st_[1 + nch * 8] - st_[0], st_[2 + nch * 8] - st_[0]);
        }
#endif
        CK(hipFree(q)); CK(hipFree(s)); CK(hipFree(xf)); CK(hipFree(out)); CK(hipFree(out2));
    }
    return 0;
}

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 2047, iters = argc > 2 ? atoi(argv[2]) : 50;
    if (argc > 3) g_only = argv[3];
    printf("== Q4_0\n"); run<WT_Q4_0>(N, iters);
    if (!(argc > 4 && atoi(argv[4]) == 4)) { printf("== Q8_0\n"); run<WT_Q8_0>(N, iters); }
    return 0;
}
