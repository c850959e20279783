// Developer probe: where does the sort-free top-p selection (samp_select_radix_kernel, nanollama_amd/csrc/nl_sample.h)
// spend its time?  Runs the kernel on 32000 synthetic p = exp((l - max) / 0.8), l ~ N(0, sigma), alone in a loop (event
// time per launch) and prints the shader-clock stamps of thread 0 (100 MHz constant clock -> x 10 ns).
// Build + run (gpurun):  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DNL_SAMP_STAMPS -Inanollama_amd/csrc tools/samp_probe.hip -o /tmp/sp && /tmp/sp
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "nl_sample.h"
using namespace nl;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main(int argc, char **argv) {
    const int V = 32000;
    const float sigma = argc > 1 ? (float)atof(argv[1]) : 1.0f;
    std::mt19937 rng(5);
    std::normal_distribution<float> nd(0.f, sigma);
    std::vector<float> l(V), p(32768, 0.f);
    float mx = -1e30f;
    for (auto &v : l) { v = nd(rng); mx = std::max(mx, v); }
    for (int i = 0; i < V; i++) p[i] = (float)std::exp((double)((l[i] - mx) / 0.8f));
    SampleParams P{};
    float *keys, *uni, *dl, *pmax; int *ctl, *ids, *recent, *recent_n; unsigned long long *h1g;
    CK(hipMalloc(&keys, 32768 * 4)); CK(hipMalloc(&uni, 4096 * 4)); CK(hipMalloc(&ctl, 64)); CK(hipMalloc(&ids, 4096 * 4));
    CK(hipMalloc(&recent, 4096)); CK(hipMalloc(&recent_n, 4)); CK(hipMalloc(&dl, V * 4)); CK(hipMalloc(&pmax, 4)); CK(hipMalloc(&h1g, 2048 * 8));
    CK(hipMemcpy(keys, p.data(), 32768 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dl, l.data(), V * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(pmax, &mx, 4, hipMemcpyHostToDevice));
    P.logits = dl; P.pmax = pmax; P.nblocks_pen = 1; P.h1g = h1g;
    std::vector<float> u(4096); for (auto &x : u) x = (float)(rng() >> 8) / 16777216.0f;
    CK(hipMemcpy(uni, u.data(), 4096 * 4, hipMemcpyHostToDevice));
    CK(hipMemset(ctl, 0, 64)); CK(hipMemset(recent_n, 0, 4));
    P.vocab = V; P.temp = 0.8f; P.top_p = 0.9f; P.top_k = 50; P.rep_penalty = 1.15f; P.recent = recent; P.recent_n = recent_n; P.rep_window = 64;
    P.uniforms = uni; P.ctl = ctl; P.ids = ids; P.keys_in = keys; P.radix = 1;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(a));
        for (int i = 0; i < 200; i++) {
            CK(hipMemsetAsync(h1g, 0, 2048 * 8, 0));          // (samp_penalty_kernel's job in the engine)
            hipLaunchKernelGGL(samp_prob_hist_kernel, dim3((V + SAMP_THREADS - 1) / SAMP_THREADS), dim3(SAMP_THREADS), 0, 0, P);
            hipLaunchKernelGGL(samp_select_radix_kernel<32>, dim3(1), dim3(SAMP_THREADS), 0, 0, P);
        }
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("sigma %.2f: %.2f us per (memset + samp_prob_hist_kernel + samp_select_radix_kernel), back to back\n", sigma, ms * 1000.f / 200);
    }
    unsigned long long st[32];
    CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_samp_stamps), sizeof(st)));
    const char *name[16] = {"entry", "loads issued", "(-)", "keys in", "cut: scan 1", "cut: list", "cut: level 2", "cut: level 3",
                            "pick: scan 1", "pick: list", "pick: level 2", "pick: level 3", "pick done", "rank found", "end", ""};
    for (int i = 1; i < 15; i++) printf("  %-14s +%6.2f us (at %6.2f)\n", name[i], (double)(st[i] - st[i - 1]) * 0.01, (double)(st[i] - st[0]) * 0.01);
    std::vector<int> hid(8); CK(hipMemcpy(hid.data(), ids, 32, hipMemcpyDeviceToHost));
    printf("first picks: %d %d %d %d\n", hid[0], hid[1], hid[2], hid[3]);
    return 0;
}
