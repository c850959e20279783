// Developer probe: what does flushing per-workgroup LDS histograms into ONE global histogram with device-scope 64-bit atomics cost?
// nb workgroups each add `na` (value != 0) buckets out of 2048, all hitting the SAME na addresses (the sampler's level-1 histogram:
// a flat distribution populates ~150 buckets); time per launch, back to back.
// Build + run (gpurun):  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/atomic_probe.hip -o /tmp/ap && /tmp/ap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void flush(unsigned long long *h, int na, int spread) {
    // thread t < na adds to bucket (t * spread) % 2048
    const int t = threadIdx.x;
    if (t < na) atomicAdd(&h[(t * spread) & 2047], (unsigned long long)(blockIdx.x + 1));
}
__global__ void nothing(unsigned long long *h) { if (threadIdx.x == 4096) h[0] = 1; }
int main() {
    unsigned long long *h; CK(hipMalloc(&h, 2048 * 8)); CK(hipMemset(h, 0, 2048 * 8));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto time = [&](auto launch) { float best = 1e9; for (int r = 0; r < 3; r++) { CK(hipEventRecord(a)); for (int i = 0; i < 200; i++) launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); best = ms < best ? ms : best; } return best * 1000.f / 200; };
    printf("empty launch of 125 x 256: %.2f us\n", time([&] { hipLaunchKernelGGL(nothing, dim3(125), dim3(256), 0, 0, h); }));
    for (int nb : {8, 32, 125, 500})
        for (int na : {32, 150, 256})
            printf("%3d workgroups x %3d atomics on the same %3d addresses: %.2f us per launch\n", nb, na, na, time([&] { hipLaunchKernelGGL(flush, dim3(nb), dim3(256), 0, 0, h, na, 7); }));
    return 0;
}
