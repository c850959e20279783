// l2_prefetch_probe.hip -- can a launch warm the NEXT launch's first weight bytes?  (round 5, big tier: DESIGN.md §3.4)
//
// reader: 256 workgroups x 1024 threads, workgroup b streams its own contiguous chunk with 16-byte loads (a decode GEMV's weight
// fetch).  toucher: workgroup b reads ONE dword per 128-byte line of the chunk of workgroup (b + shift) % 256 -- shift 0 puts
// the lines into the L2 of the XCD that will read them (block b -> XCD b % 8 in both launches), shift 1 into a neighbour's L2,
// i.e. only the memory-side Infinity Cache can serve the reader.  Between experiments a 600 MB sweep evicts everything.
// Reported: the reader's duration (HIP events) cold, after a same-XCD touch, after a cross-XCD touch, and re-run on itself.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(1024) reader(const uint4 *w, size_t chunk16, unsigned *out) {
    const uint4 *p = w + (size_t)blockIdx.x * chunk16;
    unsigned acc = 0;
    for (size_t i = threadIdx.x; i < chunk16; i += 4096) {
        uint4 a = p[i], b = i + 1024 < chunk16 ? p[i + 1024] : make_uint4(0, 0, 0, 0);
        uint4 c = i + 2048 < chunk16 ? p[i + 2048] : make_uint4(0, 0, 0, 0), d = i + 3072 < chunk16 ? p[i + 3072] : make_uint4(0, 0, 0, 0);
        acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}
__global__ void __launch_bounds__(1024) toucher(const unsigned *w, size_t chunk_bytes, int shift, unsigned *out) {
    const unsigned *p = w + (size_t)((blockIdx.x + shift) % gridDim.x) * (chunk_bytes / 4);
    unsigned acc = 0;
    for (size_t line = threadIdx.x; line < chunk_bytes / 128; line += 1024) acc ^= p[line * 32];
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}
__global__ void sweep(const uint4 *w, size_t n16, unsigned *out) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { uint4 a = w[i]; acc ^= a.x ^ a.y ^ a.z ^ a.w; }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const size_t total = (size_t)1 << 30;      // 1 GiB: [0, 256 MB) the chunks, [400 MB, 1 GiB) the eviction sweep
    uint4 *w; unsigned *out;
    CK(hipMalloc(&w, total)); CK(hipMalloc(&out, 4096));
    CK(hipMemset(w, 1, total));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto evict = [&]() { hipLaunchKernelGGL(sweep, dim3(2048), dim3(256), 0, 0, w + (400u << 20) / 16, (size_t)(600u << 20) / 16, out); };
    auto time_reader = [&](size_t chunk_bytes, float &us) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(reader, dim3(256), dim3(1024), 0, 0, w, chunk_bytes / 16, out);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); us = ms * 1e3f;
    };
    printf("chunk per workgroup | chip total | reader cold | after same-XCD touch | after cross-XCD touch | re-run | toucher alone (cold)\n");
    for (size_t kb : {36, 74, 110, 148, 300, 600}) {
        const size_t cb = kb * 1024;
        std::vector<float> cold, same, cross, rerun, tch;
        for (int rep = 0; rep < 7; rep++) {
            float us;
            evict(); CK(hipDeviceSynchronize()); time_reader(cb, us); cold.push_back(us);
            time_reader(cb, us); rerun.push_back(us);
            evict(); hipLaunchKernelGGL(toucher, dim3(256), dim3(1024), 0, 0, (const unsigned *)w, cb, 0, out); CK(hipDeviceSynchronize()); time_reader(cb, us); same.push_back(us);
            evict(); hipLaunchKernelGGL(toucher, dim3(256), dim3(1024), 0, 0, (const unsigned *)w, cb, 1, out); CK(hipDeviceSynchronize()); time_reader(cb, us); cross.push_back(us);
            evict(); CK(hipDeviceSynchronize());
            hipEventRecord(e0, 0); hipLaunchKernelGGL(toucher, dim3(256), dim3(1024), 0, 0, (const unsigned *)w, cb, 0, out); hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); tch.push_back(ms * 1e3f);
        }
        auto med = [](std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        printf("%5zu KB | %6.1f MB | %7.2f us | %7.2f us | %7.2f us | %7.2f us | %7.2f us\n", kb, cb * 256 / 1048576.0, med(cold), med(same), med(cross), med(rerun), med(tch));
    }
    // an empty launch for scale
    {
        std::vector<float> v;
        for (int rep = 0; rep < 7; rep++) { float us; time_reader(0, us); v.push_back(us); }
        std::sort(v.begin(), v.end());
        printf("empty reader launch: %.2f us\n", v[3]);
    }
    return 0;
}
