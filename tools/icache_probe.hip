// icache_probe.hip -- developer probe: does a wavefront that runs LONG STRAIGHT-LINE code once (the shape of every decode
// launch here: ~2000 instructions, each executed one to three times) pay for instruction fetch?  A dependent v_fma chain
// of N instructions, (a) fully unrolled (N x 8 bytes of code, each line fetched once) and (b) as a loop over a 64-instruction
// body (code stays in the instruction cache), timed with s_memtime inside the kernel (cycles per instruction), cold (another
// kernel's code has run in between) and warm (same kernel back to back).
// Build: hipcc --offload-arch=gfx950 -O3 tools/icache_probe.hip -o /tmp/icache_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int N>
__global__ void straight(float *out, long long *cyc, float a, float b) {
    float v = out[threadIdx.x];
    const long long t0 = clock64();
#pragma unroll
    for (int i = 0; i < N; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(a), "v"(b));
    const long long t1 = clock64();
    out[threadIdx.x] = v;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int N>
__global__ void looped(float *out, long long *cyc, float a, float b) {
    float v = out[threadIdx.x];
    const long long t0 = clock64();
#pragma unroll 1
    for (int k = 0; k < N / 64; k++) {
#pragma unroll
        for (int i = 0; i < 64; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(a), "v"(b));
    }
    const long long t1 = clock64();
    out[threadIdx.x] = v;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// something else with a large code footprint, to push the probe's lines out of the instruction cache
template <int N>
__global__ void evict(float *out, float a, float b) {
    float v = out[threadIdx.x], w = v + 1.f;
#pragma unroll
    for (int i = 0; i < N; i++) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(a), "v"(w)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(w) : "v"(b), "v"(v)); }
    out[threadIdx.x] = v + w;
}

template <typename F>
double run(F launch, long long *d_cyc, int grid, bool cold, float *d_out) {
    long long h[256];
    double sum = 0;
    const int reps = 20;
    for (int r = 0; r < reps + 2; r++) {
        if (cold) hipLaunchKernelGGL(evict<6000>, dim3(512), dim3(64), 0, 0, d_out, 0.5f, 0.25f);
        launch();
        hipMemcpy(h, d_cyc, grid * sizeof(long long), hipMemcpyDeviceToHost);
        if (r >= 2) { double m = 0; for (int i = 0; i < grid; i++) m += (double)h[i]; sum += m / grid; }
    }
    return sum / reps;
}

int main() {
    float *d_out; long long *d_cyc;
    CK(hipMalloc(&d_out, 4096)); CK(hipMemset(d_out, 0, 4096)); CK(hipMalloc(&d_cyc, 256 * 8));
    for (int grid : {1, 64, 256}) {
#define CASE(N) { \
        double sc = run([&] { hipLaunchKernelGGL(straight<N>, dim3(grid), dim3(64), 0, 0, d_out, d_cyc, 0.999f, 0.001f); }, d_cyc, grid, true, d_out); \
        double sw = run([&] { hipLaunchKernelGGL(straight<N>, dim3(grid), dim3(64), 0, 0, d_out, d_cyc, 0.999f, 0.001f); }, d_cyc, grid, false, d_out); \
        double lc = run([&] { hipLaunchKernelGGL(looped<N>, dim3(grid), dim3(64), 0, 0, d_out, d_cyc, 0.999f, 0.001f); }, d_cyc, grid, true, d_out); \
        double lw = run([&] { hipLaunchKernelGGL(looped<N>, dim3(grid), dim3(64), 0, 0, d_out, d_cyc, 0.999f, 0.001f); }, d_cyc, grid, false, d_out); \
        printf("grid %3d  N=%5d  cycles/instr: straight cold %.2f warm %.2f | looped cold %.2f warm %.2f\n", grid, N, sc / N, sw / N, lc / N, lw / N); }
        CASE(512) CASE(2048) CASE(8192)
    }
    return 0;
}
