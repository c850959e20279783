// clock_probe2.hip -- what clock does a latency-bound decode-like kernel actually run at?
// (a) independent v_fma_f32 issue (2 SIMD cycles each for a wave64 on a SIMD-32 pipe) against wall time gives the REAL shader
//     clock; s_memtime (clock64) against the 100 MHz wall clock gives the tick rate of clock64;
// (b) the same dependent chain alone, and beside a background kernel that keeps other compute units busy: does load raise the clock?
// Build: hipcc --offload-arch=gfx950 -O3 tools/clock_probe2.hip -o /tmp/clock_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void indep(float *out, long long *t, int iters) {
    float a0 = out[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(0.999f), "v"(0.001f));
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
}
__global__ void dep(float *out, long long *t, int iters) {
    float a = out[threadIdx.x];
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 64; k++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(0.999f), "v"(0.001f));
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
}
__global__ void burn(float *out, int iters) {     // background load: dense VALU on every lane
    float a0 = out[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) { a0 = fmaf(a0, 0.999f, 0.001f); a1 = fmaf(a1, 0.999f, 0.001f); a2 = fmaf(a2, 0.999f, 0.001f); a3 = fmaf(a3, 0.999f, 0.001f); }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}

int main() {
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    float *a, *b; long long *t; CK(hipMalloc(&a, 1 << 22)); CK(hipMalloc(&b, 1 << 24)); CK(hipMalloc(&t, 256)); CK(hipMemset(a, 0, 1 << 22)); CK(hipMemset(b, 0, 1 << 24));
    long long h[2];
    auto report = [&](const char *what, double ninstr, double cyc_per_instr_ideal) {
        hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
        const double us = h[1] / 100.0;
        printf("%-44s %9.1f us  clock64 rate %6.0f MHz  %6.2f clock64 ticks / instr  -> real clock if %.0f cyc/instr: %6.0f MHz\n", what, us, h[0] / us,
               h[0] / ninstr, cyc_per_instr_ideal, ninstr * cyc_per_instr_ideal / us);
    };
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(indep, dim3(1), dim3(64), 0, s1, a, t, 2000); CK(hipStreamSynchronize(s1));
        report("independent FMAs, 1 wave alone", 2000.0 * 64, 2.0);
        hipLaunchKernelGGL(indep, dim3(1024), dim3(256), 0, s1, a, t, 2000); CK(hipStreamSynchronize(s1));
        report("independent FMAs, 4096 waves (4 per SIMD)", 2000.0 * 64 * 4, 2.0);
        hipLaunchKernelGGL(dep, dim3(1), dim3(64), 0, s1, a, t, 2000); CK(hipStreamSynchronize(s1));
        report("dependent FMA chain, 1 wave alone", 2000.0 * 64, 4.0);
        hipLaunchKernelGGL(dep, dim3(64), dim3(64), 0, s1, a, t, 2000); CK(hipStreamSynchronize(s1));
        report("dependent FMA chain, 64 waves", 2000.0 * 64, 4.0);
        // the dependent chain beside a background load on the other compute units
        for (int bg : {64, 192, 1024}) {
            hipLaunchKernelGGL(burn, dim3(bg), dim3(256), 0, s2, b, 400000);
            hipLaunchKernelGGL(dep, dim3(1), dim3(64), 0, s1, a, t, 2000); CK(hipStreamSynchronize(s1));
            char lab[96]; snprintf(lab, sizeof lab, "dependent chain, 1 wave + %d burn workgroups", bg);
            report(lab, 2000.0 * 64, 4.0);
            CK(hipStreamSynchronize(s2));
        }
    }
    return 0;
}
