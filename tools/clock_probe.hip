// clock_probe.hip -- shader clock under a decode-like (launch-latency-bound) load, and in-kernel phase timing.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_chain(float *out, long long *t, int n) {
    long long c0 = clock64(), w0 = wall_clock64();
    float a = out[threadIdx.x];
    for (int i = 0; i < n; i++) a = fmaf(a, 1.0001f, 0.5f);
    out[threadIdx.x] = a;
    long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
}
__global__ void k_f64(const float *in, float *out, long long *t) {
    long long c0 = clock64();
    double v = (double)in[threadIdx.x];
    long long c1 = clock64();
    float e = (float)exp(-v);
    long long c2 = clock64();
    float s = (float)(1.0 / sqrt(v * v / 576.0 + 1e-5));
    long long c3 = clock64();
    out[threadIdx.x] = e + s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = c1 - c0; t[1] = c2 - c1; t[2] = c3 - c2; }
}
int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    float *a; long long *t; CK(hipMalloc(&a, 1 << 20)); CK(hipMalloc(&t, 256)); CK(hipMemset(a, 0, 1 << 20));
    long long h[4];
    for (int rep = 0; rep < 3; rep++)
    for (int n : {1024, 16384}) for (int wgs : {1, 256}) {
        hipLaunchKernelGGL(k_chain, dim3(wgs), dim3(64), 0, st, a, t, n);
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(h, t, 16, hipMemcpyDeviceToHost));
        printf("chain n=%5d wgs=%3d: %lld shader cycles, %lld wall ticks(100MHz) -> %.0f MHz, %.2f cyc/fma\n", n, wgs, h[0], h[1],
               h[0] / (h[1] / 100.0), (double)h[0] / n);
    }
    // now under a graph of tiny dependent kernels (decode-like): is the clock lower?
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 200; i++) hipLaunchKernelGGL(k_chain, dim3(32), dim3(64), 0, st, a, t, 2048);
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < 50; r++) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(h, t, 16, hipMemcpyDeviceToHost));
    printf("inside graph chain n=2048: %lld cycles %lld ticks -> %.0f MHz\n", h[0], h[1], h[0] / (h[1] / 100.0));
    hipLaunchKernelGGL(k_f64, dim3(1), dim3(64), 0, st, a, a + 1024, t);
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(h, t, 24, hipMemcpyDeviceToHost));
    printf("f64: load %lld cyc, exp(double) %lld cyc, 1/sqrt(double) %lld cyc\n", h[0], h[1], h[2]);
    return 0;
}
