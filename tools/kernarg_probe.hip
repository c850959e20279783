// kernarg_probe.hip -- does the kernel-argument fetch show in a dependent-launch chain, and does gfx950's kernarg preload
// (-mllvm -amdgpu-kernarg-preload-count=N: the first N dwords of explicit scalar arguments arrive in SGPRs with the
// wavefront) remove it?  Chains of 200 dependent load -> store kernels in a hipGraph, arguments as two scalars or inside a
// 256-byte struct passed by value (what the engine's launches do).
//   hipcc --offload-arch=gfx950 -O3 tools/kernarg_probe.hip -o /tmp/kp0
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=16 tools/kernarg_probe.hip -o /tmp/kp16
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
struct Big { int pad0[40]; const float *in; float *out; int pad1[20]; };
__global__ void k_scalar(const float *in, float *out) { out[blockIdx.x * blockDim.x + threadIdx.x] = in[blockIdx.x * blockDim.x + threadIdx.x] + 1.f; }
__global__ void k_struct(Big P) { P.out[blockIdx.x * blockDim.x + threadIdx.x] = P.in[blockIdx.x * blockDim.x + threadIdx.x] + 1.f; }
__global__ void k_both(const float *in, float *out, Big P) {   // the two pointers lead the argument list, the struct follows
    out[blockIdx.x * blockDim.x + threadIdx.x] = in[blockIdx.x * blockDim.x + threadIdx.x] + (float)P.pad1[3];
}
template <typename F>
int time_chain(const char *name, int nk, hipStream_t st, F launch) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < nk; i++) launch(i);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int w = 0; w < 3; w++) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    const int reps = 30;
    CK(hipEventRecord(a, st));
    for (int r = 0; r < reps; r++) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(b, st));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("  %-52s %7.3f us per kernel\n", name, ms * 1e3 / (reps * nk));
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return 0;
}
int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    float *a, *b; size_t n = 1 << 20;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4));
    for (int wgs : {32, 256}) {
        char nm[128];
        snprintf(nm, sizeof nm, "two scalar pointer arguments, %d WG", wgs);
        time_chain(nm, 200, st, [&](int i) { hipLaunchKernelGGL(k_scalar, dim3(wgs), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b); });
        snprintf(nm, sizeof nm, "pointers inside a 256-byte struct by value, %d WG", wgs);
        time_chain(nm, 200, st, [&](int i) { Big P{}; P.in = (i & 1) ? b : a; P.out = (i & 1) ? a : b; hipLaunchKernelGGL(k_struct, dim3(wgs), dim3(256), 0, st, P); });
        snprintf(nm, sizeof nm, "two leading pointers + the struct, %d WG", wgs);
        time_chain(nm, 200, st, [&](int i) { Big P{}; hipLaunchKernelGGL(k_both, dim3(wgs), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, P); });
    }
    return 0;
}
