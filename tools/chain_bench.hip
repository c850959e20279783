// chain_bench.hip -- what does one link of a dependent-kernel chain cost on this GPU?
// Builds hipGraphs of N identical kernels (each depends on the previous) and times replays.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_empty() {}
__global__ void k_store(float *out) { if (threadIdx.x == 0) out[blockIdx.x] = 1.f; }
__global__ void k_load1(const float *in, float *out) { out[blockIdx.x * blockDim.x + threadIdx.x] = in[blockIdx.x * blockDim.x + threadIdx.x] + 1.f; }
__global__ void k_load2(const int *ctl, const float *in, float *out) {  // data address depends on a loaded word
    int off = ctl[0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = in[off + blockIdx.x * blockDim.x + threadIdx.x] + 1.f;
}
__global__ void k_lds(const float *in, float *out) {  // load -> LDS -> barrier -> reduce -> barrier -> store
    __shared__ float s[256];
    __shared__ float r[4];
    float v = in[blockIdx.x * blockDim.x + threadIdx.x];
    s[threadIdx.x] = v;
    __syncthreads();
    float a = s[(threadIdx.x * 7) & 255];
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if ((threadIdx.x & 63) == 0) r[threadIdx.x >> 6] = a;
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = r[0] + r[1] + r[2] + r[3];
}
__global__ void k_stream(const float4 *w, const float *in, float *out, int n4_per_thread) {  // stream weights
    float acc = in[threadIdx.x];
    const float4 *p = w + (size_t)(blockIdx.x * blockDim.x + threadIdx.x);
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (int i = 0; i < n4_per_thread; i++) { float4 v = p[i * stride]; acc += v.x + v.y + v.z + v.w; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
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
    const int reps = 20;
    CK(hipEventRecord(a, st));
    for (int r = 0; r < reps; r++) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(b, st));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-44s %7.3f us per kernel\n", name, ms * 1e3 / (reps * nk));
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return 0;
}

int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    float *a, *b; int *ctl; float4 *w;
    size_t n = 1 << 22;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&ctl, 64)); CK(hipMalloc(&w, (size_t)64 << 20));
    CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4)); CK(hipMemset(ctl, 0, 64)); CK(hipMemset(w, 0, (size_t)64 << 20));
    const int NK = 200;
    for (int wgs : {1, 32, 256, 1024}) {
        char nm[128];
        snprintf(nm, sizeof nm, "empty, %d WG x 256", wgs);
        time_chain(nm, NK, st, [&](int) { hipLaunchKernelGGL(k_empty, dim3(wgs), dim3(256), 0, st); });
        snprintf(nm, sizeof nm, "store only, %d WG", wgs);
        time_chain(nm, NK, st, [&](int) { hipLaunchKernelGGL(k_store, dim3(wgs), dim3(256), 0, st, b); });
        snprintf(nm, sizeof nm, "load->store (ping-pong), %d WG", wgs);
        time_chain(nm, NK, st, [&](int i) { hipLaunchKernelGGL(k_load1, dim3(wgs), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b); });
        snprintf(nm, sizeof nm, "ctl->load->store, %d WG", wgs);
        time_chain(nm, NK, st, [&](int i) { hipLaunchKernelGGL(k_load2, dim3(wgs), dim3(256), 0, st, ctl, (i & 1) ? b : a, (i & 1) ? a : b); });
        snprintf(nm, sizeof nm, "load->LDS->2 barriers->store, %d WG", wgs);
        time_chain(nm, NK, st, [&](int i) { hipLaunchKernelGGL(k_lds, dim3(wgs), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b); });
    }
    // streaming: 256 / 1024 WGs reading 1, 4, 16 MB per launch from a 64 MB pool (rotating offset)
    for (int wgs : {256, 1024}) for (int mb : {1, 4, 16}) {
        char nm[128];
        int per = (int)(((size_t)mb << 20) / 16 / ((size_t)wgs * 256));
        snprintf(nm, sizeof nm, "stream %d MB, %d WG (%d x16B/thread)", mb, wgs, per);
        time_chain(nm, NK, st, [&](int i) {
            const float4 *wp = w + ((size_t)(i % (64 / mb)) * ((size_t)mb << 20) / 16);
            hipLaunchKernelGGL(k_stream, dim3(wgs), dim3(256), 0, st, wp, a, b, per);
        });
    }
    // does L2 content survive a kernel boundary?  same buffer every launch vs a different one each launch
    for (int mb : {1, 4}) {
        int wgs = 256;
        int per = (int)(((size_t)mb << 20) / 16 / ((size_t)wgs * 256));
        char nm[128];
        snprintf(nm, sizeof nm, "stream %d MB SAME buffer each launch", mb);
        time_chain(nm, NK, st, [&](int) { hipLaunchKernelGGL(k_stream, dim3(wgs), dim3(256), 0, st, w, a, b, per); });
        snprintf(nm, sizeof nm, "stream %d MB rotating over 64 MB", mb);
        time_chain(nm, NK, st, [&](int i) {
            const float4 *wp = w + ((size_t)(i % (64 / mb)) * ((size_t)mb << 20) / 16);
            hipLaunchKernelGGL(k_stream, dim3(wgs), dim3(256), 0, st, wp, a, b, per);
        });
    }
    return 0;
}
