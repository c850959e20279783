// Developer probe: launch-to-launch time of back-to-back dependent launches of a kernel that does nothing, by grid, block size and static
// LDS -- the fixed cost under every launch of a decode-batch layer (nl_dgemm.h: 192-256 workgroups of 512-1024 threads, 150 KB of LDS).
// Build + run (gpurun): hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/launch_floor_probe.hip -o /tmp/lfp && /tmp/lfp
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <int LDS_U4>
__global__ void __launch_bounds__(1024) nop_kernel(float *out, int touch) {
    __shared__ uint4 lds[LDS_U4 > 0 ? LDS_U4 : 1];
    if (touch) { lds[threadIdx.x % (LDS_U4 > 0 ? LDS_U4 : 1)] = make_uint4(threadIdx.x, 0, 0, 0); __syncthreads(); }
    if (touch && threadIdx.x == 0 && lds[0].x == 12345u) out[blockIdx.x] = 1.f;
}
struct BigArgs { unsigned long long w[60]; };      // 480 bytes by value: the size of the engine's GEMM parameter blocks
__global__ void __launch_bounds__(1024) nop_big_kernel(float *out, int touch, BigArgs A) {
    if (touch == 12345 && threadIdx.x == 0) out[blockIdx.x] = (float)A.w[blockIdx.x % 60];
}
template <int LDS_U4>
int run(const char *name, dim3 grid, int threads, hipStream_t st, float *out) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(a, st));
        for (int i = 0; i < 400; i++) hipLaunchKernelGGL(nop_kernel<LDS_U4>, grid, dim3(threads), 0, st, out, 1);
        CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    }
    printf("%-34s grid %4u x %-2u  %4d threads  LDS %6d B: %6.2f us per launch\n", name, grid.x, grid.y, threads, LDS_U4 * 16, ms * 1e3 / 400);
    return 0;
}
int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    float *out; CK(hipMalloc(&out, 4096 * 4));
    run<0>("no LDS", dim3(32, 6), 512, st, out);
    run<0>("no LDS", dim3(32, 8), 512, st, out);
    run<0>("no LDS", dim3(32, 6), 1024, st, out);
    run<4096>("64 KB", dim3(32, 6), 512, st, out);
    run<9472>("148 KB (one workgroup per CU)", dim3(32, 6), 512, st, out);
    run<9472>("148 KB", dim3(32, 8), 512, st, out);
    run<9472>("148 KB", dim3(32, 6), 1024, st, out);
    run<9472>("148 KB", dim3(32, 12), 512, st, out);
    run<0>("no LDS, 384 x 256 (attention)", dim3(6, 1, 64), 256, st, out);
    {
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        BigArgs A{};
        float ms = 0;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(a, st));
            for (int i = 0; i < 400; i++) hipLaunchKernelGGL(nop_big_kernel, dim3(32, 6), dim3(512), 0, st, out, 1, A);
            CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
        }
        printf("%-34s grid   32 x 6    512 threads  480 bytes of kernel arguments: %6.2f us per launch\n", "no LDS", ms * 1e3 / 400);
        // the same 400 launches as one captured graph, replayed
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 400; i++) hipLaunchKernelGGL(nop_big_kernel, dim3(32, 6), dim3(512), 0, st, out, 1, A);
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(a, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
        }
        printf("%-34s the same 400 launches as a replayed graph:                    %6.2f us per launch\n", "no LDS", ms * 1e3 / 400);
        hipGraph_t g2; hipGraphExec_t ge2;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 400; i++) hipLaunchKernelGGL(nop_kernel<0>, dim3(32, 6), dim3(512), 0, st, out, 1);
        CK(hipStreamEndCapture(st, &g2)); CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(a, st)); CK(hipGraphLaunch(ge2, st)); CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
        }
        printf("%-34s ... with 12 bytes of kernel arguments, replayed graph:         %6.2f us per launch\n", "no LDS", ms * 1e3 / 400);
        hipGraph_t g3; hipGraphExec_t ge3;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 400; i++) hipLaunchKernelGGL(nop_kernel<9472>, dim3(32, 6), dim3(512), 0, st, out, 1);
        CK(hipStreamEndCapture(st, &g3)); CK(hipGraphInstantiate(&ge3, g3, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(a, st)); CK(hipGraphLaunch(ge3, st)); CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
        }
        printf("%-34s ... 12 bytes, 148 KB of LDS, replayed graph:                   %6.2f us per launch\n", "148 KB", ms * 1e3 / 400);
    }
    return 0;
}
