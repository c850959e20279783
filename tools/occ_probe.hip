// occ_probe.hip -- developer probe: resident workgroups per compute unit of the multi-token kernels (occupancy API), and a
// census (how many workgroups of a full grid run at once) for the prompt attention kernel's shape.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I nanollama_amd/csrc tools/occ_probe.hip -o /tmp/occ_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include "nl_batch.h"
#include "nl_qgemm2.h"
using namespace nl;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int LDSB, int THREADS>
__global__ void __launch_bounds__(THREADS) census(int *live, int *peak, int spin) {
    __shared__ char pad[LDSB];
    if (threadIdx.x == 0) {
        pad[0] = 1;
        const int now = atomicAdd(live, 1) + 1;
        atomicMax(peak, now);
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
        atomicSub(live, 1);
    }
    __syncthreads();
    if (pad[0] == 7) live[1] = 1;
}

int main() {
    int n = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn_tile16_kernel<64, 4, 32>, 512, 0)); printf("attn_tile16_kernel<64,4,32>  512 threads, 72192 B LDS: %d blocks per CU (API)\n", n);
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn_tile16_kernel<64, 1, 128>, 512, 0)); printf("attn_tile16_kernel<64,1,128>: %d blocks per CU (API)\n", n);
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, qgemm2_kernel<WT_Q4_0, 4, 2, 8, QG_EPI_SWIGLU>, 512, 0)); printf("qgemm2<4,2,8,swiglu>: %d blocks per CU (API)\n", n);
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, qgemm2_kernel<WT_Q4_0, 4, 1, 4, QG_EPI_PLAIN>, 256, 0)); printf("qgemm2<4,1,4,plain>: %d blocks per CU (API)\n", n);
    int *d; CK(hipMalloc(&d, 16));
    auto run = [&](const char *what, auto kernel, int threads) {
        hipMemset(d, 0, 16);
        hipLaunchKernelGGL(kernel, dim3(2048), dim3(threads), 0, 0, d, d + 2, 20000);   // 200 us of residence per workgroup
        hipDeviceSynchronize();
        int h[4]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("census %-40s peak resident workgroups %d (= %.2f per CU)\n", what, h[2], h[2] / 256.0);
    };
    run("512 threads, 72192 B LDS", census<72192, 512>, 512);
    run("512 threads, 65536 B LDS", census<65536, 512>, 512);
    run("512 threads, 40960 B LDS", census<40960, 512>, 512);
    run("256 threads, 72192 B LDS", census<72192, 256>, 256);
    return 0;
}
