// Developer microbenchmark (round 4): do two dependent launch chains on two HIP streams overlap on gfx950 / ROCm 7.2?
// Each chain = NK launches of a small kernel (WGS workgroups that spin ~T us); measured: one chain alone, two chains on two
// streams (eager launches from one host thread), and the same as two captured graphs launched back to back.
//   hipcc --offload-arch=gfx950 -O3 tools/stream_overlap_probe.hip -o /tmp/sop && /tmp/sop
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void spin_kernel(float *p, int ticks) {
    const long long t0 = wall_clock64();
    float v = p[blockIdx.x * 64 + (threadIdx.x & 63)];
    while (wall_clock64() - t0 < ticks) v = v * 1.0001f + 0.5f;
    if (v == 12345.f) p[0] = v;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    float *buf;
    CK(hipMalloc(&buf, 1 << 20));
    CK(hipMemset(buf, 0, 1 << 20));
    hipStream_t s[4];
    for (auto &x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    const int NK = 400;
    for (int wgs : {64, 144, 256, 512}) {
        for (int ticks : {300, 600}) {     // wall clock 100 MHz: 3 us, 6 us
            auto chain = [&](hipStream_t st) { for (int i = 0; i < NK; i++) hipLaunchKernelGGL(spin_kernel, dim3(wgs), dim3(256), 0, st, buf, ticks); };
            chain(s[0]); CK(hipStreamSynchronize(s[0]));
            double t0 = now(); chain(s[0]); CK(hipStreamSynchronize(s[0])); const double one = now() - t0;
            t0 = now(); chain(s[0]); chain(s[1]); CK(hipStreamSynchronize(s[0])); CK(hipStreamSynchronize(s[1])); const double two = now() - t0;
            // interleaved submission (launch i of both chains alternately)
            t0 = now();
            for (int i = 0; i < NK; i++) { hipLaunchKernelGGL(spin_kernel, dim3(wgs), dim3(256), 0, s[0], buf, ticks); hipLaunchKernelGGL(spin_kernel, dim3(wgs), dim3(256), 0, s[1], buf, ticks); }
            CK(hipStreamSynchronize(s[0])); CK(hipStreamSynchronize(s[1])); const double two_i = now() - t0;
            // graphs
            hipGraph_t g[2]; hipGraphExec_t ge[2];
            for (int k = 0; k < 2; k++) {
                CK(hipStreamBeginCapture(s[k], hipStreamCaptureModeThreadLocal));
                chain(s[k]);
                CK(hipStreamEndCapture(s[k], &g[k]));
                CK(hipGraphInstantiate(&ge[k], g[k], nullptr, nullptr, 0));
            }
            CK(hipGraphLaunch(ge[0], s[0])); CK(hipStreamSynchronize(s[0]));
            t0 = now(); CK(hipGraphLaunch(ge[0], s[0])); CK(hipStreamSynchronize(s[0])); const double g1 = now() - t0;
            t0 = now(); CK(hipGraphLaunch(ge[0], s[0])); CK(hipGraphLaunch(ge[1], s[1])); CK(hipStreamSynchronize(s[0])); CK(hipStreamSynchronize(s[1])); const double g2 = now() - t0;
            printf("%3d workgroups x %d us, %d launches per chain: eager 1 chain %.2f us/launch; 2 chains on 2 streams %.2f (sequential submit) / %.2f (interleaved submit) "
                   "us per launch PAIR; graphs: 1 chain %.2f, 2 chains %.2f\n", wgs, ticks / 100, NK, one / NK * 1e6, two / NK * 1e6, two_i / NK * 1e6, g1 / NK * 1e6, g2 / NK * 1e6);
            for (int k = 0; k < 2; k++) { (void)hipGraphExecDestroy(ge[k]); (void)hipGraphDestroy(g[k]); }
        }
    }
    return 0;
}
