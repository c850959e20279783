// overlap_bench.hip -- can a decode step's dependent launches overlap?
//
// A token is a chain of ~67 small launches, each needing a short vector from its predecessor; a dependent launch
// costs ~1.5 us at the boundary plus one cold weight round trip inside.  Here the same chain is captured with its
// launches ALTERNATING between 2 (or 3) streams, so launch k+1 starts -- and issues its weight loads, which do not
// depend on the predecessor -- while launch k still runs; the vector travels as 8-byte {tag, value} granules
// (sc1 stores, sc1 polling loads: cdna_hip_programming.md G16 form R2), tag = per-step counter * 256 + stage.
// Prints us per stage for: one stream (plain boundary), 2 streams, 3 streams.
//   hipcc --offload-arch=gfx950 -O3 -o tools/overlap_bench.bin tools/overlap_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned long long u64;

__global__ void k_tick(unsigned *epoch) { if (threadIdx.x == 0) *epoch = *epoch + 1; }

// one stage: stream `per` x 16 B of weights per thread, wait for the predecessor's n_in granules, publish 16 per WG
__global__ void __launch_bounds__(256) k_stage(const float4 *w, int per, const u64 *in, u64 *out, int n_in, unsigned stage,
                                               const unsigned *epoch, unsigned *fail, int first) {
    const float4 *p = w + (size_t)(blockIdx.x * blockDim.x + threadIdx.x);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float4 v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = p[(size_t)min(i, per - 1) * stride];
    const unsigned ep = __hip_atomic_load(epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned want = (ep << 8) | ((stage - 1) & 255u);
    float xs = 0.f;
    const bool dead = __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    if (!first) {
        // every thread owns granules t, t+256, ...; the wave retries until all of its lanes see the tag
        for (int base = threadIdx.x; base < n_in; base += 256) {
            u64 g;
            int spins = 0;
            for (;;) {
                g = __hip_atomic_load(in + base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all((unsigned)(g >> 32) == want)) break;
                if (dead || ++spins > 200000) { if ((threadIdx.x & 63) == 0) atomicOr(fail, 1u); break; }
                __builtin_amdgcn_s_sleep(1);
            }
            xs += __uint_as_float((unsigned)g);
        }
    }
    float acc = xs;
#pragma unroll
    for (int i = 0; i < 8; i++) acc += v[i].x + v[i].y + v[i].z + v[i].w;
    for (int i = 8; i < per; i++) { float4 q = p[(size_t)i * stride]; acc += q.x + q.y + q.z + q.w; }
    // reduce over the workgroup (LDS + barrier), then 16 granules per workgroup
    __shared__ float red[4];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x < 16) {
        const float r = (red[0] + red[1] + red[2] + red[3]) * 1e-30f + 1.0f;
        const unsigned tag = (ep << 8) | (stage & 255u);
        __hip_atomic_store(out + blockIdx.x * 16 + threadIdx.x, ((u64)tag << 32) | __float_as_uint(r), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
}

struct Bufs { float4 *w; u64 *g[4]; unsigned *epoch, *fail; };

int run(const char *name, int nstreams, int wgs, int mb_x16, int n_stage, int steps, Bufs &B, hipStream_t *st) {
    // per = float4 per thread so that one launch streams mb_x16/16 MB
    const int per = (int)std::max<size_t>(1, ((size_t)mb_x16 << 16) / 16 / ((size_t)wgs * 256));
    const size_t bytes = (size_t)per * wgs * 256 * 16;
    const int slots = (int)std::max<size_t>(1, ((size_t)64 << 20) / bytes);
    hipGraph_t g; hipGraphExec_t ge;
    hipEvent_t ev[8];
    for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    CK(hipStreamBeginCapture(st[0], hipStreamCaptureModeThreadLocal));
    int li = 0;
    for (int s = 0; s < steps; s++) {
        // join: every stream's tail before the tick (the step's first launch), fork after it
        for (int q = 1; q < nstreams; q++) {
            if (s > 0) { CK(hipEventRecord(ev[q], st[q])); CK(hipStreamWaitEvent(st[0], ev[q], 0)); }
        }
        hipLaunchKernelGGL(k_tick, dim3(1), dim3(64), 0, st[0], B.epoch);
        CK(hipEventRecord(ev[0], st[0]));
        for (int q = 1; q < nstreams; q++) CK(hipStreamWaitEvent(st[q], ev[0], 0));
        for (int k = 0; k < n_stage; k++, li++) {
            const float4 *wp = B.w + (size_t)(li % slots) * (bytes / 16);
            // the stage reads the granules its predecessor wrote; 4 rotating granule buffers
            hipLaunchKernelGGL(k_stage, dim3(wgs), dim3(256), 0, st[k % nstreams], wp, per, B.g[(k + 3) & 3], B.g[k & 3],
                               wgs * 16, (unsigned)(k + 1), B.epoch, B.fail, k == 0 ? 1 : 0);
        }
    }
    for (int q = 1; q < nstreams; q++) { CK(hipEventRecord(ev[q], st[q])); CK(hipStreamWaitEvent(st[0], ev[q], 0)); }
    CK(hipStreamEndCapture(st[0], &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int w = 0; w < 3; w++) CK(hipGraphLaunch(ge, st[0]));
    CK(hipStreamSynchronize(st[0]));
    const int reps = 20;
    CK(hipEventRecord(a, st[0]));
    for (int r = 0; r < reps; r++) CK(hipGraphLaunch(ge, st[0]));
    CK(hipEventRecord(b, st[0]));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    unsigned fail = 0;
    CK(hipMemcpy(&fail, B.fail, 4, hipMemcpyDeviceToHost));
    printf("%-34s %d stream(s): %7.3f us per stage%s\n", name, nstreams, ms * 1e3 / (reps * steps * n_stage), fail ? "  [POLL TIMEOUT]" : "");
    CK(hipMemset(B.fail, 0, 4));
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    for (auto &e : ev) hipEventDestroy(e);
    return 0;
}

int main() {
    hipStream_t st[3];
    for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    Bufs B;
    CK(hipMalloc(&B.w, (size_t)64 << 20)); CK(hipMemset(B.w, 0, (size_t)64 << 20));
    for (auto &g : B.g) { CK(hipMalloc(&g, 4096 * 16 * 8)); CK(hipMemset(g, 0, 4096 * 16 * 8)); }
    CK(hipMalloc(&B.epoch, 8)); CK(hipMemset(B.epoch, 0, 8));
    B.fail = B.epoch + 1;
    struct Cfg { const char *name; int wgs, mb16; } cfgs[] = {
        {"36 WG, 0.35 MB (nano wo)", 36, 6}, {"96 WG, 1 MB (nano down)", 96, 16}, {"192 WG, 1.9 MB (nano gate/up)", 192, 30},
        {"256 WG, 9.5 MB (big wo)", 256, 152}, {"688 WG, 25 MB (big down)", 688, 400}};   // (a launch pair must be co-resident: 2 x 1376 WG x 4 waves would not be)
    for (auto &c : cfgs)
        for (int ns : {1, 2, 3})
            if (run(c.name, ns, c.wgs, c.mb16, 65, 4, B, st)) return 1;
    return 0;
}
