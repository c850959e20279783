// Developer probe: how fast can ONE compute unit pull an L2 / Infinity-Cache resident array?
// The decode-batch GEMM (qgemm_kernel at 64 tokens) re-reads the step's 393 KB of activation fragments in every 64-row
// workgroup and its time follows those bytes at ~32 GB/s per compute unit; the guide quotes 62-122 GB/s per CU.
// Consumer: a workgroup of W wavefronts streams the 393 KB array with L x 16-byte loads outstanding per lane (all issued,
// then consumed), REP times; in-kernel wall clock (100 MHz) per workgroup and event time for the launch.
// Knobs swept: W in {4, 8, 16}, L in {4, 8, 16}, workgroups per CU in {1, 2, 4} (grid = 256 k: blocks i and i + 256 share a
// CU), producer on the consumer's XCD vs on another XCD (block b runs on XCD b % 8; checked with HW_REG_XCC_ID), first
// pass after the producer vs warm passes, plain vs non-temporal loads.
// Build + run (gpurun):  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ingest_probe.hip -o /tmp/ip && /tmp/ip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int NVEC = 393216 / 16;     // 24576 16-byte vectors

// xcd >= 0: only the blocks of that XCD write the array; -1: every block writes an interleaved share (the real producers:
// a launch spread over the chip); -2: the blocks of XCD j write COPY j (a + j * NVEC) -- one copy per XCD
__global__ void producer(uint4 *a, int xcd, unsigned seed) {
    if (xcd >= 0 && (int)(blockIdx.x & 7) != xcd) return;
    if (xcd == -1) {
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < NVEC; i += gridDim.x * blockDim.x) a[i] = make_uint4(i ^ seed, seed, i, 1);
        return;
    }
    const int nb = gridDim.x / 8, b = blockIdx.x / 8;
    uint4 *dst = xcd == -2 ? a + (size_t)(blockIdx.x & 7) * NVEC : a;
    for (int i = b * blockDim.x + threadIdx.x; i < NVEC; i += nb * blockDim.x) dst[i] = make_uint4(i ^ seed, seed, i, 1);
}

struct Out { unsigned long long t0, t1; unsigned xcc, sum; };

// load policy: 0 plain, 1 nt, 2 sc1 (agent scope: the vector L1 is not consulted / filled), 3 sc0, 4 sc0 sc1, 5 sc1 nt
template <int POL>
__device__ __forceinline__ uint4 load_pol(const uint4 *p) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 t;
    if (POL == 0) return *p;
    else if (POL == 1) t = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    else if (POL == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(t) : "v"(p) : "memory");
    else if (POL == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(t) : "v"(p) : "memory");
    else if (POL == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(t) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(t) : "v"(p) : "memory");
    return make_uint4(t.x, t.y, t.z, t.w);
}

template <int L, int POL>
__global__ void consumer(const uint4 *a, int xcd, int reps, Out *out) {
    if (xcd >= 0 && (int)(blockIdx.x & 7) != xcd) return;
    if (xcd == -2) a += (size_t)(blockIdx.x & 7) * NVEC;       // its own XCD's copy
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const int nthr = blockDim.x, tid = threadIdx.x;
    unsigned acc = 0;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    for (int r = 0; r < reps; r++) {
        for (int base = 0; base < NVEC; base += nthr * L) {
            uint4 v[L];
            if constexpr (POL >= 2) {
                // eight loads and their wait in ONE asm statement (the compiler does not count inline-asm loads)
                static_assert(POL < 2 || L == 8, "inline-asm policies: L = 8");
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                const uint4 *q[8];
#pragma unroll
                for (int j = 0; j < 8; j++) q[j] = a + min(base + j * nthr + tid, NVEC - 1);
                u32x4 t[8];
#define LD8(POLSTR) asm volatile( \
                    "global_load_dwordx4 %0, %8, off " POLSTR "\n\tglobal_load_dwordx4 %1, %9, off " POLSTR "\n\t" \
                    "global_load_dwordx4 %2, %10, off " POLSTR "\n\tglobal_load_dwordx4 %3, %11, off " POLSTR "\n\t" \
                    "global_load_dwordx4 %4, %12, off " POLSTR "\n\tglobal_load_dwordx4 %5, %13, off " POLSTR "\n\t" \
                    "global_load_dwordx4 %6, %14, off " POLSTR "\n\tglobal_load_dwordx4 %7, %15, off " POLSTR "\n\ts_waitcnt vmcnt(0)" \
                    : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7]) \
                    : "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(q[4]), "v"(q[5]), "v"(q[6]), "v"(q[7]) : "memory")
                if (POL == 2) LD8("sc1"); else if (POL == 3) LD8("sc0"); else if (POL == 4) LD8("sc0 sc1"); else LD8("sc1 nt");
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = make_uint4(t[j].x, t[j].y, t[j].z, t[j].w);
            } else {
#pragma unroll
                for (int j = 0; j < L; j++) {
                    const int i = min(base + j * nthr + tid, NVEC - 1);
                    v[j] = load_pol<POL>(a + i);
                }
            }
#pragma unroll
            for (int j = 0; j < L; j++) acc += v[j].x ^ v[j].w;
        }
    }
    __syncthreads();
    const unsigned long long t1 = wall_clock64();
    if (tid == 0) { out[blockIdx.x].t0 = t0; out[blockIdx.x].t1 = t1; out[blockIdx.x].xcc = xcc & 0xf; }
    if (acc == 0x12345678u) out[blockIdx.x].sum = acc;         // keep the loads
}

template <int L, int POL>
void run(const char *label, uint4 *a, Out *dout, int waves, int wg_per_cu, int prod_xcd, int cons_xcd, int reps, bool cold) {
    const int grid = 256 * wg_per_cu;
    std::vector<Out> h(grid);
    double best_us = 1e9, best_wg = 1e9, worst_wg = 0;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 5; it++) {
        if (cold) { hipLaunchKernelGGL(producer, dim3(64), dim3(256), 0, 0, a, prod_xcd, (unsigned)it); }
        CK(hipMemset(dout, 0, grid * sizeof(Out)));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((consumer<L, POL>), dim3(grid), dim3(waves * 64), 0, 0, a, cons_xcd, reps, dout);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h.data(), dout, grid * sizeof(Out), hipMemcpyDeviceToHost));
        double mx = 0, mn = 1e9; int n = 0;
        for (auto &o : h) if (o.t1) { const double us = (o.t1 - o.t0) * 0.01; mx = std::max(mx, us); mn = std::min(mn, us); n++; }
        if (it == 0 && cons_xcd >= 0) for (auto &o : h) if (o.t1 && (int)o.xcc != cons_xcd) { printf("  (block ran on XCC %u, expected %d)\n", o.xcc, cons_xcd); break; }
        if (mx < worst_wg || it == 0) worst_wg = mx;
        best_wg = std::min(best_wg, mn); best_us = std::min(best_us, (double)ms * 1e3);
        (void)n;
    }
    const double bytes = 393216.0 * reps;
    printf("%-58s W=%2d L=%2d wg/CU=%d: slowest workgroup %7.2f us = %6.1f GB/s per workgroup, %6.1f GB/s per CU (fastest %6.2f us; launch %7.2f us)\n",
           label, waves, L, wg_per_cu, worst_wg, bytes / worst_wg * 1e-3, bytes * wg_per_cu / worst_wg * 1e-3, best_wg, best_us);
}

int main() {
    uint4 *a; Out *dout;
    CK(hipMalloc(&a, 393216 * 8)); CK(hipMalloc(&dout, 1024 * sizeof(Out)));
    hipLaunchKernelGGL(producer, dim3(64), dim3(256), 0, 0, a, 0, 1u);
    CK(hipDeviceSynchronize());
    printf("== one pass right after a producer that ran on ONE XCD; consumers = the 32 x k workgroups of one XCD ==\n");
    for (int k : {1, 2}) {
        run<8, 0>("cold, producer XCD 0 -> consumers XCD 0 (same L2)", a, dout, 4, k, 0, 0, 1, true);
        run<8, 0>("cold, producer XCD 3 -> consumers XCD 0 (other XCD)", a, dout, 4, k, 3, 0, 1, true);
    }
    run<8, 1>("cold, producer XCD 0 -> consumers XCD 0, nt loads", a, dout, 4, 1, 0, 0, 1, true);
    run<8, 1>("cold, producer XCD 3 -> consumers XCD 0, nt loads", a, dout, 4, 1, 3, 0, 1, true);
    printf("== the real case: ONE pass right after a producer launch spread over the chip; 128 / 256 consumers on all XCDs ==\n");
    run<8, 0>("cold, producers everywhere, plain loads", a, dout, 4, 1, -1, -1, 1, true);
    run<8, 1>("cold, producers everywhere, nt loads", a, dout, 4, 1, -1, -1, 1, true);
    run<8, 0>("cold, one copy per XCD written on that XCD, plain loads", a, dout, 4, 1, -2, -2, 1, true);
    run<8, 1>("cold, one copy per XCD written on that XCD, nt loads", a, dout, 4, 1, -2, -2, 1, true);
    run<8, 2>("cold, producers everywhere, sc1 loads", a, dout, 4, 1, -1, -1, 1, true);
    run<8, 3>("cold, producers everywhere, sc0 loads", a, dout, 4, 1, -1, -1, 1, true);
    run<8, 4>("cold, producers everywhere, sc0 sc1 loads", a, dout, 4, 1, -1, -1, 1, true);
    run<8, 5>("cold, producers everywhere, sc1 nt loads", a, dout, 4, 1, -1, -1, 1, true);
    run<8, 2>("cold, one copy per XCD written on that XCD, sc1 loads", a, dout, 4, 1, -2, -2, 1, true);
    printf("== warm: 8 passes over the array, every CU of the chip, k workgroups per CU ==\n");
    run<8, 2>("warm, all XCDs, sc1 loads", a, dout, 4, 1, 0, -1, 8, false);
    run<8, 3>("warm, all XCDs, sc0 loads", a, dout, 4, 1, 0, -1, 8, false);
    run<8, 4>("warm, all XCDs, sc0 sc1 loads", a, dout, 4, 1, 0, -1, 8, false);
    run<8, 5>("warm, all XCDs, sc1 nt loads", a, dout, 4, 1, 0, -1, 8, false);
    for (int k : {1, 2, 4}) {
        run<4, 0>("warm, all XCDs", a, dout, 4, k, 0, -1, 8, false);
        run<8, 0>("warm, all XCDs", a, dout, 4, k, 0, -1, 8, false);
        run<16, 0>("warm, all XCDs", a, dout, 4, k, 0, -1, 8, false);
    }
    for (int w : {8, 16}) {
        run<4, 0>("warm, all XCDs", a, dout, w, 1, 0, -1, 8, false);
        run<8, 0>("warm, all XCDs", a, dout, w, 1, 0, -1, 8, false);
        run<16, 0>("warm, all XCDs", a, dout, w, 1, 0, -1, 8, false);
    }
    run<8, 1>("warm, all XCDs, non-temporal loads", a, dout, 4, 1, 0, -1, 8, false);
    run<8, 1>("warm, all XCDs, non-temporal loads", a, dout, 16, 1, 0, -1, 8, false);
    printf("== warm, ONE workgroup on the whole chip (no neighbours on its L2 channel) ==\n");
    // grid 256 with only block 0 active: emulate by cons_xcd filter on a 1-block grid is not possible here; use wg_per_cu=1 and xcd 0 (32 CUs)
    run<8, 0>("warm, the 32 CUs of XCD 0 only", a, dout, 4, 1, 0, 0, 8, false);
    run<16, 0>("warm, the 32 CUs of XCD 0 only", a, dout, 16, 1, 0, 0, 8, false);
    return 0;
}
