// Developer microbenchmark (round 4): what does an in-launch ALL-GATHER of a few thousand floats cost on gfx950, by transport?
// P producer workgroups each publish 16 values after a delay (they emulate the gate / up tiles of tp_ffn_kernel, nl_tp.h); C
// consumer workgroups (1024 threads, one per CU) gather all N = 16 P values into LDS.  Reported: time from the LAST publish to
// the LAST consumer done (and the mean), wall clock, over REPS launches; every gathered word is checked.
//   mode 0: 8-byte {tag, value} granules, consumers sweep-poll everything until every tag matches
//   mode 1: 8-byte granules, consumers first spin on ONE probe granule per producer, then sweep + validate once
//   mode 2: float payload (sc1 write-through stores) + drain + one tagged flag per producer; consumers spin on the flags,
//           then read the payload with 16-byte sc1 loads (no validation possible)
//   mode 3: as 2, but the payload is read with PLAIN 16-byte loads behind one agent-scope acquire fence
//   mode 4: 16-byte {tag, v0, v1, v2} granules (one dwordx4 sc1 store each), probe + sweep + validate
// Build + run (gpurun):  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/allgather_probe.hip -o /tmp/agp && /tmp/agp
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned long long u64;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Args {
    u64 *gran;          // [N] 8-byte granules (modes 0, 1)
    float *pay;         // [N] float payload (modes 2, 3)
    u64 *flag;          // [P] tagged flags (modes 2, 3)
    uint4 *gran16;      // [ceil(N / 3)] 16-byte granules (mode 4)
    long long *t_pub;   // [P]
    long long *t_done;  // [grid]
    int *bad;           // mismatching words seen by consumers
    int P, C, N, delay; // delay: s_sleep iterations before a producer publishes
    unsigned tag;
};

__device__ __forceinline__ float val_of(int i, unsigned tag) { return (float)((i * 7 + (int)tag * 13) & 0xffff) * 0.25f; }

template <int MODE>
__global__ void __launch_bounds__(1024) ag_kernel(Args A) {
    extern __shared__ float lds[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const bool prod = b < A.P, cons = b >= (int)gridDim.x - A.C;
    if (prod) {
        if (tid < 64) {
            for (int i = 0; i < A.delay; i++) __builtin_amdgcn_s_sleep(8);
            if (MODE <= 1) {
                if (tid < 16) __hip_atomic_store(A.gran + b * 16 + tid, ((u64)A.tag << 32) | __float_as_uint(val_of(b * 16 + tid, A.tag)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (MODE == 2 || MODE == 3) {
                if (tid < 16) __hip_atomic_store(A.pay + b * 16 + tid, val_of(b * 16 + tid, A.tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (tid == 0) __hip_atomic_store(A.flag + b, ((u64)A.tag << 32) | 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                // 16 values = 6 granules of 3 values (the last holds one); granule index g covers values 3g .. 3g+2
                const int g0 = (b * 16) / 3, g1 = (b * 16 + 15) / 3;
                const int g = g0 + tid;
                // a granule that straddles two producers is written by BOTH with the same content (values are a function of i)
                if (g <= g1) {
                    u32x4 v;
                    v.x = A.tag;
                    v.y = __float_as_uint(val_of(3 * g, A.tag)); v.z = __float_as_uint(val_of(3 * g + 1, A.tag)); v.w = __float_as_uint(val_of(3 * g + 2, A.tag));
                    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(A.gran16 + g), "v"(v) : "memory");
                }
            }
            if (tid == 0) A.t_pub[b] = wall_clock64();
        }
    }
    if (!cons) return;
    const int N = A.N;
    int spins = 0;
    if (MODE == 0) {
        for (;; spins++) {
            bool ok = true;
            for (int i = tid; i < N; i += 1024) {
                const u64 g = __hip_atomic_load(A.gran + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok &= (unsigned)(g >> 32) == A.tag;
                lds[i] = __uint_as_float((unsigned)g);
            }
            if (__syncthreads_and(ok) || spins > 200000) break;
            __builtin_amdgcn_s_sleep(1);
        }
    } else {
        // probes: one word per producer
        if ((tid & ~63) < A.P) {
            const int p = min(tid, A.P - 1);
            for (;; spins++) {
                bool ok;
                if (MODE == 1) ok = (unsigned)(__hip_atomic_load(A.gran + p * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) == A.tag;
                else if (MODE == 4) ok = __hip_atomic_load(reinterpret_cast<unsigned *>(A.gran16 + (p * 16) / 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == A.tag;
                else ok = (unsigned)(__hip_atomic_load(A.flag + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) == A.tag;
                if (__all(ok) || spins > 200000) break;
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
        if (MODE == 1) {
            for (;; spins++) {
                bool ok = true;
                u64 g[3];
#pragma unroll
                for (int q = 0; q < 3; q++) g[q] = __hip_atomic_load(A.gran + min(tid + q * 1024, N - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int q = 0; q < 3; q++) ok &= (unsigned)(g[q] >> 32) == A.tag;
#pragma unroll
                for (int q = 0; q < 3; q++) if (tid + q * 1024 < N) lds[tid + q * 1024] = __uint_as_float((unsigned)g[q]);
                if (__all(ok) || spins > 200000) break;
            }
        } else if (MODE == 2) {
            if (tid * 4 < N) {
                u32x4 v;
                asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(A.pay + tid * 4) : "memory");
                *reinterpret_cast<u32x4 *>(lds + tid * 4) = v;
            }
        } else if (MODE == 3) {
            if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __syncthreads();
            if (tid * 4 < N) *reinterpret_cast<float4 *>(lds + tid * 4) = *reinterpret_cast<const float4 *>(A.pay + tid * 4);
        } else {
            const int ng = (N + 2) / 3;
            for (;; spins++) {
                bool ok = true;
                u32x4 v = {A.tag, 0u, 0u, 0u};
                if (tid < ng) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(A.gran16 + tid) : "memory");
                ok = v.x == A.tag;
                if (tid < ng) { lds[3 * tid] = __uint_as_float(v.y); lds[3 * tid + 1] = __uint_as_float(v.z); lds[3 * tid + 2] = __uint_as_float(v.w); }
                if (__all(ok) || spins > 200000) break;
            }
        }
    }
    __syncthreads();
    if (tid == 0) A.t_done[b] = wall_clock64();
    int bad = 0;
    for (int i = tid; i < N; i += 1024) bad += lds[i] != val_of(i, A.tag);
    if (bad) atomicAdd(A.bad, bad);
}

template <int MODE>
void run(const char *name, Args A, int grid, int reps, bool pre_read) {
    std::vector<double> last, mean;
    int bad_total = 0;
    for (int rep = 0; rep < reps; rep++) {
        A.tag = 100 + rep * 7 + MODE * 1000;
        CK(hipMemset(A.bad, 0, 4));
        CK(hipMemset(A.t_done, 0, grid * 8));
        hipLaunchKernelGGL(ag_kernel<MODE>, dim3(grid), dim3(1024), (A.N + 8) * 4, 0, A);
        CK(hipDeviceSynchronize());
        std::vector<long long> tp(A.P), td(grid);
        int bad = 0;
        CK(hipMemcpy(tp.data(), A.t_pub, A.P * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(td.data(), A.t_done, grid * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&bad, A.bad, 4, hipMemcpyDeviceToHost));
        bad_total += bad;
        const long long pub = *std::max_element(tp.begin(), tp.end());
        long long mx = 0;
        double sum = 0;
        int n = 0;
        for (int b = grid - A.C; b < grid; b++) { mx = std::max(mx, td[b]); sum += (double)(td[b] - pub); n++; }
        if (rep >= 2) { last.push_back((mx - pub) / 100.0); mean.push_back(sum / n / 100.0); }
    }
    std::sort(last.begin(), last.end());
    std::sort(mean.begin(), mean.end());
    printf("  %-46s last consumer %5.2f us (min %5.2f), mean consumer %5.2f us, bad words %d\n", name, last[last.size() / 2], last[0],
           mean[mean.size() / 2], bad_total);
}

int main(int argc, char **argv) {
    const int reps = 12;
    struct Cfg { int P, C, N; } cfgs[] = {{172, 256, 2752}, {172, 128, 2752}, {172, 64, 2752}, {86, 256, 1376}, {86, 64, 1376}, {32, 192, 512}, {32, 48, 512}};
    for (const Cfg &c : cfgs) {
        Args A{};
        A.P = c.P; A.C = c.C; A.N = c.N; A.delay = 40;
        const int grid = std::max(c.P, c.C);    // consumers = the LAST C blocks; producers = the first P (roles overlap as in tp_ffn_kernel)
        CK(hipMalloc(&A.gran, c.N * 8)); CK(hipMalloc(&A.pay, c.N * 4 + 64)); CK(hipMalloc(&A.flag, c.P * 8));
        CK(hipMalloc(&A.gran16, (c.N / 3 + 2) * 16)); CK(hipMalloc(&A.t_pub, c.P * 8)); CK(hipMalloc(&A.t_done, grid * 8)); CK(hipMalloc(&A.bad, 4));
        CK(hipMemset(A.gran, 0, c.N * 8)); CK(hipMemset(A.pay, 0, c.N * 4 + 64)); CK(hipMemset(A.flag, 0, c.P * 8)); CK(hipMemset(A.gran16, 0, (c.N / 3 + 2) * 16));
        printf("P = %d producers x 16 values, C = %d consumers (grid %d x 1024 threads), N = %d values\n", c.P, c.C, grid, c.N);
        run<0>("0: 8-B granules, sweep-poll", A, grid, reps, false);
        run<1>("1: 8-B granules, probe then sweep", A, grid, reps, false);
        run<2>("2: f32 payload sc1 + drain + flag, 16-B sc1 loads", A, grid, reps, false);
        run<3>("3: f32 payload sc1 + drain + flag, acquire + plain", A, grid, reps, false);
        run<4>("4: 16-B granules {tag,3 values}, probe then sweep", A, grid, reps, false);
        (void)hipFree(A.gran); (void)hipFree(A.pay); (void)hipFree(A.flag); (void)hipFree(A.gran16); (void)hipFree(A.t_pub); (void)hipFree(A.t_done); (void)hipFree(A.bad);
    }
    return 0;
}
