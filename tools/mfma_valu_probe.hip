// Developer probe (gfx950): can the vector work of a softmax-like phase hide in the gaps of an MFMA phase?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/mfma_valu_probe.hip -o /tmp/mfma_valu_probe
// One workgroup per CU, W wavefronts per SIMD.  Per iteration a wavefront owes 48 v_mfma_f32_16x16x32_f16 (eight
// independent accumulators, chains of six) and NV vector instructions on 32 independent registers.  Variants:
//   serial      : all MFMAs, then all vector instructions (what a phase-structured kernel does)
//   interleaved : after every MFMA, NV/48 vector instructions (sched_group_barrier pins the order)
// with the vector work as plain v_fma_f32, as v_pk_fma_f32 (half as many instructions for the same work) or as v_exp_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int MODE, int KIND, int PER>   // MODE 0 serial, 1 interleaved; KIND 0 fma, 1 pk_fma, 2 exp; PER = vector instructions per MFMA
__global__ void __launch_bounds__(1024) probe(float *out, long long *cyc, int iters) {
    half8_t a, b;
    for (int e = 0; e < 8; e++) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
    v4f acc[8];
    for (int i = 0; i < 8; i++) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
    v16f big[4];
    for (int i = 0; i < 4; i++) for (int e = 0; e < 16; e++) big[i][e] = 0.f;
    float r[32];
    for (int i = 0; i < 32; i++) r[i] = threadIdx.x * 0.01f + i;
    const float c = 1.0001f, d = 0.5f;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        auto vec = [&](int k) {
            if constexpr (KIND == 0) { r[k & 31] = fmaf(r[k & 31], c, d); }
            else if constexpr (KIND == 1) {
                v2f x = {r[(2 * k) & 31], r[(2 * k + 1) & 31]};
                x = x * (v2f){c, c} + (v2f){d, d};
                r[(2 * k) & 31] = x[0]; r[(2 * k + 1) & 31] = x[1];
            } else { r[k & 31] = __builtin_amdgcn_exp2f(r[k & 31]) * 0.5f; }
        };
        if constexpr (MODE >= 2) {   // the same flops as 24 v_mfma_f32_32x32x16_f16 on four accumulators
            if constexpr (MODE == 2) {
#pragma unroll
                for (int m = 0; m < 24; m++) big[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, big[m & 3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 24 * PER; k++) vec(k);
                __builtin_amdgcn_sched_barrier(0);
            } else {
#pragma unroll
                for (int m = 0; m < 24; m++) {
                    big[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, big[m & 3], 0, 0, 0);
#pragma unroll
                    for (int k = 0; k < PER; k++) vec(m * PER + k);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else if constexpr (MODE == 0) {
#pragma unroll
            for (int m = 0; m < 48; m++) acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[m & 7], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 48 * PER; k++) vec(k);
            __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
            for (int m = 0; m < 48; m++) {
                acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[m & 7], 0, 0, 0);
#pragma unroll
                for (int k = 0; k < PER; k++) vec(m * PER + k);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][3];
    for (int i = 0; i < 32; i++) s += r[i];
    for (int i = 0; i < 4; i++) s += big[i][0] + big[i][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    // (the oldest wavefront of a SIMD wins issue arbitration and would show its solo time: the span of the whole workgroup)
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { atomicMin((unsigned long long *)&cyc[1], (unsigned long long)t0); atomicMax((unsigned long long *)&cyc[0], (unsigned long long)t1); }
}

template <int MODE, int KIND, int PER>
static void run(const char *name, int waves_per_simd, float *out, long long *cyc) {
    const int iters = 200;
    hipLaunchKernelGGL((probe<MODE, KIND, PER>), dim3(256), dim3(256 * waves_per_simd), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    const long long init[2] = {0, 0x7fffffffffffffffLL};
    hipMemcpy(cyc, init, 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((probe<MODE, KIND, PER>), dim3(256), dim3(256 * waves_per_simd), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long hh[2] = {0, 0};
    hipMemcpy(hh, cyc, 16, hipMemcpyDeviceToHost);
    const long long h = hh[0] - hh[1];
    printf("  %-44s %d wave(s)/SIMD: %7.0f cycles per iteration (48 MFMA = %d MFMA-pipe cycles for the SIMD)\n", name, waves_per_simd,
           (double)h / iters, 48 * 16 * waves_per_simd);
}

int main() {
    float *out; long long *cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 16);
    for (int w = 1; w <= 4; w *= 2) {
        printf("%d wavefront(s) per SIMD\n", w);
        run<0, 0, 0>("MFMA only", w, out, cyc);
        run<0, 0, 4>("serial      + 192 v_fma_f32", w, out, cyc);
        run<1, 0, 4>("interleaved + 192 v_fma_f32 (4 per MFMA)", w, out, cyc);
        run<0, 0, 8>("serial      + 384 v_fma_f32", w, out, cyc);
        run<1, 0, 8>("interleaved + 384 v_fma_f32 (8 per MFMA)", w, out, cyc);
        run<0, 1, 2>("serial      + 96 v_pk_fma_f32", w, out, cyc);
        run<1, 1, 2>("interleaved + 96 v_pk_fma_f32 (2 per MFMA)", w, out, cyc);
        run<2, 0, 0>("24 x 32x32x16 MFMA only", w, out, cyc);
        run<2, 0, 8>("24 x 32x32x16 serial      + 192 v_fma_f32", w, out, cyc);
        run<3, 0, 8>("24 x 32x32x16 interleaved + 192 v_fma_f32 (8 per MFMA)", w, out, cyc);
        run<3, 0, 5>("24 x 32x32x16 interleaved + 120 v_fma_f32 (5 per MFMA)", w, out, cyc);
        run<2, 0, 16>("24 x 32x32x16 serial      + 384 v_fma_f32", w, out, cyc);
        run<3, 0, 16>("24 x 32x32x16 interleaved + 384 v_fma_f32 (16 per MFMA)", w, out, cyc);
        run<0, 2, 1>("serial      + 48 v_exp_f32 (+ mul)", w, out, cyc);
        run<1, 2, 1>("interleaved + 48 v_exp_f32 (+ mul, 1 per MFMA)", w, out, cyc);
    }
    return 0;
}
