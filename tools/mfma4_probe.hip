// Developer probe (round 5): operand / result lane layout of v_mfma_i32_4x4x4_16b_i8 on gfx950, as nl_persist.h uses it -- sixteen
// independent 4x4x4 products per wavefront: lane l = (block l / 4, index l % 4) supplies A row i = l % 4 (four int8 k-values in one
// dword) and B column j = l % 4; the claim checked here: lane l's result register r holds D[r][l % 4] = sum_k A[r][k] * B[k][l % 4]
// of its block, i.e. (limb r of x) . (this lane's own weight row).
// Build + run (gpurun): hipcc --offload-arch=gfx950 -O2 tools/mfma4_probe.hip -o /tmp/m4p && /tmp/m4p
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void k(const int *a, const int *b, int *o) {
    v4i c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_i32_4x4x4i8(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
    for (int r = 0; r < 4; r++) o[threadIdx.x * 4 + r] = c[r];
}
int main() {
    signed char A[64][4], B[64][4];
    for (int l = 0; l < 64; l++)
        for (int kk = 0; kk < 4; kk++) { A[l][kk] = (signed char)((l * 7 + kk * 3) % 23 - 11); B[l][kk] = (signed char)((l * 5 + kk * 11) % 29 - 14); }
    int *da, *db, *dout, out[256];
    hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dout, 1024);
    hipMemcpy(da, A, 256, hipMemcpyHostToDevice); hipMemcpy(db, B, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dout);
    hipMemcpy(out, dout, 1024, hipMemcpyDeviceToHost);
    int bad_claim = 0, bad_transposed = 0;
    for (int l = 0; l < 64; l++)
        for (int r = 0; r < 4; r++) {
            const int blk = l / 4;
            int claim = 0, transposed = 0;
            for (int kk = 0; kk < 4; kk++) {
                claim += (int)A[blk * 4 + r][kk] * (int)B[l][kk];            // D[r][l % 4]: A row from lane (blk, r), B column = this lane
                transposed += (int)A[l][kk] * (int)B[blk * 4 + r][kk];       // D[l % 4][r]
            }
            bad_claim += out[l * 4 + r] != claim;
            bad_transposed += out[l * 4 + r] != transposed;
        }
    printf("v_mfma_i32_4x4x4_16b_i8: lane l reg r = D[r][l%%4] (A row r of the block, B = own lane): %s (%d mismatches); transposed reading: %d mismatches\n",
           bad_claim ? "NO" : "yes", bad_claim, bad_transposed);
    return 0;
}
