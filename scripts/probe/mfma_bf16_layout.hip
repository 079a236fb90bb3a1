// Discovers the operand layout of v_mfma_f32_16x16x16_bf16 (the _1k form) on gfx950 empirically.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ short bf(float x) { return (short)(__float_as_uint(x) >> 16); }
__global__ void probe(float* outRow, float* outCol, float* outK) {
    int l = threadIdx.x;
    // hypothesis: A[i=l%16][k=4*(l/16)+t], B[k=4*(l/16)+t][j=l%16]
    s4 a, b; f4 acc;
    // test 1: A[i][0] = i, B[0][j] = 1 -> D[i][j] = i
    for (int t = 0; t < 4; ++t) { a[t] = (l / 16 == 0 && t == 0) ? bf((float)(l % 16)) : 0; b[t] = (l / 16 == 0 && t == 0) ? bf(1.f) : 0; }
    acc = f4{0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) outRow[l * 4 + r] = acc[r];
    // test 2: A[i][0] = 1, B[0][j] = j -> D = j
    for (int t = 0; t < 4; ++t) { a[t] = (l / 16 == 0 && t == 0) ? bf(1.f) : 0; b[t] = (l / 16 == 0 && t == 0) ? bf((float)(l % 16)) : 0; }
    acc = f4{0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) outCol[l * 4 + r] = acc[r];
    // test 3: k pairing: A[i][k] = 2^(k%8) (k = 4*(l/16)+t), B[k][j] = (k+1) only if same k index pairs up:
    // sum_k 2^(k%8)*(k+1) for k=0..15
    for (int t = 0; t < 4; ++t) { int k = 4 * (l / 16) + t; a[t] = bf((float)(1 << (k % 8))); b[t] = bf((float)(k + 1)); }
    acc = f4{0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, acc, 0, 0, 0);
    outK[l] = acc[0];
}
int main() {
    float *dr, *dc, *dk; hipMalloc(&dr, 256 * 4); hipMalloc(&dc, 256 * 4); hipMalloc(&dk, 64 * 4);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dr, dc, dk);
    float hr[256], hc[256], hk[64];
    hipMemcpy(hr, dr, sizeof hr, hipMemcpyDeviceToHost); hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost); hipMemcpy(hk, dk, sizeof hk, hipMemcpyDeviceToHost);
    float expect = 0; for (int k = 0; k < 16; ++k) expect += (float)(1 << (k % 8)) * (k + 1);
    for (int l = 0; l < 64; l += 5) printf("lane %2d: rows %g %g %g %g  cols %g %g %g %g  k %g (expect %g)\n", l, hr[l*4], hr[l*4+1], hr[l*4+2], hr[l*4+3], hc[l*4], hc[l*4+1], hc[l*4+2], hc[l*4+3], hk[l], expect);
    return 0;
}
