// Discovers the operand layout of v_mfma_f64_16x16x4_f64 on gfx950 empirically.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void probe(double* outRow, double* outCol, double* outK) {
    int l = threadIdx.x;
    // hypothesis for inputs: A[i=l%16][k=l/16], B[k=l/16][j=l%16]
    // test 1: A[i][k] = (k==0)? i : 0 ; B[k][j] = (k==0)? 1 : 0  -> D[i][j] = i
    double a = (l / 16 == 0) ? (double)(l % 16) : 0.0, b = (l / 16 == 0) ? 1.0 : 0.0;
    d4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) outRow[l * 4 + r] = acc[r];
    // test 2: A[i][0] = 1 ; B[0][j] = j -> D[i][j] = j
    a = (l / 16 == 0) ? 1.0 : 0.0; b = (l / 16 == 0) ? (double)(l % 16) : 0.0;
    acc = d4{0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) outCol[l * 4 + r] = acc[r];
    // test 3: k pairing: A[i][k] = 10^k, B[k][j] = (k+1) -> D = sum_k 10^k (k+1) = 1+20+300+4000 = 4321 iff k's pair up
    double p10[4] = {1, 10, 100, 1000};
    a = p10[l / 16]; b = (double)(l / 16 + 1);
    acc = d4{0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) outK[l * 4 + r] = acc[r];
}
int main() {
    double *dr, *dc, *dk; hipMalloc(&dr, 256 * 8); hipMalloc(&dc, 256 * 8); hipMalloc(&dk, 256 * 8);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dr, dc, dk);
    double hr[256], hc[256], hk[256];
    hipMemcpy(hr, dr, sizeof hr, hipMemcpyDeviceToHost); hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost);
    hipMemcpy(hk, dk, sizeof hk, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 1) printf("lane %2d: rows %g %g %g %g  cols %g %g %g %g  k %g\n", l, hr[l*4], hr[l*4+1], hr[l*4+2], hr[l*4+3], hc[l*4], hc[l*4+1], hc[l*4+2], hc[l*4+3], hk[l*4]);
    return 0;
}
