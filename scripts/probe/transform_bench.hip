// Ablation bench for the FDM transform kernel (M=3456 complex rows, NYP=208).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct alignas(16) cplx { double re, im; };
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NTW, int MODE>   // MODE 0 full, 1 B always k-group 0 (L1 resident), 2 no loads in loop, 3 A only, 4 B only
__device__ __forceinline__ void body(const cplx* __restrict__ A, const double* __restrict__ Bsw, cplx* __restrict__ C,
                                     int M, int NYP, int m0, int t0, int lane) {
    const int NT = NYP >> 4, KG = NYP >> 4;
    const int li = lane & 15, lk = lane >> 4;
    const bool im = (li >> 3) != 0;
    const int arow = min(m0 + (li & 7), M - 1);
    const d4* Ap = reinterpret_cast<const d4*>(A + (long)arow * NYP + 4 * lk);
    const d4* Bp = reinterpret_cast<const d4*>(Bsw) + (long)t0 * 64 + lane;
    const long bstride = (long)NT * 64;
    d4 acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[t] = d4{0, 0, 0, 0};
    d4 a0 = Ap[0], a1 = Ap[1];
    d4 b[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) b[t] = Bp[t * 64];
    for (int kg = 0; kg < KG; ++kg) {
        const int kn = (kg + 1 < KG) ? kg + 1 : kg;
        d4 na0 = a0, na1 = a1;
        d4 nb[NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t) nb[t] = b[t];
        if (MODE == 0 || MODE == 1 || MODE == 3) { na0 = Ap[kn * 8]; na1 = Ap[kn * 8 + 1]; }
        if (MODE == 0 || MODE == 4) {
#pragma unroll
            for (int t = 0; t < NTW; ++t) nb[t] = Bp[kn * bstride + t * 64];
        }
        if (MODE == 1) {
#pragma unroll
            for (int t = 0; t < NTW; ++t) nb[t] = Bp[(kn & 1) * bstride + t * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
        const double av[4] = {im ? a0[1] : a0[0], im ? a0[3] : a0[2], im ? a1[1] : a1[0], im ? a1[3] : a1[2]};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int t = 0; t < NTW; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], b[t][i], acc[t], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        a0 = na0; a1 = na1;
        asm volatile("" : "+v"(a0), "+v"(a1));
#pragma unroll
        for (int t = 0; t < NTW; ++t) { b[t] = nb[t]; asm volatile("" : "+v"(b[t])); }
    }
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const long col = (long)(t0 + t) * 16 + li;
        if (m0 + lk < M) C[(long)(m0 + lk) * NYP + col] = cplx{acc[t][0], acc[t][2]};
        if (m0 + 4 + lk < M) C[(long)(m0 + 4 + lk) * NYP + col] = cplx{acc[t][1], acc[t][3]};
    }
}
template <int MODE>
__global__ __launch_bounds__(256) void k(const cplx* A, const double* B, cplx* C, int M, int NYP) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = blockIdx.x * 8;
    if (wave == 0) body<7, MODE>(A, B, C, M, NYP, m0, 0, lane);
    else body<6, MODE>(A, B, C, M, NYP, m0, 7, lane);
}
template <int MODE>
__global__ __launch_bounds__(256) void k4(const cplx* A, const double* B, cplx* C, int M, int NYP) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = (blockIdx.x * 2 + (wave >> 1)) * 8;
    if ((wave & 1) == 0) body<7, MODE>(A, B, C, M, NYP, m0, 0, lane);
    else body<6, MODE>(A, B, C, M, NYP, m0, 7, lane);
}
// one wave per block, 7 tiles only (half the work) to see the single-wave time
template <int MODE>
__global__ __launch_bounds__(64) void k1(const cplx* A, const double* B, cplx* C, int M, int NYP) {
    body<7, MODE>(A, B, C, M, NYP, blockIdx.x * 8, 0, threadIdx.x & 63);
}
// pure MFMA rate: 1 wave per SIMD, no memory
__global__ void k_mfma(double* out, int n) {
    d4 acc[7];
    for (int t = 0; t < 7; ++t) acc[t] = d4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int t = 0; t < 7; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
    }
    double s = 0;
    for (int t = 0; t < 7; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class F> float timeit(F f, int reps = 50) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < reps; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps * 1e3f;
}
int main() {
    const int M = 3456, NYP = 208;
    cplx *A, *C; double* B; double* out;
    hipMalloc(&A, sizeof(cplx) * 9000 * NYP); hipMalloc(&C, sizeof(cplx) * 9000 * NYP); hipMalloc(&B, 8 * NYP * NYP);
    hipMalloc(&out, 8 * 1024 * 64 * 4);
    hipMemset(A, 0, sizeof(cplx) * M * NYP); hipMemset(B, 0, 8 * NYP * NYP);
    const int groups = M / 8;
    printf("full            %.2f us\n", timeit([&] { hipLaunchKernelGGL(k<0>, dim3(groups), dim3(128), 0, 0, A, B, C, M, NYP); }));
    printf("B L1-resident   %.2f us\n", timeit([&] { hipLaunchKernelGGL(k<1>, dim3(groups), dim3(128), 0, 0, A, B, C, M, NYP); }));
    printf("no loads        %.2f us\n", timeit([&] { hipLaunchKernelGGL(k<2>, dim3(groups), dim3(128), 0, 0, A, B, C, M, NYP); }));
    printf("A loads only    %.2f us\n", timeit([&] { hipLaunchKernelGGL(k<3>, dim3(groups), dim3(128), 0, 0, A, B, C, M, NYP); }));
    printf("B loads only    %.2f us\n", timeit([&] { hipLaunchKernelGGL(k<4>, dim3(groups), dim3(128), 0, 0, A, B, C, M, NYP); }));
    printf("4-wave WG full     %.2f us\n", timeit([&] { hipLaunchKernelGGL(k4<0>, dim3(groups / 2), dim3(256), 0, 0, A, B, C, M, NYP); }));
    printf("4-wave WG no loads %.2f us\n", timeit([&] { hipLaunchKernelGGL(k4<2>, dim3(groups / 2), dim3(256), 0, 0, A, B, C, M, NYP); }));
    printf("1-wave WG x432 (7 tiles) no loads %.2f us\n", timeit([&] { hipLaunchKernelGGL(k1<2>, dim3(groups), dim3(64), 0, 0, A, B, C, M, NYP); }));
    printf("1-wave WG x256 (7 tiles) no loads %.2f us\n", timeit([&] { hipLaunchKernelGGL(k1<2>, dim3(256), dim3(64), 0, 0, A, B, C, M, NYP); }));
    printf("1-wave WG x1024 (7 tiles) no loads %.2f us\n", timeit([&] { hipLaunchKernelGGL(k1<2>, dim3(1024), dim3(64), 0, 0, A, B, C, 1024*8, NYP); }));
    printf("1-wave WG x1 (7 tiles) no loads %.2f us\n", timeit([&] { hipLaunchKernelGGL(k1<2>, dim3(1), dim3(64), 0, 0, A, B, C, M, NYP); }));
    printf("1-wave WG x1 (7 tiles) full %.2f us\n", timeit([&] { hipLaunchKernelGGL(k1<0>, dim3(1), dim3(64), 0, 0, A, B, C, M, NYP); }));
    // pure MFMA: 1024 waves (256 blocks x 256 thr), n iterations of 7 MFMA
    const int n = 52 * 4;   // same MFMA count per wave as the transform (7*4*13 = 364 -> n*7 = 1456: 4x)
    float t = timeit([&] { hipLaunchKernelGGL(k_mfma, dim3(256), dim3(256), 0, 0, out, n); });
    printf("pure mfma 1 wave/SIMD: %.2f us for %d MFMA/wave -> %.1f clk/MFMA @2.4GHz, %.1f TF\n", t, n * 7, t * 2400.0 / (n * 7),
           1024.0 * n * 7 * 2048 / (t * 1e-6) / 1e12);
    t = timeit([&] { hipLaunchKernelGGL(k_mfma, dim3(512), dim3(256), 0, 0, out, n); });
    printf("pure mfma 2 wave/SIMD: %.2f us -> %.1f TF\n", t, 2048.0 * n * 7 * 2048 / (t * 1e-6) / 1e12);
    return 0;
}
