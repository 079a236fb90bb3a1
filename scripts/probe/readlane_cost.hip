// What does a v_readlane_b32 cost in issue slots?  (The persistent kernel reloads ~500 spilled scalars per iteration with it.)
// One 512-thread workgroup per CU (two waves per SIMD, as the kernel runs), a loop of N independent instructions of one kind per
// wave, timed with s_memtime: cycles per instruction per wave, for v_readlane_b32 (to distinct SGPRs), v_mov_b32, v_add_f32,
// v_pk_fma_f32, v_fma_f64, s_mov_b32, and a mix readlane + pk_fma.
// hipcc --offload-arch=gfx950 -O2 -o readlane_cost readlane_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int KIND>
__global__ __launch_bounds__(512) void k(long long* out, int iters) {
    float a = threadIdx.x, b = 1.0f, c = 2.0f, d = 3.0f;
    double e = threadIdx.x, f = 1.0;
    unsigned u = threadIdx.x;
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) { REP64(asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %0, 5\n v_readlane_b32 s22, %0, 7\n v_readlane_b32 s23, %0, 9" :: "v"(u) : "s20", "s21", "s22", "s23");) }
        if (KIND == 1) { REP64(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %2, %1\n v_mov_b32 %3, %1\n v_mov_b32 %4, %1" : "=v"(a), "+v"(u), "=v"(b), "=v"(c), "=v"(d));) }
        if (KIND == 2) { REP64(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(1.0f));) }
        if (KIND == 3) { REP64(asm volatile("v_fma_f64 %0, %0, %2, %1\n v_fma_f64 %1, %1, %2, %0\n v_fma_f64 %0, %0, %2, %1\n v_fma_f64 %1, %1, %2, %0" : "+v"(e), "+v"(f) : "v"(1.0000001));) }
        if (KIND == 4) { REP64(asm volatile("s_mov_b32 s20, 3\n s_mov_b32 s21, 5\n s_mov_b32 s22, 7\n s_mov_b32 s23, 9" ::: "s20", "s21", "s22", "s23");) }
        if (KIND == 5) { REP64(asm volatile("v_readlane_b32 s20, %4, 3\n v_add_f32 %0, %0, %5\n v_readlane_b32 s21, %4, 5\n v_add_f32 %1, %1, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(u), "v"(1.0f) : "s20", "s21");) }
    }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x % 64 == 0) out[blockIdx.x * 8 + threadIdx.x / 64] = t1 - t0;
    if (a + b + c + d + e + f + u == -1.0) out[0] = 0;
}
template <int KIND> void run(const char* name, long long* d, int perIter) {
    const int iters = 200;
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, d, iters);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, d, iters);
    hipDeviceSynchronize();
    std::vector<long long> h(256 * 8);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += v;
    printf("%-28s %7.2f shader-clock ticks per instruction per wave (two waves per SIMD)\n", name, s / h.size() / ((double)iters * perIter));
}
int main() {
    long long* d; hipMalloc(&d, 256 * 8 * 8);
    run<0>("v_readlane_b32", d, 256); run<1>("v_mov_b32", d, 256); run<2>("v_add_f32", d, 256); run<3>("v_fma_f64 (dependent pairs)", d, 256);
    run<4>("s_mov_b32", d, 256); run<5>("readlane + v_add interleaved", d, 256);
    return 0;
}
