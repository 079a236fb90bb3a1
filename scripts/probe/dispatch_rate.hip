// How long does the dispatcher need per workgroup?  Kernels whose workgroups return at once (or after one global load, as
// the workgroups of a converged system do), N workgroups of T threads, with and without dynamic LDS; time per launch from
// 200 back-to-back launches.  hipcc --offload-arch=gfx950 -O2 -o dispatch_rate dispatch_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_empty() {}
__global__ void k_flag(const int* flag, int* out) { if (flag[blockIdx.x & 31]) out[blockIdx.x] = 1; }
__global__ void k_flag_lds(const int* flag, int* out) {
    extern __shared__ char sm[];
    if (flag[blockIdx.x & 31]) { sm[threadIdx.x] = 1; out[blockIdx.x] = sm[0]; }
}
int main() {
    int *flag, *out; hipMalloc(&flag, 256); hipMalloc(&out, 1 << 20); hipMemset(flag, 0, 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, int which, int n, int t, size_t lds) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            for (int i = 0; i < 200; ++i) {
                if (which == 0) hipLaunchKernelGGL(k_empty, dim3(n), dim3(t), 0, 0);
                else if (which == 1) hipLaunchKernelGGL(k_flag, dim3(n), dim3(t), 0, 0, flag, out);
                else hipLaunchKernelGGL(k_flag_lds, dim3(n), dim3(t), lds, 0, flag, out);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%-28s %5d workgroups x %4d threads, LDS %6zu B: %6.2f us per launch\n", name, n, t, lds, 1e3 * ms / 200);
        }
    };
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_flag_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int n : {1, 128, 256, 512, 1024, 2048}) run("empty", 0, n, 256, 0);
    for (int n : {256, 512, 1024}) run("empty", 0, n, 512, 0);
    for (int n : {256, 512, 1024}) run("one flag load, exit", 1, n, 256, 0);
    for (int n : {256, 512, 1024}) run("one flag load, exit", 1, n, 512, 0);
    for (int n : {256, 512}) run("flag load, 60 KB LDS", 2, n, 512, 60 * 1024);
    for (int n : {256, 512}) run("flag load, 30 KB LDS", 2, n, 256, 30 * 1024);
    return 0;
}
