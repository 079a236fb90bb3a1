// What does a grid-wide barrier cost on MI355X next to a kernel boundary?  256 workgroups x 512 threads (one per CU, as a
// persistent COCG iteration kernel would run): every phase each workgroup reads the 64 KB another workgroup wrote in the
// previous phase and writes its own 64 KB (so the barrier has to make data visible across the 8 XCDs' L2s).
//   A  one launch per phase (the kernel boundary is the barrier)
//   B  one persistent launch, phases separated by an atomic-counter barrier (release/acquire fences at agent scope)
// hipcc --offload-arch=gfx950 -O2 -o grid_barrier grid_barrier.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int NB = 256, NT = 512, PER = 8;            // float4 per thread
constexpr long CH = (long)NT * PER;                  // float4 per workgroup chunk (64 KB)

__device__ __forceinline__ void phase(const float4* in, float4* out, int b, int it) {
    const int src = (b * 37 + 1 + it) % NB;
    float4 v[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) v[q] = in[src * CH + q * NT + threadIdx.x];
#pragma unroll
    for (int q = 0; q < PER; ++q) { v[q].x += 1.f; v[q].y += v[q].x * 0.5f; out[b * CH + q * NT + threadIdx.x] = v[q]; }
}
__global__ __launch_bounds__(NT) void k_phase(const float4* in, float4* out, int it) { phase(in, out, blockIdx.x, it); }

template <int MODE>     // 0: every thread fences (agent scope) on both sides; 1: thread 0 only; 2: thread 0, no explicit fences (acq_rel atomics only)
__device__ __forceinline__ void grid_barrier(unsigned* count, unsigned* gen) {
    if (MODE == 0) __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        if (MODE == 1) __threadfence();
        const unsigned g = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_fetch_add(count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
            __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(gen, g + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(gen, MODE == 2 ? __ATOMIC_ACQUIRE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g) __builtin_amdgcn_s_sleep(1);
        }
        if (MODE == 1) __threadfence();
    }
    __syncthreads();
    if (MODE == 0) __threadfence();
}
template <int MODE>
__global__ __launch_bounds__(NT) void k_persist(float4* a, float4* b, int nphase, unsigned* count, unsigned* gen, int work) {
    extern __shared__ char lds[];
    if (threadIdx.x == 0) lds[0] = 0;
    for (int it = 0; it < nphase; ++it) {
        if (work) phase((it & 1) ? b : a, (it & 1) ? a : b, blockIdx.x, it);
        grid_barrier<MODE>(count, gen);
    }
}
int main() {
    float4 *a, *b; unsigned* sync;
    const size_t bytes = sizeof(float4) * NB * CH;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&sync, 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_persist<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    const int NP = 400;
    std::vector<float> ra(NB * CH * 4), rb(NB * CH * 4);
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
        hipEventRecord(e0);
        for (int it = 0; it < NP; ++it) hipLaunchKernelGGL(k_phase, dim3(NB), dim3(NT), 0, 0, (it & 1) ? b : a, (it & 1) ? a : b, it);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("A  one launch per phase:          %6.2f us per phase\n", 1e3 * ms / NP);
    }
    hipMemcpy(ra.data(), a, bytes, hipMemcpyDeviceToHost);
    auto run = [&](auto kern, const char* name) {
        for (int work : {1, 0}) for (int rep = 0; rep < 2; ++rep) {
            hipMemset(a, 0, bytes); hipMemset(b, 0, bytes); hipMemset(sync, 0, 256);
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(NB), dim3(NT), 150 * 1024, 0, a, b, NP, sync, sync + 32, work);
            hipEventRecord(e1);
            if (hipEventSynchronize(e1) != hipSuccess) { printf("persistent launch failed\n"); return; }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("B  persistent (%s), %s: %6.2f us per phase\n", name, work ? "with the phase work" : "barrier only      ", 1e3 * ms / NP);
            if (rep && work) {
                hipMemcpy(rb.data(), a, bytes, hipMemcpyDeviceToHost);
                long bad = 0; for (size_t i = 0; i < ra.size(); ++i) bad += ra[i] != rb[i];
                printf("   results differ from A in %ld of %zu values\n", bad, ra.size());
            }
        }
    };
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_persist<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_persist<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    run(k_persist<1>, "thread 0 fences");
    run(k_persist<2>, "acq_rel atomics only");
    run(k_persist<0>, "every thread fences");
    return 0;
}
