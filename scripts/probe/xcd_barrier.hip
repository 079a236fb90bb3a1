// A barrier among the workgroups of ONE XCD (no data crosses an L2): what would it cost next to a kernel boundary?
// 256 workgroups x 512 threads, workgroup b on XCD b % 8 (round-robin dispatch; checked with XCC_ID), groups of 32.
// Every phase a workgroup reads the 64 KB its group neighbour wrote in the previous phase and writes its own 64 KB.
//   A  one launch per phase
//   C  persistent, per-XCD barrier: stores complete (vmcnt 0) -> atomic add in L2 (workgroup scope: no L2 write-back, no
//      invalidate) -> poll with a never-matching compare-and-swap (never served by the L1) -> loads of the neighbour's chunk with
//      the L1 bypassed (sc0 / "glc": __builtin_nontemporal_load is not enough; an agent-scope atomic load per element is)
// hipcc --offload-arch=gfx950 -O2 -o xcd_barrier xcd_barrier.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int NB = 256, NT = 512, PER = 16;            // 8-byte words per thread (64 KB per workgroup)
constexpr long CH = (long)NT * PER;

__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }

template <bool BYPASS>
__device__ __forceinline__ void phase(const unsigned long long* in, unsigned long long* out, int b, int it) {
    const int grp = b & 7, j = b >> 3;                  // group = XCD, j = index within the group (0..31)
    const int src = (((j * 5 + 1 + it) & 31) << 3) | grp;      // a neighbour of the same group
    unsigned long long v[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const unsigned long long* p = &in[src * CH + q * NT + threadIdx.x];
        v[q] = BYPASS ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
    }
#pragma unroll
    for (int q = 0; q < PER; ++q) out[b * CH + q * NT + threadIdx.x] = v[q] * 3 + 1 + q;
}
__global__ __launch_bounds__(NT) void k_phase(const unsigned long long* in, unsigned long long* out, int it) { phase<false>(in, out, blockIdx.x, it); }

__device__ __forceinline__ void xcd_barrier(unsigned* cnt, unsigned target) {
    __builtin_amdgcn_s_waitcnt(0);                      // this wave's stores have reached the L2
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        // (a compare-and-swap that never matches: it returns the value in the L2; an atomic add of 0 is folded into a load)
        for (;;) {
            unsigned expect = 0xffffffffu;
            __hip_atomic_compare_exchange_strong(cnt, &expect, 0u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (expect >= target) break;
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
}
__global__ __launch_bounds__(NT) void k_persist(unsigned long long* a, unsigned long long* b, int nphase, unsigned* cnt, int work, unsigned* xcc) {
    extern __shared__ char lds[];
    if (threadIdx.x == 0) { lds[0] = 0; xcc[blockIdx.x] = xcc_id(); }
    unsigned* c = cnt + 32 * (blockIdx.x & 7);         // one counter (own 128-byte line) per group
    for (int it = 0; it < nphase; ++it) {
        if (work) phase<true>((it & 1) ? b : a, (it & 1) ? a : b, blockIdx.x, it);
        xcd_barrier(c, 32u * (it + 1));
    }
}
int main() {
    unsigned long long *a, *b; unsigned *sync, *xcc;
    const size_t bytes = sizeof(unsigned long long) * NB * CH;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&sync, 4096); hipMalloc(&xcc, NB * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_persist), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    const int NP = 400;
    std::vector<unsigned long long> ra(NB * CH), rb(NB * CH);
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
        hipEventRecord(e0);
        for (int it = 0; it < NP; ++it) hipLaunchKernelGGL(k_phase, dim3(NB), dim3(NT), 0, 0, (it & 1) ? b : a, (it & 1) ? a : b, it);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("A  one launch per phase:                       %6.2f us per phase\n", 1e3 * ms / NP);
    }
    hipMemcpy(ra.data(), a, bytes, hipMemcpyDeviceToHost);
    for (int work : {1, 0}) for (int rep = 0; rep < 2; ++rep) {
        hipMemset(a, 0, bytes); hipMemset(b, 0, bytes); hipMemset(sync, 0, 4096);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_persist, dim3(NB), dim3(NT), 150 * 1024, 0, a, b, NP, sync, work, xcc);
        hipEventRecord(e1);
        if (hipEventSynchronize(e1) != hipSuccess) { printf("persistent launch failed\n"); return 1; }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("C  persistent, per-XCD barrier, %s: %6.2f us per phase\n", work ? "with the phase work" : "barrier only       ", 1e3 * ms / NP);
        if (rep && work) {
            hipMemcpy(rb.data(), a, bytes, hipMemcpyDeviceToHost);
            long bad = 0; for (size_t i = 0; i < ra.size(); ++i) bad += ra[i] != rb[i];
            printf("   results differ from A in %ld of %zu values\n", bad, ra.size());
            std::vector<unsigned> hx(NB); hipMemcpy(hx.data(), xcc, NB * 4, hipMemcpyDeviceToHost);
            int mism = 0; for (int i = 0; i < NB; ++i) mism += hx[i] != hx[i & 7];
            printf("   workgroups whose XCC_ID differs from that of workgroup (b %% 8): %d of %d\n", mism, NB);
        }
    }
    return 0;
}
