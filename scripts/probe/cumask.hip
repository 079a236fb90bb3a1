// probe: how does a CU-masked stream (hipExtStreamCreateWithCUMask) spread workgroups over the XCDs of an MI355X?
// hipcc --offload-arch=gfx950 -O2 scripts/probe/cumask.hip -o /tmp/cumask && /tmp/cumask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_where(unsigned* out, long long spin) {
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    extern __shared__ char sm[];
    sm[threadIdx.x] = 1;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) out[blockIdx.x] = (xcc & 0xf) | (hwid << 4);
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("CUs %d\n", p.multiProcessorCount);
    hipFuncSetAttribute((const void*)k_where, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int variant = 0; variant < 4; ++variant) {
        // variant 0: no mask; 1: bits with (bit % 8) < 4; 2: bits with (bit % 8) >= 4; 3: the first 128 bits
        std::vector<uint32_t> mask(8, 0);
        for (int b = 0; b < 256; ++b) {
            bool on = variant == 0 ? true : variant == 1 ? (b % 8) < 4 : variant == 2 ? (b % 8) >= 4 : b < 128;
            if (on) mask[b / 32] |= 1u << (b % 32);
        }
        hipStream_t s;
        if (variant == 0) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        else if (hipExtStreamCreateWithCUMask(&s, 8, mask.data()) != hipSuccess) { printf("variant %d: create failed\n", variant); continue; }
        const int nb = 256;
        unsigned* d; hipMalloc(&d, nb * 4); hipMemset(d, 0xff, nb * 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, s);
        hipLaunchKernelGGL(k_where, dim3(nb), dim3(64), 160 * 1024, s, d, 200000ll);   // 2 ms per workgroup, one workgroup per CU
        hipEventRecord(e1, s);
        hipStreamSynchronize(s);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned> h(nb); hipMemcpy(h.data(), d, nb * 4, hipMemcpyDeviceToHost);
        printf("variant %d: xcc of block 0..31:", variant);
        for (int i = 0; i < 32; ++i) printf(" %u", h[i] & 0xf);
        int cnt[16] = {0}; for (int i = 0; i < nb; ++i) cnt[h[i] & 0xf]++;
        printf("  | per-xcc counts:"); for (int i = 0; i < 8; ++i) printf(" %d", cnt[i]);
        std::vector<unsigned> ids; for (int i = 0; i < nb; ++i) { unsigned id = (h[i] & 0xf) | (((h[i] >> 4) >> 8) & 0xff) << 4; bool f = false; for (unsigned q : ids) f |= q == id; if (!f) ids.push_back(id); }
        printf("  | distinct (xcc, cu/sh/se) %zu  | %.2f ms\n", ids.size(), ms);
        hipFree(d); hipStreamDestroy(s);
    }
    return 0;
}
