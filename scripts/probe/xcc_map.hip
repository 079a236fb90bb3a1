// Which XCD does workgroup b of a launch run on?  (s_getreg HW_REG_XCC_ID per workgroup; 1-D and 2-D grids, with and
// without a delay that keeps all workgroups resident)  hipcc --offload-arch=gfx950 -O2 -o xcc_map xcc_map.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(int* out, int spin) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    const int b = blockIdx.x + gridDim.x * blockIdx.y;
    if (threadIdx.x == 0) out[b] = (int)(x & 0xf);
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
}
int main() {
    int* d; const int N = 4096; hipMalloc(&d, N * sizeof(int));
    std::vector<int> h(N);
    struct { dim3 g; int spin; const char* name; } cases[] = {{dim3(1920), 0, "1-D 1920"}, {dim3(1920), 500, "1-D 1920, 5 us spin"},
        {dim3(30, 64), 0, "2-D 30x64"}, {dim3(15, 32), 500, "2-D 15x32, 5 us spin"}};
    for (auto& c : cases) {
        hipMemset(d, 0xff, N * sizeof(int));
        hipLaunchKernelGGL(k, c.g, dim3(256), 0, 0, d, c.spin);
        hipMemcpy(h.data(), d, N * sizeof(int), hipMemcpyDeviceToHost);
        const int n = c.g.x * c.g.y;
        int same = 0;                      // workgroups whose XCD equals that of workgroup (b % 8)
        for (int b = 0; b < n; ++b) same += h[b] == h[b % 8];
        printf("%-24s first 24:", c.name);
        for (int b = 0; b < 24; ++b) printf(" %d", h[b]);
        printf("   | XCD(b) == XCD(b %% 8) for %d of %d\n", same, n);
    }
    return 0;
}
