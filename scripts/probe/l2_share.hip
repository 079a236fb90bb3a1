// Do two workgroups on the SAME XCD that read the same bytes at the same time share them in that XCD's L2, i.e. does the
// data cross the fabric once?  Every workgroup streams REG bytes (16 B per lane, coalesced) into registers and writes one
// checksum.  Case "own": workgroup b reads region b.  Case "pair-same-xcd": workgroups b and b+8 (same XCD under the
// round-robin dispatch) read the same region.  Case "pair-other-xcd": workgroups b and b+1 read the same region.
// Total distinct bytes in the pair cases = half.  hipcc --offload-arch=gfx950 -O2 -o l2_share l2_share.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int REG = 128 * 1024;            // bytes per workgroup
__global__ __launch_bounds__(256) void k(const float4* __restrict__ src, float* out, int mode) {
    const int b = blockIdx.x;
    int region;
    if (mode == 0) region = b;
    else if (mode == 1) region = (b & 7) + 8 * ((b >> 3) >> 1);          // b and b+8 share
    else region = b >> 1;                                                 // b and b+1 share
    const float4* p = src + (size_t)region * (REG / 16);
    float4 acc = make_float4(0, 0, 0, 0);
    float4 v[8];
    for (int i0 = threadIdx.x; i0 < REG / 16; i0 += 8 * 256) {
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[i0 + u * 256];
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[b] = acc.x;
}
int main() {
    const int NWG = 512;
    float4* src; float* out;
    hipMalloc(&src, (size_t)NWG * REG); hipMalloc(&out, NWG * sizeof(float));
    hipMemset(src, 0, (size_t)NWG * REG);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[3] = {"own region per workgroup (64 MB distinct)", "b and b+8 share (same XCD; 32 MB distinct)", "b and b+1 share (other XCD; 32 MB distinct)"};
    for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 3; ++mode) {
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k, dim3(NWG), dim3(256), 0, 0, src, out, mode);
        hipEventRecord(e0);
        for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(k, dim3(NWG), dim3(256), 0, 0, src, out, mode);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-48s %.2f us per launch  (%.2f TB/s of requested bytes)\n", names[mode], 1e3 * ms / 20, (double)NWG * REG / (ms / 20 * 1e-3) / 1e12);
    }
    return 0;
}
