// Does a hipGraph shorten a chain of DEPENDENT launches on MI355X?  N launches of a kernel of 288 workgroups x 256 threads
// (the shape of one COCG iteration kernel) touching `bytes` of memory each, (a) launched into a stream one by one,
// (b) captured once into a graph and launched as a whole.  Prints microseconds per launch.
//   hipcc --offload-arch=gfx950 -O2 -o graph_chain graph_chain.hip && ./graph_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); return 1; } } while (0)
__global__ __launch_bounds__(256) void k_touch(const double2* __restrict__ a, double2* __restrict__ b, long n, double s) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
        double2 v = a[i];
        b[i] = double2{v.x * s + 1.0, v.y * s - 1.0};
    }
}
int main() {
    const int N = 400, WG = 288;
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (long n : {0L, 1L << 16, 1L << 20, 3L << 20}) {           // 0, 1 MB, 16 MB, 48 MB read (+ the same written)
        double2 *a, *b; CK(hipMalloc(&a, (n + 1) * 16)); CK(hipMalloc(&b, (n + 1) * 16));
        CK(hipMemset(a, 0, (n + 1) * 16));
        auto chain = [&]() { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_touch, dim3(WG), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, n, 0.5); };
        chain(); CK(hipStreamSynchronize(st));
        float msS = 1e9f, msG = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0, st)); chain(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < msS) msS = ms;
        }
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal)); chain(); CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < msG) msG = ms;
        }
        printf("%8.1f MB read + written per launch: stream %.2f us/launch, graph %.2f us/launch\n", n * 16 / 1e6, msS * 1e3 / N, msG * 1e3 / N);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipFree(a)); CK(hipFree(b));
    }
    return 0;
}
