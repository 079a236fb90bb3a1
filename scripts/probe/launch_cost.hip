// Host-side cost of kernel launches on this box: plain launches with small / 400 B / 1 KB by-value arguments,
// and a hipGraph of 5 kernel nodes replayed.  Kernels are ~8 us busy loops so the queue stays full.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct A400 { double v[50]; };
struct A1k { double v[128]; };
__global__ void k0(int n, float* o) { float a = threadIdx.x; for (int i = 0; i < n; ++i) a = a * 1.0001f + 0.5f; if (a == 123.f) o[0] = a; }
__global__ void k400(A400 s, int n, float* o) { float a = threadIdx.x + (float)s.v[3]; for (int i = 0; i < n; ++i) a = a * 1.0001f + 0.5f; if (a == 123.f) o[0] = a; }
__global__ void k1k(A1k s, int n, float* o) { float a = threadIdx.x + (float)s.v[3]; for (int i = 0; i < n; ++i) a = a * 1.0001f + 0.5f; if (a == 123.f) o[0] = a; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    float* o; hipMalloc(&o, 4);
    hipStream_t st; hipStreamCreate(&st);
    A400 a4{}; A1k a1{};
    const int N = 2000;
    for (int busy : {0, 2000}) {
        for (int which = 0; which < 3; ++which) {
            hipStreamSynchronize(st);
            double t0 = now();
            for (int i = 0; i < N; ++i) {
                if (which == 0) hipLaunchKernelGGL(k0, dim3(256), dim3(256), 0, st, busy, o);
                else if (which == 1) hipLaunchKernelGGL(k400, dim3(256), dim3(256), 0, st, a4, busy, o);
                else hipLaunchKernelGGL(k1k, dim3(256), dim3(256), 0, st, a1, busy, o);
            }
            double t1 = now();
            hipStreamSynchronize(st);
            double t2 = now();
            printf("busy %4d args %s: host %.2f us/launch, end-to-end %.2f us/kernel\n", busy, which == 0 ? "16 B " : which == 1 ? "400 B" : "1 KB ",
                   1e6 * (t1 - t0) / N, 1e6 * (t2 - t0) / N);
        }
    }
    // graph of 5 kernels
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k400, dim3(256), dim3(256), 0, st, a4, 2000, o);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    double t0 = now();
    for (int i = 0; i < N / 5; ++i) hipGraphLaunch(ge, st);
    double t1 = now();
    hipStreamSynchronize(st);
    double t2 = now();
    printf("graph of 5 x k400(busy 2000): host %.2f us/kernel, end-to-end %.2f us/kernel\n", 1e6 * (t1 - t0) / N, 1e6 * (t2 - t0) / N);
    return 0;
}
