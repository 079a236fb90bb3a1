// Is a captured multi-stream hipGraph cheaper for the host than issuing the same calls?  The shape of the library's
// prologue: main: A, record e0, B, [side: wait e0, C, D, record e1], [side2: wait e0, E, record e2, F, record e3],
// main: wait e1, wait e2, G, wait e3 -- seven kernels with ~1 KB of by-value arguments, 3 streams, 4 events.
// hipcc --offload-arch=gfx950 -O2 -o graph_prologue graph_prologue.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct A1k { double v[120]; };
__global__ void kk(A1k s, int n, float* o) { float a = threadIdx.x + (float)s.v[3]; for (int i = 0; i < n; ++i) a = a * 1.0001f + 0.5f; if (a == 123.f) o[0] = a; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    float* o; hipMalloc(&o, 4);
    hipStream_t st, s1, s2;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking); hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    hipEvent_t e[4]; for (auto& x : e) hipEventCreateWithFlags(&x, hipEventDisableTiming);
    A1k a{};
    const int busy = 600;          // ~ a few us per kernel
    auto issue = [&]() {
        hipLaunchKernelGGL(kk, dim3(64), dim3(64), 0, st, a, busy, o);
        hipEventRecord(e[0], st);
        hipLaunchKernelGGL(kk, dim3(256), dim3(256), 0, st, a, 4 * busy, o);
        hipStreamWaitEvent(s1, e[0], 0);
        hipLaunchKernelGGL(kk, dim3(32), dim3(256), 0, s1, a, busy, o);
        hipLaunchKernelGGL(kk, dim3(512), dim3(256), 0, s1, a, busy, o);
        hipEventRecord(e[1], s1);
        hipStreamWaitEvent(s2, e[0], 0);
        hipLaunchKernelGGL(kk, dim3(512), dim3(256), 0, s2, a, busy, o);
        hipEventRecord(e[2], s2);
        hipLaunchKernelGGL(kk, dim3(128), dim3(64), 0, s2, a, 3 * busy, o);
        hipEventRecord(e[3], s2);
        hipStreamWaitEvent(st, e[1], 0);
        hipStreamWaitEvent(st, e[2], 0);
        hipLaunchKernelGGL(kk, dim3(256), dim3(256), 0, st, a, busy, o);
        hipStreamWaitEvent(st, e[3], 0);
    };
    const int N = 300;
    for (int rep = 0; rep < 2; ++rep) {
        hipStreamSynchronize(st);
        double host = 0, t0 = now();
        for (int i = 0; i < N; ++i) { double h0 = now(); issue(); host += now() - h0; hipStreamSynchronize(st); }
        double t1 = now();
        if (rep) printf("direct: host %.1f us per sequence, end-to-end %.1f us\n", 1e6 * host / N, 1e6 * (t1 - t0) / N);
    }
    auto issue1 = [&]() {       // one stream, no events
        hipLaunchKernelGGL(kk, dim3(64), dim3(64), 0, st, a, busy, o);
        hipLaunchKernelGGL(kk, dim3(256), dim3(256), 0, st, a, 4 * busy, o);
        hipLaunchKernelGGL(kk, dim3(32), dim3(256), 0, st, a, busy, o);
        hipLaunchKernelGGL(kk, dim3(512), dim3(256), 0, st, a, busy, o);
        hipLaunchKernelGGL(kk, dim3(512), dim3(256), 0, st, a, busy, o);
        hipLaunchKernelGGL(kk, dim3(128), dim3(64), 0, st, a, 3 * busy, o);
        hipLaunchKernelGGL(kk, dim3(256), dim3(256), 0, st, a, busy, o);
    };
    auto issue2 = [&]() {       // two streams, two events
        hipLaunchKernelGGL(kk, dim3(64), dim3(64), 0, st, a, busy, o);
        hipEventRecord(e[0], st);
        hipLaunchKernelGGL(kk, dim3(256), dim3(256), 0, st, a, 4 * busy, o);
        hipStreamWaitEvent(s1, e[0], 0);
        hipLaunchKernelGGL(kk, dim3(32), dim3(256), 0, s1, a, busy, o);
        hipLaunchKernelGGL(kk, dim3(512), dim3(256), 0, s1, a, busy, o);
        hipLaunchKernelGGL(kk, dim3(512), dim3(256), 0, s1, a, busy, o);
        hipLaunchKernelGGL(kk, dim3(128), dim3(64), 0, s1, a, 3 * busy, o);
        hipEventRecord(e[1], s1);
        hipStreamWaitEvent(st, e[1], 0);
        hipLaunchKernelGGL(kk, dim3(256), dim3(256), 0, st, a, busy, o);
    };
    auto timeit = [&](auto f, const char* name) {
        for (int rep = 0; rep < 2; ++rep) {
            hipStreamSynchronize(st);
            double host = 0, t0 = now();
            for (int i = 0; i < N; ++i) { double h0 = now(); f(); host += now() - h0; hipStreamSynchronize(st); }
            double t1 = now();
            if (rep) printf("%s: host %.1f us per sequence, end-to-end %.1f us\n", name, 1e6 * host / N, 1e6 * (t1 - t0) / N);
        }
    };
    timeit(issue1, "one stream, 7 kernels, no events");
    timeit(issue2, "two streams, 2 records + 2 waits");
    for (int nb : {0, 1}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipStreamSynchronize(st);
            double t0 = now();
            for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(kk, dim3(64), dim3(64), 0, st, a, nb ? busy : 0, o); hipStreamSynchronize(st); }
            double t1 = now();
            if (rep) printf("one kernel (%s) + synchronize: %.1f us\n", nb ? "busy" : "empty", 1e6 * (t1 - t0) / N);
        }
    }
    hipGraph_t g; hipGraphExec_t ge;
    if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) { printf("capture failed\n"); return 1; }
    issue();
    if (hipStreamEndCapture(st, &g) != hipSuccess) { printf("end capture failed\n"); return 1; }
    if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) { printf("instantiate failed\n"); return 1; }
    size_t nn = 0; hipGraphGetNodes(g, nullptr, &nn); printf("graph: %zu nodes\n", nn);
    for (int rep = 0; rep < 2; ++rep) {
        hipStreamSynchronize(st);
        double host = 0, t0 = now();
        for (int i = 0; i < N; ++i) { double h0 = now(); hipGraphLaunch(ge, st); host += now() - h0; hipStreamSynchronize(st); }
        double t1 = now();
        if (rep) printf("graph:  host %.1f us per sequence, end-to-end %.1f us\n", 1e6 * host / N, 1e6 * (t1 - t0) / N);
    }
    return 0;
}
