// Micro-benchmark: cost of a dependent chain of tiny kernels on one stream -- plain launches vs a captured hipGraph.
// Build: hipcc -O2 --offload-arch=gfx950 launch_floor.hip -o launch_floor ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); return 1; } } while (0)

__global__ void tiny(float* p, int n) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = p[0] + 1.0f; }
__global__ void medium(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.0f; }

int main()
{
    float* d; const int n = 1 << 20;
    CK(hipMalloc(&d, n * 4)); CK(hipMemset(d, 0, n * 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    const int K = 2000;
    for (int variant = 0; variant < 2; variant++) {
        auto launch = [&](hipStream_t st) { if (variant == 0) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, d, n); else hipLaunchKernelGGL(medium, dim3(n / 256), dim3(256), 0, st, d, n); };
        for (int i = 0; i < 200; i++) launch(s);
        CK(hipStreamSynchronize(s));
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < K; i++) launch(s);
        CK(hipStreamSynchronize(s));
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("%s: stream launches  %.2f us per kernel\n", variant ? "medium(4MB rw)" : "tiny", us / K);
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 100; i++) launch(s);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 3; i++) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < K / 100; i++) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("%s: graph of 100     %.2f us per kernel\n", variant ? "medium(4MB rw)" : "tiny", us / K);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
