// Diagnostic: what a chain of N dependent tiny kernels costs end to end, launched one by one on a stream and as one hipGraph
// (the frame-by-frame call of the detector is 21 such launches).   hipcc --offload-arch=gfx950 -O3 -o build/launch_floor tools/launch_floor.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

__global__ void k_tiny(unsigned int *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 21, reps = 2000;
    unsigned int *d;
    CHK(hipMalloc(&d, 4));
    CHK(hipMemset(d, 0, 4));
    hipStream_t st;
    CHK(hipStreamCreate(&st));
    for (int w = 0; w < 50; w++) { for (int k = 0; k < n; k++) hipLaunchKernelGGL(k_tiny, dim3(64), dim3(64), 0, st, d); CHK(hipStreamSynchronize(st)); }
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; r++) { for (int k = 0; k < n; k++) hipLaunchKernelGGL(k_tiny, dim3(64), dim3(64), 0, st, d); CHK(hipStreamSynchronize(st)); }
    const double ms_stream = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
    hipGraph_t g;
    hipGraphExec_t ge;
    CHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < n; k++) hipLaunchKernelGGL(k_tiny, dim3(64), dim3(64), 0, st, d);
    CHK(hipStreamEndCapture(st, &g));
    CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 50; w++) { CHK(hipGraphLaunch(ge, st)); CHK(hipStreamSynchronize(st)); }
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; r++) { CHK(hipGraphLaunch(ge, st)); CHK(hipStreamSynchronize(st)); }
    const double ms_graph = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
    printf("%d dependent tiny kernels + synchronize: stream %.4f ms (%.2f us per kernel), graph %.4f ms (%.2f us per kernel)\n", n, ms_stream, 1e3 * ms_stream / n, ms_graph,
           1e3 * ms_graph / n);
    return 0;
}
