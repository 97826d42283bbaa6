// launch_floor.hip — measures the dependent-kernel boundary cost on this box: a chain of N trivial kernels on one
// stream, eager and as a replayed hipGraph, for 1 / 256 / 2048 workgroups.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void tiny(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ void touch(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    int* d; float* f;
    CK(hipMalloc(&d, 4096)); CK(hipMemset(d, 0, 4096));
    CK(hipMalloc(&f, 1 << 22)); CK(hipMemset(f, 0, 1 << 22));
    for (int nb : {0, 1}) {
        hipStream_t st;
        CK(hipStreamCreateWithFlags(&st, nb ? hipStreamNonBlocking : hipStreamDefault));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int wgs : {1, 256, 2048}) {
            const int N = 40, reps = 50;
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(wgs), dim3(256), 0, st, d);
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int r = 0; r < reps; ++r) for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(wgs), dim3(256), 0, st, d);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("stream %s  %4d WGs  eager: %.2f us/kernel\n", nb ? "nonblocking" : "default", wgs, ms * 1e3 / (N * reps));
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(wgs), dim3(256), 0, st, d);
            CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("stream %s  %4d WGs  graph: %.2f us/kernel\n", nb ? "nonblocking" : "default", wgs, ms * 1e3 / (N * reps));
            // a kernel that dirties 256 KB
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(touch, dim3(256), dim3(256), 0, st, f, 65536);
            CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (wgs == 256) printf("stream %s  touch 256KB graph: %.2f us/kernel\n", nb ? "nonblocking" : "default", ms * 1e3 / (N * reps));
        }
    }
    return 0;
}
