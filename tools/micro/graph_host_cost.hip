// graph_host_cost.hip — host-side cost of hipGraphLaunch / eager launches for a 40-kernel chain whose kernels take a
// 176-byte by-value argument (like DecLinearParams) and run ~5 us each, on 1 and 2 streams.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { float* p; int n; int pad[40]; };
__global__ void work(Big b) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = b.p[i % b.n];
    for (int k = 0; k < 300; ++k) v = v * 1.0001f + 0.5f;
    if (v == 123.f) b.p[0] = v;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    float* f; CK(hipMalloc(&f, 1 << 22)); CK(hipMemset(f, 0, 1 << 22));
    Big b{f, 1 << 20, {0}};
    hipStream_t st[2]; hipGraphExec_t ge[2];
    const int N = 40, reps = 100;
    for (int s = 0; s < 2; ++s) {
        CK(hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking));
        hipGraph_t g;
        CK(hipStreamBeginCapture(st[s], hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(work, dim3(72), dim3(256), 0, st[s], b);
        CK(hipStreamEndCapture(st[s], &g)); CK(hipGraphInstantiate(&ge[s], g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge[s], st[s])); CK(hipStreamSynchronize(st[s]));
    }
    for (int ns = 1; ns <= 2; ++ns) {
        double t0 = now();
        for (int r = 0; r < reps; ++r) for (int s = 0; s < ns; ++s) CK(hipGraphLaunch(ge[s], st[s]));
        double t1 = now();
        for (int s = 0; s < ns; ++s) CK(hipStreamSynchronize(st[s]));
        double t2 = now();
        printf("graph  %d stream(s): host %.1f us per launch (%d nodes), total wall %.1f us per launch-round\n", ns,
               (t1 - t0) * 1e6 / (reps * ns), N, (t2 - t0) * 1e6 / reps);
        t0 = now();
        for (int r = 0; r < reps; ++r) for (int s = 0; s < ns; ++s) for (int i = 0; i < N; ++i) hipLaunchKernelGGL(work, dim3(72), dim3(256), 0, st[s], b);
        t1 = now();
        for (int s = 0; s < ns; ++s) CK(hipStreamSynchronize(st[s]));
        t2 = now();
        printf("eager  %d stream(s): host %.1f us per %d launches, total wall %.1f us per round\n", ns, (t1 - t0) * 1e6 / (reps * ns), N,
               (t2 - t0) * 1e6 / reps);
    }
    return 0;
}
