// graph_block_probe.hip — which graph property makes hipGraphLaunch block the host until the previous replay ends?
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
__global__ void work_dyn(Big b) {
    extern __shared__ float sm[];
    sm[threadIdx.x] = b.p[threadIdx.x];
    __syncthreads();
    float v = sm[(threadIdx.x + 1) % blockDim.x];
    for (int k = 0; k < 300; ++k) v = v * 1.0001f + 0.5f;
    if (v == 123.f) b.p[0] = v;
}
__global__ __launch_bounds__(1024) void work_1024(Big b) {
    float v = b.p[threadIdx.x];
    for (int k = 0; k < 300; ++k) v = v * 1.0001f + 0.5f;
    if (v == 123.f) b.p[0] = v;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    float* f; CK(hipMalloc(&f, 1 << 22)); CK(hipMemset(f, 0, 1 << 22));
    Big b{f, 1 << 20, {0}};
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int variant = 0; variant < 5; ++variant) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        int nodes = 0;
        for (int i = 0; i < 38; ++i) {
            if (variant == 1 && i == 37) hipLaunchKernelGGL(work_dyn, dim3(72), dim3(256), 50 * 1024, st, b);
            else if (variant == 2 && i == 37) hipLaunchKernelGGL(work_1024, dim3(64), dim3(1024), 0, st, b);
            else if (variant == 3 && i % 9 == 5) hipLaunchKernelGGL(work, dim3(3072), dim3(256), 0, st, b);
            else if (variant == 4) hipLaunchKernelGGL(work, dim3(72 + i), dim3(192 + 64 * (i % 2)), 0, st, b);
            else hipLaunchKernelGGL(work, dim3(72), dim3(256), 0, st, b);
            ++nodes;
        }
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        const int reps = 100;
        double t0 = now();
        for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
        double t1 = now();
        CK(hipStreamSynchronize(st));
        double t2 = now();
        printf("variant %d (%d nodes): host %.1f us per launch, wall %.1f us per launch\n", variant, nodes, (t1 - t0) * 1e6 / reps, (t2 - t0) * 1e6 / reps);
    }
    return 0;
}
