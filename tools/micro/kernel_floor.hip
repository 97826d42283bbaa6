// kernel_floor.hip — what a dependent chain of small kernels costs per kernel (graph replay), by what the kernel does.
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { float* p; float* q; int n; int pad[38]; };
__global__ void k_nothing(Big b) {}
__global__ void k_store(Big b) { int i = blockIdx.x * blockDim.x + threadIdx.x; b.q[i] = 1.0f; }
__global__ void k_load_store(Big b) { int i = blockIdx.x * blockDim.x + threadIdx.x; b.q[i] = b.p[i] + 1.0f; }
__global__ void k_dep2(Big b) { int i = blockIdx.x * blockDim.x + threadIdx.x; int j = (int)b.p[i] & 1023; b.q[i] = b.p[j + 4096] + 1.0f; }
__global__ void k_lds(Big b) {
    __shared__ float s[256];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    s[threadIdx.x] = b.p[i];
    __syncthreads();
    float v = s[(threadIdx.x + 7) & 255];
    __syncthreads();
    s[threadIdx.x] = v * 2.f;
    __syncthreads();
    b.q[i] = s[(threadIdx.x + 3) & 255];
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <typename F> int run(const char* name, F launch, hipStream_t st) {
    hipGraph_t g; hipGraphExec_t ge; hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 40; ++i) launch(i);
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < 50; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s %.2f us/kernel\n", name, ms * 1e3 / 2000);
    return 0;
}
int main() {
    float *p, *q; CK(hipMalloc(&p, 1 << 24)); CK(hipMemset(p, 0, 1 << 24)); CK(hipMalloc(&q, 1 << 24)); CK(hipMemset(q, 0, 1 << 24));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    Big b{p, q, 1 << 20, {0}};
    for (int wgs : {72, 384}) {
        printf("-- %d workgroups x 256 threads, ping-pong buffers so each kernel depends on the previous one's stores\n", wgs);
        run("nothing", [&](int i) { hipLaunchKernelGGL(k_nothing, dim3(wgs), dim3(256), 0, st, b); }, st);
        run("store only", [&](int i) { hipLaunchKernelGGL(k_store, dim3(wgs), dim3(256), 0, st, b); }, st);
        run("load -> store", [&](int i) { Big c = b; if (i & 1) { c.p = q; c.q = p; } hipLaunchKernelGGL(k_load_store, dim3(wgs), dim3(256), 0, st, c); }, st);
        run("load -> load -> store", [&](int i) { Big c = b; if (i & 1) { c.p = q; c.q = p; } hipLaunchKernelGGL(k_dep2, dim3(wgs), dim3(256), 0, st, c); }, st);
        run("load -> 3 barriers via LDS -> store", [&](int i) { Big c = b; if (i & 1) { c.p = q; c.q = p; } hipLaunchKernelGGL(k_lds, dim3(wgs), dim3(256), 0, st, c); }, st);
    }
    return 0;
}
