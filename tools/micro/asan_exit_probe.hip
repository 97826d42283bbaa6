// Bare HIP program for tools/asan_host_check.sh: does the sanitizer's exit-time CHECK ("dev_runtime_unloaded_", reached through
// __cxa_finalize -> libamdhip64 -> libhsa-runtime64 -> operator delete) appear WITHOUT this repository's library?
//   mode 0: hipMalloc / kernel / hipFree
//   mode 1: the same + a kernel with a raised dynamic-LDS attribute (hipFuncSetAttribute, 128 KB)
//   mode 2: mode 1 + what the Whisper-tiny run adds on the host side: a 151 MB std::vector (the weight image) allocated and
//           freed, a non-blocking stream, a captured + instantiated + replayed graph, pinned host memory — all released
//           before main returns
//   mode 3: mode 0 + 1.8 GB of device memory in six buffers allocated and freed (the Whisper-tiny state arena is ~1.3 GB): under
//           host ASan every hipMalloc is a chunk of the sanitizer's DEVICE allocator and hipFree parks it in the quarantine; what
//           exceeds the quarantine budget (256 MB) is recycled — the last of it when the exiting thread commits its cache
//           (AsanThread::Destroy -> CommitBack -> Recycle -> DeviceAllocator::Deallocate), after the HSA runtime has unloaded
//   mode 4: mode 0 + sixty 16 MB device buffers allocated and freed.  Unlike mode 3's 300 MB chunks (each larger than the whole
//           quarantine budget, so recycled at once, while the runtime is alive) these STAY in the quarantine: at process exit it
//           holds ~256 MB of freed DEVICE chunks, as after wm_model_free of a Whisper-tiny model (about 150 buffers of 1 KB-150 MB)
// Prints "Done." before returning, like examples/main.cpp.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void fill(float* p, int n, float v) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void fill_lds(float* p, int n) {
    extern __shared__ float s[];
    s[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = s[(threadIdx.x + 1) % blockDim.x];
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const int n = 1 << 24;
    float* d = nullptr;
    CK(hipMalloc((void**)&d, (size_t)n * 4));
    hipLaunchKernelGGL(fill, dim3(n / 256), dim3(256), 0, nullptr, d, n, 1.0f);
    if (mode >= 1) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fill_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        hipLaunchKernelGGL(fill_lds, dim3(n / 256), dim3(256), 128 * 1024, nullptr, d, n);
    }
    CK(hipDeviceSynchronize());
    if (mode >= 2) {
        {
            std::vector<float> w(37760640, 0.5f);  // the size of whisper_tiny_weights.bin
            CK(hipMemcpy(d, w.data(), (size_t)n * 4, hipMemcpyHostToDevice));
        }
        hipStream_t st;
        CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        void* pinned = nullptr;
        CK(hipHostMalloc(&pinned, 4096, 0));
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(fill, dim3(n / 256), dim3(256), 0, st, d, n, 2.0f);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphDestroy(g));
        for (int i = 0; i < 8; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        CK(hipGraphExecDestroy(ge));
        CK(hipHostFree(pinned));
        CK(hipStreamDestroy(st));
    }
    if (mode >= 3) {
        void* big[6];
        for (auto& b : big) CK(hipMalloc(&b, (size_t)300 << 20));
        for (auto& b : big) CK(hipMemsetAsync(b, 0, (size_t)300 << 20, nullptr));
        CK(hipDeviceSynchronize());
        for (auto& b : big) CK(hipFree(b));
    }
    if (mode >= 4) {
        std::vector<void*> bufs(60);
        for (auto& b : bufs) CK(hipMalloc(&b, (size_t)16 << 20));
        for (auto& b : bufs) CK(hipMemsetAsync(b, 0, (size_t)16 << 20, nullptr));
        CK(hipDeviceSynchronize());
        for (auto& b : bufs) CK(hipFree(b));
    }
    CK(hipFree(d));
    printf("Done.\n");
    fflush(stdout);
    return 0;
}
