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
//   mode 5: the library's HIP usage pattern without the library: two non-blocking streams, 150 device buffers of 1 KB-150 MB
//           (1.3 GB), pinned host memory, an H2D copy from pageable memory on a stream, three instantiated graphs of 38 kernel
//           nodes with 200-byte kernel arguments replayed 30 times, events — everything destroyed / freed before main returns
//   mode 6: mode 0 + 1 200 device buffers of 1 MB allocated and freed LAST.  The quarantine budget is 256 MB and a recycle drains
//           it to ~230 MB; with 1 MB chunks it then climbs back to within 1 MB of the budget, so the few MB the HIP / HSA teardown
//           itself deletes at exit tip it over and the recycle that follows meets freed DEVICE chunks after the runtime has unloaded
//           (coarse 16-300 MB chunks, modes 3-5, leave it up to a chunk below the budget: no recycle at exit).  wm_model_free ends
//           with ~150 buffers of 1 KB-2.4 MB: the same fine-grained tail
// Prints "Done." before returning, like examples/main.cpp.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void fill(float* p, int n, float v) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
struct BigArgs {
    float* p;
    int n;
    char pad[200];
};
__global__ void fill_args(BigArgs a) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.n) a.p[i] += 1.0f;
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
    if (mode >= 5) {
        hipStream_t st[2];
        for (auto& q : st) CK(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
        std::vector<void*> bufs;
        const size_t sizes[] = {1 << 10, 16 << 10, 1 << 20, 2400 << 10, 9 << 20, 18 << 20, 75 << 20, 150 << 20};
        size_t tot = 0;
        for (int i = 0; tot < ((size_t)1300 << 20) && i < 150; ++i) {
            void* b = nullptr;
            const size_t sz = sizes[i % 8];
            CK(hipMalloc(&b, sz));
            CK(hipMemsetAsync(b, 0, sz, st[0]));
            bufs.push_back(b);
            tot += sz;
        }
        void* pinned = nullptr;
        CK(hipHostMalloc(&pinned, 1 << 16, hipHostMallocMapped));
        std::vector<float> pageable(240000, 1.f);
        CK(hipMemcpyAsync(d, pageable.data(), pageable.size() * 4, hipMemcpyHostToDevice, st[1]));
        hipEvent_t ev;
        CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        hipGraphExec_t ge[3];
        hipGraph_t g;
        CK(hipStreamBeginCapture(st[1], hipStreamCaptureModeThreadLocal));
        for (int k = 0; k < 38; ++k) {
            BigArgs a{};
            a.p = d;
            a.n = 4096;
            hipLaunchKernelGGL(fill_args, dim3(16), dim3(256), 0, st[1], a);
        }
        CK(hipStreamEndCapture(st[1], &g));
        for (auto& e : ge) CK(hipGraphInstantiate(&e, g, nullptr, nullptr, 0));
        CK(hipGraphDestroy(g));
        for (int i = 0; i < 30; ++i) CK(hipGraphLaunch(ge[i % 3], st[1]));
        CK(hipEventRecord(ev, st[1]));
        CK(hipEventSynchronize(ev));
        CK(hipStreamSynchronize(st[0]));
        for (auto& e : ge) CK(hipGraphExecDestroy(e));
        CK(hipEventDestroy(ev));
        CK(hipHostFree(pinned));
        for (void* b : bufs) CK(hipFree(b));
        for (auto& q : st) CK(hipStreamDestroy(q));
    }
    if (mode >= 6) {
        std::vector<void*> bufs(1200);
        for (auto& b : bufs) CK(hipMalloc(&b, (size_t)1 << 20));
        CK(hipDeviceSynchronize());
        for (auto& b : bufs) CK(hipFree(b));
    }
    CK(hipFree(d));
    printf("Done.\n");
    fflush(stdout);
    return 0;
}
