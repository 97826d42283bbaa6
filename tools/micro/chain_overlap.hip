// chain_overlap.hip — what does a dependent chain of small latency-bound kernels cost per link on gfx950, and can links overlap?
//   hipcc --offload-arch=gfx950 -O3 -o chain_overlap chain_overlap.hip && ./chain_overlap
// A "phase" = 96 workgroups x 256 threads: load 16 B of private weights per lane (independent of the previous phase), take
// the previous phase's 24 KB activation, add 1, publish it.  Modes:
//   A  plain launches, one stream, graph replay                       (today's decode step)
//   B  A + the in-kernel hand-off protocol (never has to wait: measures the protocol's own cost)
//   C  eager hipExtLaunchKernelGGL(hipExtAnyOrderLaunch) + protocol   (barrier bit off, if the runtime honours it here)
//   D  eager plain launches + protocol                                (baseline for C)
//   E  two alternating streams captured as one graph + protocol       (links k and k+1 on different queues)
// Hand-off: sc1 (write-through) stores -> every wave s_waitcnt vmcnt(0) -> barrier -> one relaxed agent atomic add;
// consumer: one lane polls relaxed (bounded), one agent acquire, barrier, plain loads.  The final activation must equal
// the chain length everywhere, or a stale read happened.
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int G = 96, T = 256, NPH = 32, XN = 6144;  // 24 KB activation

__global__ __launch_bounds__(256) void phase_kernel(const float* __restrict__ w, const float* xin, float* xout, unsigned* cnt_in,
                                                    unsigned* cnt_out, unsigned* tmo, int proto, float* sink) {
    __shared__ int s_ok;
    const f32x4 wv = *reinterpret_cast<const f32x4*>(w + ((size_t)blockIdx.x * T + threadIdx.x) * 4);  // does not depend on the chain
    if (proto && cnt_in) {
        if (threadIdx.x == 0) {
            unsigned spins = 0;
            int ok = 1;
            while (__hip_atomic_load(cnt_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)G) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 22)) {
                    ok = 0;
                    atomicAdd(tmo, 1u);
                    break;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            s_ok = ok;
        }
        __syncthreads();
    }
    float acc = wv[0] + wv[1] + wv[2] + wv[3];
    // every workgroup reads the whole activation (as a decode linear reads its 16 rows), writes its 64-float slice
    float v = 0.f;
    for (int i = threadIdx.x; i < XN; i += T) v += xin[i];
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    const float out = xin[j] + 1.0f;
    if (threadIdx.x < 64) {
        if (proto)
            __hip_atomic_store(xout + j, out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // sc1 store
        else
            xout[j] = out;
    }
    if (acc + v == 12345.678f) sink[0] = acc;  // keep the loads
    if (proto) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__global__ void reset_kernel(unsigned* cnt, float* x0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < NPH + 1) cnt[i * 32] = 0;
    if (i < XN) x0[i] = 0.f;
}

struct Bufs {
    float *w, *x[2], *sink;
    unsigned *cnt, *tmo;
};

static void launch_chain(const Bufs& b, hipStream_t s0, hipStream_t s1, int proto, int anyorder, hipEvent_t fork = nullptr) {
    hipLaunchKernelGGL(reset_kernel, dim3((XN + 255) / 256), dim3(256), 0, s0, b.cnt, b.x[0]);
    if (fork) {  // the second stream joins AFTER the reset
        CK(hipEventRecord(fork, s0));
        CK(hipStreamWaitEvent(s1, fork, 0));
    }
    for (int i = 0; i < NPH; ++i) {
        hipStream_t st = (s1 && (i & 1)) ? s1 : s0;
        const float* w = b.w + (size_t)i * G * T * 4;
        unsigned* cin = i ? b.cnt + (size_t)(i - 1) * 32 : nullptr;
        unsigned* cout = b.cnt + (size_t)i * 32;
        if (anyorder && i > 0)  // link 0 keeps its barrier bit: it (and so every later link) starts after the reset has completed
            hipExtLaunchKernelGGL(phase_kernel, dim3(G), dim3(T), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, w, (const float*)b.x[i & 1],
                                  b.x[(i + 1) & 1], cin, cout, b.tmo, proto, b.sink);
        else
            hipLaunchKernelGGL(phase_kernel, dim3(G), dim3(T), 0, st, w, (const float*)b.x[i & 1], b.x[(i + 1) & 1], cin, cout, b.tmo, proto,
                               b.sink);
    }
}

static bool check(const Bufs& b, const char* tag) {
    std::vector<float> h(XN);
    unsigned tmo = 0;
    CK(hipMemcpy(h.data(), b.x[NPH & 1], XN * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&tmo, b.tmo, 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < XN; ++i) bad += h[i] != (float)NPH;
    if (bad || tmo) printf("  [%s] WRONG: %d of %d values stale, %u spin time-outs\n", tag, bad, XN, tmo);
    return !bad && !tmo;
}

int main() {
    Bufs b;
    CK(hipMalloc(&b.w, (size_t)NPH * G * T * 16));
    CK(hipMemset(b.w, 0, (size_t)NPH * G * T * 16));
    CK(hipMalloc(&b.x[0], XN * 4));
    CK(hipMalloc(&b.x[1], XN * 4));
    CK(hipMalloc(&b.sink, 64));
    CK(hipMalloc(&b.cnt, (NPH + 1) * 32 * 4));
    CK(hipMalloc(&b.tmo, 4));
    CK(hipMemset(b.tmo, 0, 4));
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t e0, e1, ef, ej;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    const int REPS = 50;
    auto time_graph = [&](const char* tag, int proto, bool two) {
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
        launch_chain(b, s0, two ? s1 : nullptr, proto, 0, two ? ef : nullptr);
        if (two) {
            CK(hipEventRecord(ej, s1));
            CK(hipStreamWaitEvent(s0, ej, 0));
        }
        CK(hipStreamEndCapture(s0, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s0));
        CK(hipStreamSynchronize(s0));
        const bool ok = check(b, tag);
        CK(hipEventRecord(e0, s0));
        for (int r = 0; r < REPS; ++r) CK(hipGraphLaunch(ge, s0));
        CK(hipEventRecord(e1, s0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const bool ok2 = check(b, tag);
        printf("%-58s %7.2f us per link %s\n", tag, ms * 1000.f / (REPS * NPH), ok && ok2 ? "" : "(RESULT WRONG)");
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    };
    auto time_eager = [&](const char* tag, int proto, int anyorder) {
        launch_chain(b, s0, nullptr, proto, anyorder);
        CK(hipStreamSynchronize(s0));
        const bool ok = check(b, tag);
        CK(hipEventRecord(e0, s0));
        for (int r = 0; r < REPS; ++r) launch_chain(b, s0, nullptr, proto, anyorder);
        CK(hipEventRecord(e1, s0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const bool ok2 = check(b, tag);
        printf("%-58s %7.2f us per link %s\n", tag, ms * 1000.f / (REPS * NPH), ok && ok2 ? "" : "(RESULT WRONG)");
    };
    time_graph("A graph, one stream, plain", 0, false);
    time_graph("B graph, one stream, + hand-off protocol", 1, false);
    time_eager("D eager, one stream, + protocol", 1, 0);
    time_eager("C eager, hipExtAnyOrderLaunch, + protocol", 1, 1);
    time_graph("E graph, two alternating streams, + protocol", 1, true);
    time_graph("A graph, one stream, plain (again)", 0, false);
    return 0;
}
