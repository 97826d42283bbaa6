// kernels_frontend.hip — log-mel front end on the GPU: 16 kHz PCM -> [n_mels, n_frames] log-mel,
// the step immediately before the hot path.  The reference delegates it to HF's WhisperProcessor
// (export_weights.py:100-116).
//   frames_kernel      reflect-padded, Hann-windowed frames [rows, 416] fp32 (400 samples + 16 zero columns)
//   gemm_nt<float>     real DFT as an exact-fp32 MFMA GEMM against a [512, 416] cos / -sin basis (re at k, im at 256+k)
//   mel_log_kernel     |X|^2 -> sparse triangular mel filters -> log10(max(., 1e-10))
//   mel_norm_kernel    per-utterance max, clamp to max-8, (x+4)/4, channel-major output (what the encoder consumes)
#include "wm_kernels.h"

namespace wm {

__global__ __launch_bounds__(128) void frames_kernel(const float* __restrict__ pcm, float* __restrict__ F,
                                                     const float* __restrict__ window, int N, int n_frames, int hop) {
    const int t = blockIdx.x, b = blockIdx.y;
    const float* x = pcm + (size_t)b * N;
    float* row = F + ((size_t)b * n_frames + t) * 416;
    for (int n = threadIdx.x; n < 416; n += 128) {
        float v = 0.f;
        if (n < 400) {
            int i = hop * t + n - 200;
            if (i < 0) i = -i;                    // numpy 'reflect': edge sample not repeated
            if (i >= N) i = 2 * (N - 1) - i;
            v = x[i] * window[n];
        }
        row[n] = v;
    }
}
// x[b][i] = 0 for i >= len[b] (zero padding of short clips after the one-shot strided upload)
__global__ void zero_tails_kernel(float* __restrict__ pcm, const int* __restrict__ len, int N) {
    float* x = pcm + (size_t)blockIdx.y * N;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N && i >= len[blockIdx.y]) x[i] = 0.f;
}
void launch_zero_tails(float* pcm, const int* len, int B, int N, hipStream_t st) {
    hipLaunchKernelGGL(zero_tails_kernel, dim3((N + 255) / 256, B), dim3(256), 0, st, pcm, len, N);
}
void launch_frames(const float* pcm, float* F, const float* window, int B, int N, int n_frames, int hop, hipStream_t st) {
    hipLaunchKernelGGL(frames_kernel, dim3(n_frames, B), dim3(128), 0, st, pcm, F, window, N, n_frames, hop);
}

// spec [rows][512] (re at k, im at 256+k) -> logmel[b][m][t].  One workgroup = 32 frames; power spectrum in LDS; thread
// (f, m-group) walks only the non-zero band [lo[m], hi[m]) of each triangular filter.
__global__ __launch_bounds__(256) void mel_log_kernel(const float* __restrict__ spec, const float* __restrict__ fb /*[201][n_mels]*/,
                                                      const int* __restrict__ band /*[n_mels][2]*/, float* __restrict__ logmel,
                                                      int n_frames, int n_mels) {
    __shared__ float P[32][204];
    const int b = blockIdx.y, t0 = blockIdx.x * 32;
    for (int i = threadIdx.x; i < 32 * 201; i += 256) {
        const int f = i / 201, k = i % 201;
        float v = 0.f;
        if (t0 + f < n_frames) {
            const float* r = spec + ((size_t)b * n_frames + t0 + f) * 512;
            const float re = r[k], im = r[256 + k];
            v = re * re + im * im;
        }
        P[f][k] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * n_mels; i += 256) {
        const int m = i / 32, f = i % 32;  // consecutive threads -> consecutive frames: coalesced stores along t
        if (t0 + f >= n_frames) continue;
        const int lo = band[2 * m], hi = band[2 * m + 1];
        float s = 0.f;
        for (int k = lo; k < hi; ++k) s += fb[k * n_mels + m] * P[f][k];
        logmel[((size_t)b * n_mels + m) * n_frames + t0 + f] = log10f(fmaxf(s, 1e-10f));
    }
}
void launch_mel_log(const float* spec, const float* fb, const int* band, float* logmel, int B, int n_frames, int n_mels, hipStream_t st) {
    hipLaunchKernelGGL(mel_log_kernel, dim3((n_frames + 31) / 32, B), dim3(256), 0, st, spec, fb, band, logmel, n_frames, n_mels);
}

__global__ __launch_bounds__(1024) void mel_norm_kernel(const float* __restrict__ logmel, float* __restrict__ out, int n) {
    __shared__ float s_m[16];
    const float* x = logmel + (size_t)blockIdx.x * n;
    float* y = out + (size_t)blockIdx.x * n;
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < n; i += 1024) mx = fmaxf(mx, x[i]);
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = s_m[0];
    for (int k = 1; k < 16; ++k) mx = fmaxf(mx, s_m[k]);
    const float floor_v = mx - 8.0f;
    for (int i = threadIdx.x; i < n; i += 1024) y[i] = (fmaxf(x[i], floor_v) + 4.0f) / 4.0f;
}
void launch_mel_norm(const float* logmel, float* out, int B, int n, hipStream_t st) {
    hipLaunchKernelGGL(mel_norm_kernel, dim3(B), dim3(1024), 0, st, logmel, out, n);
}

}  // namespace wm
