// kernels_encoder.hip — the dense (MFMA-bound) half of the path: conv stem as implicit GEMM, the 1500-row
// projections / MLP, fused flash-style encoder attention, LayerNorm rows.  gfx950 only.
//
// Replaces (reference file:line): conv1d whisper_tensor.mojo:367-428 (+gelu :288-308, +pos add whisper.mojo:83-89),
// matmul_384x384/_384x1536/_1536x384 whisper_tensor.mojo:74-122, matmul_Q_K/_S_V + softmax + head gather/scatter
// layers.mojo:273-342, layer_norm whisper_tensor.mojo:249-285, residual adds layers.mojo:457-461,483-487,513-517.
#include "wm_kernels.h"

#include <algorithm>
#include <type_traits>
#include <cstdio>
#include <cstdlib>

namespace wm {

// ---- launcher status (wm_kernels.h) ------------------------------------------------------------------------------
static thread_local const char* g_refusal = "";
static thread_local char g_refusal_buf[256];
const char* launch_last_refusal() { return g_refusal; }
int launch_refuse(const char* why) {
    g_refusal = why;
    return WM_LAUNCH_BAD_SHAPE;
}
int launch_hip_failed(const char* what, hipError_t e) {
    snprintf(g_refusal_buf, sizeof g_refusal_buf, "%s: %s", what, hipGetErrorString(e));
    g_refusal = g_refusal_buf;
    return WM_LAUNCH_HIP;
}

// ------------------------------------------------------------------------------------------------------------
// mel [B][C][L] fp32 (channel-major, sample_input.bin layout) -> token-major, zero-padded [B][L+2][Cp] T.
// Row t+1 holds frame t; rows 0 and L+1 are the conv zero padding (the reference skips out-of-range taps,
// whisper_tensor.mojo:405,422); channels >= C are zero so the implicit-GEMM K is a multiple of 32.
// Replaces the input transpose of conv1d (whisper_tensor.mojo:383-388).
template <typename T>
__global__ __launch_bounds__(256) void mel_transpose_pad_kernel(const float* __restrict__ mel, T* __restrict__ out,
                                                                int C, int L, int Cp) {
    __shared__ float tile[64][65];
    const int b = blockIdx.z, t0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const float* src = mel + (size_t)b * C * L;
    T* dst = out + (size_t)b * (L + 2) * Cp;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        int c = i >> 6, t = i & 63;
        float v = 0.f;
        if (c0 + c < C && t0 + t < L) v = src[(size_t)(c0 + c) * L + t0 + t];
        tile[c][t] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        int t = i >> 6, c = i & 63;
        if (c0 + c < Cp && t0 + t < L) dst[(size_t)(t0 + t + 1) * Cp + c0 + c] = from_f32<T>(tile[c][t]);
    }
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        for (int c = threadIdx.x; c < Cp; c += 256) {
            dst[c] = from_f32<T>(0.f);
            dst[(size_t)(L + 1) * Cp + c] = from_f32<T>(0.f);
        }
    }
}

template <typename T>
void launch_mel_transpose_pad(const float* mel, void* out, int B, int C, int L, int Cp, hipStream_t st) {
    dim3 grid((L + 63) / 64, (Cp + 63) / 64, B);
    hipLaunchKernelGGL(mel_transpose_pad_kernel<T>, grid, dim3(256), 0, st, mel, (T*)out, C, L, Cp);
}

// ------------------------------------------------------------------------------------------------------------
// Shared GEMM epilogue.  acc[j][i][r] = C[m0 + 16i + r16][n0 + 16j + 4g + r] for a wave's 64x64 sub-tile.
// out = act(acc + bias) + pos + residual.  Every operand the epilogue reads (bias, positional rows, residual rows) is
// requested up front, before anything is waited for: 16 dependent load->wait->store rounds per wave were the longest
// phase of a K=384 tile.  A 64-wide wave tile never straddles a column group (group_n = d_model = 64·heads).
template <typename TO, bool FASTG>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x4 (&acc)[4][4], int m0, int n0, int bz, int r16, int g) {
    TO* Cb = (TO*)p.C + (size_t)bz * p.strideC;
    int nc = n0 + g * 4;  // column inside the group
    if (p.group_n > 0) {
        const int grp = n0 / p.group_n;
        Cb += (size_t)grp * p.group_stride;
        nc -= grp * p.group_n;
    }
    f32x4 bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = *reinterpret_cast<const f32x4*>(p.bias + n0 + j * 16 + g * 4);
    }
    // two halves of 32 rows: 8 addend fragments in flight per half keeps the kernel near 200 VGPRs, which leaves room on
    // a SIMD for a latency-bound wave of another queue (a decode step beside the encoder)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        bool valid[2];
        f32x4 ex[2][4];  // post-activation addend: pos + residual
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            valid[ii] = m0 + (hf * 2 + ii) * 16 + r16 < p.M;
#pragma unroll
            for (int j = 0; j < 4; ++j) ex[ii][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (p.residual) {
            const float* Rb = p.residual + (size_t)bz * p.strideR + n0 + g * 4;
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
                if (valid[ii]) {
                    const float* rr = Rb + (size_t)(m0 + (hf * 2 + ii) * 16 + r16) * p.ldr;
#pragma unroll
                    for (int j = 0; j < 4; ++j) ex[ii][j] = *reinterpret_cast<const f32x4*>(rr + j * 16);
                }
        }
        if (p.pos) {
            const float* Pb = p.pos + n0 + g * 4;
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
                if (valid[ii]) {
                    const float* pr = Pb + (size_t)(m0 + (hf * 2 + ii) * 16 + r16) * p.N;
#pragma unroll
                    for (int j = 0; j < 4; ++j) ex[ii][j] += *reinterpret_cast<const f32x4*>(pr + j * 16);
                }
        }
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            if (!valid[ii]) continue;
            const int i = hf * 2 + ii;
            TO* crow = Cb + (size_t)(m0 + i * 16 + r16) * p.ldc + nc;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v = acc[j][i] + bv[j];
                if (p.act) {
                    if constexpr (FASTG) {
                        v = gelu_fast4(v, p.gelu_mode);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = gelu_f(v[r], p.gelu_mode);
                    }
                }
                v += ex[ii][j];
                if constexpr (sizeof(TO) == 4) {
                    *reinterpret_cast<f32x4*>((float*)crow + j * 16) = v;
                } else {
                    typedef __attribute__((ext_vector_type(4))) TO to4;
                    to4 o = {from_f32<TO>(v[0]), from_f32<TO>(v[1]), from_f32<TO>(v[2]), from_f32<TO>(v[3])};
                    *reinterpret_cast<to4*>(crow + j * 16) = o;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// C[M,N] = A[M,K]·W[N,K]ᵀ with a fused epilogue.  Both operands are K-contiguous ("NT"), which is what the
// reference's HF-layout weights give (whisper_tensor.mojo:151: B is [out,in]) and what MFMA fragments want.
// 128x128 tile / 256 threads; wave (wm,wn) owns 64x64 = 4x4 MFMA 16x16 accumulators; K step 32.
// Computed as Cᵀ tiles (A-operand = W fragment) so that each lane ends up with 4 CONSECUTIVE output columns of one
// row: 16-byte epilogue loads/stores.
// An implicit-GEMM conv is the same kernel with overlapping A rows (lda = stride*C_in over the padded token-major
// input, K = 3*C_in): row t of A is the contiguous window x[t*stride-1 .. t*stride+1][:].
template <typename T, typename TO>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmParams p) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int r16 = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.y * 128 + wm * 64, n0 = blockIdx.x * 128 + wn * 64;
    const T* A = (const T*)p.A + (size_t)blockIdx.z * p.strideA;
    const T* W = (const T*)p.W;
    const T* ap[4];
    const T* wp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ap[i] = A + (size_t)(m0 + i * 16 + r16) * p.lda + g * 8;
        wp[i] = W + (size_t)(n0 + i * 16 + r16) * p.ldw + g * 8;
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int k0 = 0; k0 < p.K; k0 += 32) {
        Frag<T> a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = load_frag<T>(ap[i] + k0);
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = load_frag<T>(wp[j] + k0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = mma32(b[j], a[i], acc[j][i]);
    }

    gemm_epilogue<TO, false>(p, acc, m0, n0, blockIdx.z, r16, g);
}

// ------------------------------------------------------------------------------------------------------------
// Same contract, 16-bit operands, LDS-staged (the dense path of BASELINE configs 3-5).
// 128x128 tile, BK = 64, 256 threads (2x2 waves of 64x64).  Both operand tiles go global -> LDS with
// global_load_lds_dwordx4 (no VGPR round trip; one wave instruction = 8 rows x 128 B = 1 KiB of lane-linear LDS) into a
// double buffer; the next K tile's DMA is in flight while this one's fragments are read (ds_read_b128) and multiplied.
// LDS rows are 128 B, so the 16-byte chunk index is XOR-swizzled with (row & 7) — applied on the per-lane GLOBAL source
// address (the LDS side of a DMA cannot scatter) and again on the fragment read address.
template <typename T>
__device__ __forceinline__ void stage_tile(const T* __restrict__ gbase, long ld, T* lds_tile, int lane, int w) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = w * 4 + j;                       // 1-KiB piece index: rows 8i .. 8i+7
        const int row = 8 * i + (lane >> 3), cp = lane & 7;
        const int c = cp ^ (row & 7);                  // logical 16-byte chunk that lives at position cp
        const T* g = gbase + (size_t)row * ld + c * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(lds_tile + i * 512), 16, 0, 0);
    }
}

template <typename T, typename TO>
__global__ __launch_bounds__(256) void gemm_nt_lds_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(16))) T lds[2][2][128 * 64];  // [buffer][A|W][row][64]
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int r16 = lane & 15, g = lane >> 4;
    // XCD-aware tile order: the hardware deals workgroups round-robin to the 8 XCDs (private L2 each); give XCD x a
    // contiguous run of logical tiles (column tiles of one row tile adjacent) so an A tile is fetched into one L2 once.
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (p.xcd_remap) {
        const int G = gridDim.x * gridDim.y * gridDim.z;
        const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const int x = lin & 7, q = lin >> 3, per = G >> 3, rem = G & 7;
        const int t = x * per + (x < rem ? x : rem) + q;
        bx = t % gridDim.x;
        const int u = t / gridDim.x;
        by = u % gridDim.y;
        bz = u / gridDim.y;
    }
    const int bm = by * 128, bn = bx * 128;
    const T* A = (const T*)p.A + (size_t)bz * p.strideA + (size_t)bm * p.lda;
    const T* W = (const T*)p.W + (size_t)bn * p.ldw;
    f32x4 acc[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nt = p.K >> 6;
    stage_tile<T>(A, p.lda, lds[0][0], lane, wid);
    stage_tile<T>(W, p.ldw, lds[0][1], lane, wid);
    __syncthreads();  // drains the DMA (vmcnt(0)) and publishes the tile
    int cur = 0;
    for (int t = 0; t < nt; ++t) {
        if (t + 1 < nt) {
            stage_tile<T>(A + (t + 1) * 64, p.lda, lds[cur ^ 1][0], lane, wid);
            stage_tile<T>(W + (t + 1) * 64, p.ldw, lds[cur ^ 1][1], lane, wid);
        }
        const T* As = lds[cur][0];
        const T* Ws = lds[cur][1];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            Frag<T> a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ra = wm * 64 + i * 16 + r16;
                a[i] = load_frag<T>(As + ra * 64 + (((ks * 4 + g) ^ (ra & 7)) << 3));
                const int rb = wn * 64 + i * 16 + r16;
                b[i] = load_frag<T>(Ws + rb * 64 + (((ks * 4 + g) ^ (rb & 7)) << 3));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[j][i] = mma32(b[j], a[i], acc[j][i]);
        }
        __syncthreads();  // next tile landed; everyone is done with this one
        cur ^= 1;
    }

    gemm_epilogue<TO, true>(p, acc, bm + wm * 64, bn + wn * 64, bz, r16, g);
}

// ------------------------------------------------------------------------------------------------------------
// Row-panel GEMM for K <= 512 (QKV, fc1, cross-K/V: K = d_model): the A operand is STATIONARY IN REGISTERS.
// The 128x128-tile kernel above stages 192 KB per K = 384 tile, half of it the activation panel that comes from HBM / the
// Infinity Cache, and is bound by that staging (8 TB/s aggregate at 540 TFLOP/s).  Here a wave owns 32 rows of a 128-row
// panel and keeps their whole K extent as MFMA fragments (2 x K/32 fragments = 96 VGPRs at K = 384), loaded once per panel;
// only W tiles stream through LDS — L2-resident (<= 2.4 MB), 96 KB per 128x128 output tile — in 16 KB stages of 64 k
// through a 4-slot ring filled by global_load_lds three stages ahead, across column tiles and panels (counted vmcnt, raw
// s_barrier: the ring never drains).  A workgroup walks a contiguous range of (panel, column tile) units, column tile
// fastest, so a panel's A fragments are fetched by one or two workgroups.  Bias is folded into the accumulator
// initialisation from an LDS copy (no VMEM loads inside the stream).  No residual / positional addends (launcher).
template <typename T, typename TO, int KS /* K / 32 */, bool LNA = false /* A = fp32 rows, LayerNorm applied while loading */,
          int NW = 4 /* waves: 4 = 128-row panels, two workgroups per CU; 8 = 256-row panels, one per CU — every W stage then serves
                        twice the rows: half the L2 -> LDS traffic and half the DMA instructions per MFMA */>
__global__ __launch_bounds__(NW * 64, 2) void gemm_nt_rowpanel_kernel(GemmParams p, int n_units) {
    constexpr int SPU = KS / 2;                 // 64-k stages per unit
    constexpr int NSLOT = NW == 4 ? 4 : 8;      // ring slots of 128 W rows x 64 k (16 KB)
    constexpr int AHEAD = NW == 4 ? 3 : 5;      // stages requested ahead of the one being multiplied
    constexpr int PR = NW * 32;                 // rows per panel
    constexpr int PPW = 16 / NW;                // one-KiB DMA pieces per stage and wave
    constexpr int SLOT = 128 * 64;              // elements per slot
    extern __shared__ __attribute__((aligned(16))) unsigned char rp_smem[];  // ONE LDS object: ring, then fp32 bias [N]
    T* ring = reinterpret_cast<T*>(rp_smem);
    float* s_bias = reinterpret_cast<float*>(rp_smem + (size_t)NSLOT * SLOT * sizeof(T));
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const int nct = p.N >> 7;
    const int u0 = (int)((long)blockIdx.x * n_units / gridDim.x), u1 = (int)((long)(blockIdx.x + 1) * n_units / gridDim.x);
    const int n_stage = (u1 - u0) * SPU;
    const T* Wg = (const T*)p.W;
    for (int i = threadIdx.x; i < p.N; i += NW * 64) s_bias[i] = p.bias ? p.bias[i] : 0.f;
    float* s_gb = s_bias + p.N;  // LNA: gamma [K], beta [K]
    if constexpr (LNA) {
        for (int i = threadIdx.x; i < KS * 32; i += NW * 64) {
            s_gb[i] = p.ln_g[i];
            s_gb[KS * 32 + i] = p.ln_b[i];
        }
    }
#ifdef WM_DEV
    // developer: 100 MHz wall-clock stamps of wave 0, first three units, kept in LDS (a global store would enter the counted VMEM
    // queue) and dumped at the end: per stage [top, past the barrier, DMAs issued, MFMAs issued], per unit [epilogue start, end]
    long long* s_dbg = reinterpret_cast<long long*>(s_gb + (LNA ? 2 * KS * 32 : 0));
    const bool dbg_on = p.dbg != nullptr && w == 0 && lane == 0;
#define WM_RP_STAMP(UU, K) do { if (dbg_on && (UU) < 3) s_dbg[(UU) * 32 + (K)] = (long long)wall_clock64(); } while (0)
#else
#define WM_RP_STAMP(UU, K) do { } while (0)
#endif
    __syncthreads();  // before any DMA is in flight: a __syncthreads() later would drain the ring (vmcnt(0))

    // DMA source = (uniform base of the stage's piece) + (this lane's 32-bit byte offset, the same for every stage and piece): as
    // stage_tile, piece i = W rows 8i .. 8i+7 of the column tile, 16-byte chunk c of row r at position c ^ (r & 7) — and
    // (8 i + lane / 8) & 7 == lane / 8 whatever i.  No per-issue address arithmetic in vector registers.
    const unsigned woff = (unsigned)((((size_t)(8 * w * PPW + (lane >> 3))) * p.ldw + (((lane & 7) ^ (lane >> 3)) << 3)) * sizeof(T));
    auto issue = [&](int s) {  // stage s of this workgroup's stream -> ring slot s % NSLOT
        const int u = u0 + s / SPU, t = s % SPU;
        const int ct = u % nct;
        const char* gbase = reinterpret_cast<const char*>(Wg + (size_t)ct * 128 * p.ldw + t * 64);
        T* slot = ring + (s & (NSLOT - 1)) * SLOT + w * PPW * 512;
#pragma unroll
        for (int j = 0; j < PPW; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gbase + (size_t)j * 8 * p.ldw * sizeof(T) + woff),
                                             (__attribute__((address_space(3))) void*)(slot + j * 512), 16, 0, 0);
    };
#pragma unroll
    for (int s = 0; s < AHEAD; ++s)
        if (s < n_stage) issue(s);

    Frag<T> a[2][KS];
    f32x4 acc[8][2];
    int panel = -1;
    bool st_pend = false;  // the previous unit's 16 stores sit between the DMAs in this wave's VMEM queue
#define WM_RP_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)\n\ts_barrier" ::: "memory")
    for (int u = u0, s = 0; u < u1; ++u) {
        const int pn = u / nct, ct = u % nct;
        const bool newp = pn != panel;
        if (newp) {  // new panel: this wave's 32 rows, whole K, straight to registers (rows past M clamp to the last)
            panel = pn;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int row = pn * PR + w * 32 + i * 16 + r16;
                row = row < p.M ? row : p.M - 1;
                if constexpr (LNA) {
                    // The fp32 residual row instead of its normalised 16-bit copy: a lane holds 8 of every 32 columns (the four
                    // lanes r16, r16+16, +32, +48 share the row), so the statistics are a sum over the lane's KS*8 values and two
                    // butterfly steps; one-pass variance and operation order as layernorm_rows_kernel (whisper_tensor.mojo:249-285).
                    const float* xr = (const float*)p.A + (size_t)row * p.lda + g * 8;
                    f32x4 xv[KS][2];
#pragma unroll
                    for (int k = 0; k < KS; ++k) {
                        xv[k][0] = *reinterpret_cast<const f32x4*>(xr + k * 32);
                        xv[k][1] = *reinterpret_cast<const f32x4*>(xr + k * 32 + 4);
                    }
                    float sm = 0.f, sq = 0.f;
#pragma unroll
                    for (int k = 0; k < KS; ++k)
#pragma unroll
                        for (int h = 0; h < 2; ++h)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                sm += xv[k][h][e];
                                sq += xv[k][h][e] * xv[k][h][e];
                            }
                    sm += __shfl_xor(sm, 16, 64);
                    sq += __shfl_xor(sq, 16, 64);
                    sm += __shfl_xor(sm, 32, 64);
                    sq += __shfl_xor(sq, 32, 64);
                    const float mean = sm / (float)(KS * 32);
                    const float var = (sq / (float)(KS * 32)) - (mean * mean);
                    const float inv_std = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
                    for (int k = 0; k < KS; ++k) {
                        float y[8];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const f32x4 gm = *reinterpret_cast<const f32x4*>(&s_gb[k * 32 + g * 8 + h * 4]);
                            const f32x4 bt = *reinterpret_cast<const f32x4*>(&s_gb[KS * 32 + k * 32 + g * 8 + h * 4]);
#pragma unroll
                            for (int e = 0; e < 4; ++e) y[h * 4 + e] = (xv[k][h][e] - mean) * inv_std * gm[e] + bt[e];
                        }
                        a[i][k] = make_frag<T>(y);
                    }
                    // one half-panel at a time: with both halves' 2 x 96 fp32 values in flight together the kernel spills, and a
                    // reload inside the steady-state loop costs an s_waitcnt vmcnt(0) that drains the DMA ring
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    const T* ar = (const T*)p.A + (size_t)row * p.lda + g * 8;
#pragma unroll
                    for (int k = 0; k < KS; ++k) a[i][k] = load_frag<T>(ar + k * 32);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(&s_bias[ct * 128 + j * 16 + g * 4]);
            acc[j][0] = b4;
            acc[j][1] = b4;
        }
#pragma unroll
        for (int t = 0; t < SPU; ++t, ++s) {
            // Stage s must have landed.  vmcnt(N) = all but the N youngest VMEM operations of this wave are done, and
            // loads, stores and LDS-DMA count together in issue order: younger than stage s are the (<= 2) stages issued
            // after it, 4 DMAs each, plus — for the first three stages after an epilogue — that epilogue's 8 or 16 stores.
            // The barrier then also says everyone is done reading slot (s-1) % 4, which the next DMA overwrites.
            // Younger than stage s are min(AHEAD - 1, rem) stages of PPW DMAs each — 8 in the steady state for either shape —
            // plus, while the unit's first AHEAD stages are multiplied, the previous epilogue's stores.
            const int rem = n_stage - 1 - s;
            WM_RP_STAMP(u - u0, t * 4 + 0);
            if (t == 0 && newp) {  // fresh A fragments: the compiler waits vmcnt(0) for them anyway
                WM_RP_WAIT(0);
            } else if (rem >= AHEAD - 1) {
                if (t < AHEAD && st_pend) {
                    if constexpr (sizeof(TO) == 2)
                        WM_RP_WAIT(16);  // 8 epilogue stores
                    else
                        WM_RP_WAIT(24);  // 16 epilogue stores
                } else {
                    WM_RP_WAIT(8);
                }
            } else if constexpr (NW == 4) {
                if (t < AHEAD && st_pend) {
                    if constexpr (sizeof(TO) == 2) {
                        if (rem == 1)
                            WM_RP_WAIT(12);
                        else
                            WM_RP_WAIT(8);
                    } else {
                        if (rem == 1)
                            WM_RP_WAIT(20);
                        else
                            WM_RP_WAIT(16);
                    }
                } else {
                    if (rem == 1)
                        WM_RP_WAIT(4);
                    else
                        WM_RP_WAIT(0);
                }
            } else {  // the last three stages of the workgroup's whole stream: also waits for any stores (conservative)
                if (rem == 3)
                    WM_RP_WAIT(6);
                else if (rem == 2)
                    WM_RP_WAIT(4);
                else if (rem == 1)
                    WM_RP_WAIT(2);
                else
                    WM_RP_WAIT(0);
            }
            WM_RP_STAMP(u - u0, t * 4 + 1);
            if (s + AHEAD < n_stage) issue(s + AHEAD);
            WM_RP_STAMP(u - u0, t * 4 + 2);
            const T* Ws = ring + (s & (NSLOT - 1)) * SLOT;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                // all 8 W fragments of the k-step requested first (32 VGPRs), then the 16 MFMAs: left alone the compiler
                // reuses one 8-register pair and exposes the LDS latency in front of every four MFMAs
                Frag<T> bf[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int rb = j * 16 + r16;
                    bf[j] = load_frag<T>(Ws + rb * 64 + (((ks * 4 + g) ^ (rb & 7)) << 3));
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);  // 8 DS reads
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    acc[j][0] = mma32(bf[j], a[0][t * 2 + ks], acc[j][0]);
                    acc[j][1] = mma32(bf[j], a[1][t * 2 + ks], acc[j][1]);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);  // 16 MFMAs
            }
            WM_RP_STAMP(u - u0, t * 4 + 3);
        }
        WM_RP_STAMP(u - u0, 24);
        // epilogue: acc[j][i][r] = C[pn*128 + w*32 + 16i + r16][ct*128 + 16j + 4g + r]
        TO* Cb = (TO*)p.C;
        int ncol = ct * 128;
        if (p.group_n > 0) {
            const int grp = ncol / p.group_n;
            Cb += (size_t)grp * p.group_stride;
            ncol -= grp * p.group_n;
        }
        if constexpr (sizeof(TO) == 2) {
            // 16-bit output: a lane's 4 columns are 8 bytes, a wave store of them is 16 rows x 32 B — quarter lines, and
            // the three wide GEMMs ran at the resulting ~1.3 TB/s of output.  Transpose through LDS instead: the ring slot
            // just consumed is free until the next stage's DMA (issued after that stage's barrier); each wave takes 4 KB
            // of it = 16 rows x 256 B, written as [row][16-byte chunk ^ row] and read back 4 whole rows per instruction.
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // everyone is done reading this slot
            // (8 waves: 32 KB = the TWO slots behind the stream, (s-1) % 8 for waves 0-3 and (s-2) % 8 for waves 4-7 — with 5 stages
            // requested ahead of 8 slots both stay free until the tops of stages s+2 / s+1, past this epilogue's barrier-ordered end)
            unsigned char* scr = reinterpret_cast<unsigned char*>(ring + ((s - 1 - (w >> 2)) & (NSLOT - 1)) * SLOT) + (w & 3) * 4096;
            typedef __attribute__((ext_vector_type(4))) TO to4;
            typedef __attribute__((ext_vector_type(8))) TO to8;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    f32x4 v = acc[j][i];
                    if (p.act) {
                        v = gelu_fast4(v, p.gelu_mode);
                    }
                    const to4 o = {from_f32<TO>(v[0]), from_f32<TO>(v[1]), from_f32<TO>(v[2]), from_f32<TO>(v[3])};
                    const int chunk = (2 * j + (g >> 1)) ^ r16;
                    *reinterpret_cast<to4*>(scr + r16 * 256 + chunk * 16 + (g & 1) * 8) = o;
                }
                const int mbase = pn * PR + w * 32 + i * 16;
                if (mbase < p.M) {  // wave-uniform: 4 store instructions (rows past M masked per lane)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int row = 4 * k + (lane >> 4), cl = lane & 15;
                        const to8 v8 = *reinterpret_cast<const to8*>(scr + row * 256 + ((cl ^ row) * 16));
                        if (mbase + row < p.M) *reinterpret_cast<to8*>(Cb + (size_t)(mbase + row) * p.ldc + ncol + cl * 8) = v8;
                    }
                }
            }
            // exactly 8 store instructions per wave when all its 32 rows exist; otherwise (last panel) drain, don't count
            st_pend = __builtin_amdgcn_readfirstlane((int)(pn * PR + w * 32 + 32 <= p.M)) != 0;
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int m = pn * PR + w * 32 + i * 16 + r16;
                if (m >= p.M) continue;
                float* crow = (float*)Cb + (size_t)m * p.ldc + ncol + g * 4;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    f32x4 v = acc[j][i];
                    if (p.act) {
                        v = gelu_fast4(v, p.gelu_mode);
                    }
                    *reinterpret_cast<f32x4*>(crow + j * 16) = v;
                }
            }
            // exactly 16 store instructions per wave when all its 32 rows exist
            st_pend = __builtin_amdgcn_readfirstlane((int)(pn * PR + w * 32 + 32 <= p.M)) != 0;
        }
        if (!st_pend) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        WM_RP_STAMP(u - u0, 25);
    }
#undef WM_RP_WAIT
#ifdef WM_DEV
    if (dbg_on) {
        for (int i = 0; i < 96; ++i) p.dbg[(size_t)blockIdx.x * 96 + i] = s_dbg[i];
    }
#endif
#undef WM_RP_STAMP
}
template <typename T, typename TO, int KS, bool LNA, int NW> static int launch_rowpanel_t(const GemmParams& p, hipStream_t st) {
    const int n_units = ((p.M + NW * 32 - 1) / (NW * 32)) * (p.N / 128);
    static const int grid_env = wm_env("WM_RP_GRID") ? atoi(wm_env("WM_RP_GRID")) : 0;
    const int grid = std::min(n_units, grid_env > 0 ? grid_env : (NW == 4 ? 512 : 256));  // two 64-KB-ring workgroups per CU, or one of 128 KB
    size_t lds = (size_t)(NW == 4 ? 4 : 8) * 128 * 64 * sizeof(T) + (size_t)p.N * 4 + (LNA ? (size_t)2 * KS * 32 * 4 : 0);
#ifdef WM_DEV
    lds += 96 * 8;  // phase stamps
#endif
    if (const hipError_t e = ensure_dyn_lds<&gemm_nt_rowpanel_kernel<T, TO, KS, LNA, NW>>(NW == 4 ? 80 * 1024 : 144 * 1024); e != hipSuccess)
        return launch_hip_failed("row-panel GEMM: dynamic LDS attribute", e);
    hipLaunchKernelGGL((gemm_nt_rowpanel_kernel<T, TO, KS, LNA, NW>), dim3(grid), dim3(NW * 64), lds, st, p, n_units);
    return WM_LAUNCH_OK;
}
template <typename T, typename TO, int KS> static int launch_rowpanel(const GemmParams& p, hipStream_t st) {
    // 128-row panels, two workgroups per CU.  The 8-wave shape (256-row panels: half the W traffic and DMA instructions per MFMA)
    // measured the same on one box (encoder 4.24 vs 4.23 ms): W staging is not what bounds this kernel.  Developer A/B only.
#ifdef WM_DEV
    if (wm_env("WM_RP_NW8")) {
        if (p.ln_g) return launch_rowpanel_t<T, TO, KS, true, 8>(p, st);
        return launch_rowpanel_t<T, TO, KS, false, 8>(p, st);
    }
#endif
    if (p.ln_g) return launch_rowpanel_t<T, TO, KS, true, 4>(p, st);
    return launch_rowpanel_t<T, TO, KS, false, 4>(p, st);
}
static bool rowpanel_ok(int operand_bytes, const GemmParams& p, int batch) {
    // A-stationary row-panel kernel: plain [M,K]x[N,K] (no conv batching), K = 384 (K = 512 needs 128 A registers: spills), no addends, N <= 3072
    static const bool off = wm_env("WM_GEMM_NO_ROWPANEL") != nullptr || wm_env("WM_GEMM_DIRECT") != nullptr;
    return operand_bytes == 2 && !off && batch == 1 && !p.residual && !p.pos && p.K == 384 && p.N <= 3072 && (p.N & 127) == 0 &&
           (p.group_n == 0 || p.group_n % 128 == 0);
}
bool gemm_nt_fuses_layernorm(int operand_bytes, const GemmParams& p, int batch) {
    static const bool off = wm_env("WM_GEMM_NO_LN_FUSE") != nullptr;
    return !off && rowpanel_ok(operand_bytes, p, batch);
}

// ------------------------------------------------------------------------------------------------------------
// Full-row GEMM for the residual-stream writers (O-proj, fc2, conv2: N = 384, fp32 out): one workgroup owns 128 COMPLETE output
// rows.  (a) The A rows are staged once, not once per 128-column tile, and W once per 128 rows of ALL columns: 1.18 GB through
// L2 -> LDS for fc2 at 64 clips against 1.77 GB with 128x128 tiles — that traffic, not MFMA or HBM, is what bounds these GEMMs.
// (b) The epilogue holds whole rows: it can write the NEXT LayerNorm's 16-bit operand rows beside the fp32 residual stream
// (no LayerNorm launch, no second read of the 147 MB stream).
// 512 threads: wave (wr, wc) = rows 64 wr .. +64 (4 row blocks) x 6 column blocks of 16: blocks 2 wc, 2 wc + 1 of the first
// 128 columns and 4 wc .. 4 wc + 3 of the other 256 (96 accumulator registers; 20 fragment reads per 48 MFMAs).
// K advances 64 per pair of 32-KB HALF-STAGES of 256 rows x 128 B:
//   X(t) = [A rows 0..127 ; W rows 0..127] x k64(t)      Y(t) = [W rows 128..383] x k64(t)
// (whole 128-byte lines per row: with 64-byte rows every line travelled L2 -> L1 twice).  Half-stages go global -> LDS by
// global_load_lds into a 4-slot ring, three ahead, counted vmcnt + raw s_barrier; 16-byte chunk c of row r sits at position
// c ^ (r & 7) (as the 128x128 kernel).  Software pipeline with explicit fragment sets; reads and lgkmcnt waits are inline asm
// (hipcc's wait insertion cannot express "all but the N youngest reads" across the back edge and waits for everything):
//   step X(t): barrier; refill; request the 8 fragments of Y(t); wait until X(t)'s 12 are in; 16 MFMAs
//   step Y(t): barrier; refill; request the 12 fragments of X(t+1); wait until Y(t)'s 8 are in; 32 MFMAs (A fragments of X(t) kept)
// At every barrier each wave holds the previous half-stage in registers, so the slot before it is free for the refill.
template <typename T, bool LNO, bool ACT>
__global__ __launch_bounds__(512) void gemm_nt_fullrow_kernel(GemmParams p) {
    constexpr int NSLOT = 4, SLOT = 256 * 64;  // elements per ring slot (32 KB)
    extern __shared__ __attribute__((aligned(16))) unsigned char fr_smem[];
    T* ring = reinterpret_cast<T*>(fr_smem);
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r16 = lane & 15, g = lane >> 4;
    const int wr = w >> 2, wc = w & 3;
    const int bz = blockIdx.y, m0 = blockIdx.x * 128;
    auto colj = [&](int j) { return j < 2 ? (wc * 2 + j) * 16 : 128 + (wc * 4 + j - 2) * 16; };  // first column of block j
    f32x4 acc[6][4];
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // DMA sources as (uniform base) + (32-bit byte offset per lane).  32 one-KiB pieces per half-stage (8 rows x 128 B); wave w
    // stages pieces 4w .. 4w+3 = slot rows 32w .. 32w+31.  X: slot rows 0-127 are A rows (waves 0-3), 128-255 W rows 0-127
    // (waves 4-7);  Y: slot row r is W row 128 + r.
    const int l8 = lane >> 3, cch = (lane & 7) ^ l8;  // row inside a piece; logical 16-byte chunk that lives at position lane & 7
    const char* baseW = reinterpret_cast<const char*>(p.W);
    const char* baseX = w < 4 ? reinterpret_cast<const char*>((const T*)p.A + (size_t)bz * p.strideA) : baseW;
    unsigned offX[4], offY;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (w < 4) {
            int ar = m0 + 32 * w + 8 * j + l8;  // rows past M (last panel) clamp to the last row
            ar = ar < p.M ? ar : p.M - 1;
            offX[j] = (unsigned)(((size_t)ar * p.lda + cch * 8) * sizeof(T));
        } else {
            offX[j] = (unsigned)(((size_t)(32 * (w - 4) + 8 * j + l8) * p.ldw + cch * 8) * sizeof(T));
        }
    }
    offY = (unsigned)(((size_t)(128 + 32 * w + l8) * p.ldw + cch * 8) * sizeof(T));
    const unsigned stepY = (unsigned)(8 * p.ldw * sizeof(T));
    const int n_half = p.K >> 5;  // two half-stages per 64 k; K % 128 == 0 (launcher), so n_half % 4 == 0
    auto issue = [&](int h) {     // half-stage h -> slot h % 4
        T* slot = ring + (h & (NSLOT - 1)) * SLOT + w * 4 * 512;
        const unsigned k0 = (unsigned)((h >> 1) * 64 * sizeof(T));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const char* g = (h & 1) ? baseW + (offY + j * stepY + k0) : baseX + (offX[j] + k0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(slot + j * 512), 16, 0, 0);
        }
    };
#pragma unroll
    for (int h = 0; h < NSLOT - 1; ++h) issue(h);  // n_half >= 4
    typedef __attribute__((ext_vector_type(4))) int i32x4;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)fr_smem;
    // fragment byte addresses inside a slot: row R (== r16 mod 16), k-half kh: R*128 + (((kh*4 + g) ^ (R & 7)) << 4)
    const unsigned sw0 = (unsigned)((g ^ (r16 & 7)) << 4), sw1 = (unsigned)(((4 + g) ^ (r16 & 7)) << 4);
    const unsigned rA0 = lds0 + (unsigned)((wr * 64 + r16) * 128) + sw0, rA1 = rA0 - sw0 + sw1;           // X: A row blocks (+2048 i)
    const unsigned rX0 = lds0 + (unsigned)((128 + wc * 32 + r16) * 128) + sw0, rX1 = rX0 - sw0 + sw1;   // X: W blocks j = 0, 1
    const unsigned rY0 = lds0 + (unsigned)((wc * 64 + r16) * 128) + sw0, rY1 = rY0 - sw0 + sw1;          // Y: W blocks j = 2..5
    auto slot_off = [&](int h) { return (unsigned)(h & (NSLOT - 1)) * (unsigned)(SLOT * sizeof(T)); };
    // X fragments: a[2 i + kh] (8), wx[2 j + kh] (4);  Y fragments: wy[2 j + kh] (8)
    auto read_x = [&](int h, i32x4(&a)[8], i32x4(&wx)[4]) {
        const unsigned so = slot_off(h);
        const unsigned pa0 = rA0 + so, pa1 = rA1 + so, pw0 = rX0 + so, pw1 = rX1 + so;
        asm volatile(
            "ds_read_b128 %0, %12\n\tds_read_b128 %1, %13\n\tds_read_b128 %2, %12 offset:2048\n\tds_read_b128 %3, %13 offset:2048\n\t"
            "ds_read_b128 %4, %12 offset:4096\n\tds_read_b128 %5, %13 offset:4096\n\tds_read_b128 %6, %12 offset:6144\n\t"
            "ds_read_b128 %7, %13 offset:6144\n\t"
            "ds_read_b128 %8, %14\n\tds_read_b128 %9, %15\n\tds_read_b128 %10, %14 offset:2048\n\tds_read_b128 %11, %15 offset:2048"
            : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(a[4]), "=&v"(a[5]), "=&v"(a[6]), "=&v"(a[7]), "=&v"(wx[0]),
              "=&v"(wx[1]), "=&v"(wx[2]), "=&v"(wx[3])
            : "v"(pa0), "v"(pa1), "v"(pw0), "v"(pw1)
            : "memory");
    };
    auto read_y = [&](int h, i32x4(&wy)[8]) {
        const unsigned so = slot_off(h);
        const unsigned p0 = rY0 + so, p1 = rY1 + so;
        asm volatile(
            "ds_read_b128 %0, %8\n\tds_read_b128 %1, %9\n\tds_read_b128 %2, %8 offset:2048\n\tds_read_b128 %3, %9 offset:2048\n\t"
            "ds_read_b128 %4, %8 offset:4096\n\tds_read_b128 %5, %9 offset:4096\n\tds_read_b128 %6, %8 offset:6144\n\t"
            "ds_read_b128 %7, %9 offset:6144"
            : "=&v"(wy[0]), "=&v"(wy[1]), "=&v"(wy[2]), "=&v"(wy[3]), "=&v"(wy[4]), "=&v"(wy[5]), "=&v"(wy[6]), "=&v"(wy[7])
            : "v"(p0), "v"(p1)
            : "memory");
    };
#define WM_FR_HOLD_X(A_, W_) \
    "+v"(A_[0]), "+v"(A_[1]), "+v"(A_[2]), "+v"(A_[3]), "+v"(A_[4]), "+v"(A_[5]), "+v"(A_[6]), "+v"(A_[7]), "+v"(W_[0]), "+v"(W_[1]), "+v"(W_[2]), "+v"(W_[3])
#define WM_FR_HOLD_Y(W_) "+v"(W_[0]), "+v"(W_[1]), "+v"(W_[2]), "+v"(W_[3]), "+v"(W_[4]), "+v"(W_[5]), "+v"(W_[6]), "+v"(W_[7])
    auto frag = [&](const i32x4& v) {
        Frag<T> f;
        f.v = __builtin_bit_cast(decltype(f.v), v);
        return f;
    };
    auto mfma_x = [&](const i32x4(&a)[8], const i32x4(&wx)[4]) {
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[j][i] = mma32(frag(wx[2 * j + kh]), frag(a[2 * i + kh]), acc[j][i]);
    };
    auto mfma_y = [&](const i32x4(&a)[8], const i32x4(&wy)[8]) {
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[2 + j][i] = mma32(frag(wy[2 * j + kh]), frag(a[2 * i + kh]), acc[2 + j][i]);
    };
#define WM_FR_TOP(N) asm volatile("s_waitcnt vmcnt(" #N ")\n\ts_barrier" ::: "memory")
    i32x4 a0[8], a1[8], wx[4], wy[8];
    // step X(h): fragments of X(h) are in (ac, wx) [requested during the previous step]; requests Y(h+1) into wy
    auto step_x = [&](int h, i32x4(&ac)[8], auto YOUNGER, auto REFILL) {
        if constexpr (decltype(YOUNGER)::value)
            WM_FR_TOP(4);
        else
            WM_FR_TOP(0);
        if constexpr (decltype(REFILL)::value) issue(h + NSLOT - 1);
        read_y(h + 1, wy);
        asm volatile("s_waitcnt lgkmcnt(8)" : WM_FR_HOLD_X(ac, wx)::"memory");
        mfma_x(ac, wx);
        __builtin_amdgcn_sched_barrier(0);  // or the next step's wait + barrier is hoisted above these MFMAs (they touch no memory)
    };
    // step Y(h): fragments of Y(h) are in wy, the A fragments of its k64 in ac; requests X(h+1) into (an, wx) unless LAST
    auto step_y = [&](int h, i32x4(&ac)[8], i32x4(&an)[8], auto YOUNGER, auto REFILL, auto LAST) {
        if constexpr (!decltype(LAST)::value) {
            if constexpr (decltype(YOUNGER)::value)
                WM_FR_TOP(4);
            else
                WM_FR_TOP(0);
            if constexpr (decltype(REFILL)::value) issue(h + NSLOT - 1);
            read_x(h + 1, an, wx);
            asm volatile("s_waitcnt lgkmcnt(12)" : WM_FR_HOLD_Y(wy)::"memory");
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" : WM_FR_HOLD_Y(wy)::"memory");
        }
        mfma_y(ac, wy);
        __builtin_amdgcn_sched_barrier(0);
    };
    using Y = std::true_type;
    using N = std::false_type;
    WM_FR_TOP(8);  // half-stage 0 has landed (1 and 2 may be in flight)
    read_x(0, a0, wx);
    // steady state: groups of four half-stages (two k64), every wait the same constant and every refill unconditional; then the
    // last group with its shorter queues.  Before the top of step h the DMAs outstanding are those of half-stages h+1 and h+2.
    int h = 0;
    for (; h + 8 <= n_half; h += 4) {
        step_x(h, a0, Y{}, Y{});
        step_y(h + 1, a0, a1, Y{}, Y{}, N{});
        step_x(h + 2, a1, Y{}, Y{});
        step_y(h + 3, a1, a0, Y{}, Y{}, N{});
    }
    step_x(h, a0, Y{}, Y{});               // h == n_half - 4: the refill is half-stage n_half - 1
    step_y(h + 1, a0, a1, Y{}, N{}, N{});  // needs n_half - 2 landed; n_half - 1 may be in flight
    step_x(h + 2, a1, N{}, N{});           // needs n_half - 1 landed
    step_y(h + 3, a1, a0, N{}, N{}, Y{});
#undef WM_FR_TOP
#undef WM_FR_HOLD_X
#undef WM_FR_HOLD_Y
    // epilogue: acc[j][i][r] = C[m0 + 64 wr + 16 i + r16][colj(j) + 4 g + r];  out = act(acc + bias) + pos + residual.
    // Every addend row of all four row blocks is requested before anything is stored (the fragment registers are free now: 96
    // more live values fit): written row block by row block, each block's loads waited for the previous block's STORES too
    // (loads and stores share the in-order vmcnt) — four exposed memory round trips per workgroup, at one workgroup per CU.
    float* Cb = (float*)p.C + (size_t)bz * p.strideC;
    float sm[4] = {0.f, 0.f, 0.f, 0.f}, sq[4] = {0.f, 0.f, 0.f, 0.f};
    f32x4 bv[6], ex[4][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) bv[j] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + colj(j) + g * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    int mrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wr * 64 + i * 16 + r16;
        mrow[i] = m < p.M ? m : p.M - 1;
#pragma unroll
        for (int j = 0; j < 6; ++j) ex[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float* addend = p.residual ? p.residual + (size_t)bz * p.strideR + g * 4 : p.pos ? p.pos + g * 4 : nullptr;
    const long ld_add = p.residual ? p.ldr : p.N;
    if (addend) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) ex[i][j] = *reinterpret_cast<const f32x4*>(addend + (size_t)mrow[i] * ld_add + colj(j));
    }
    if (p.residual && p.pos) {  // both (no caller today): the positional rows in a second round
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) ex[i][j] += *reinterpret_cast<const f32x4*>(p.pos + g * 4 + (size_t)mrow[i] * p.N + colj(j));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bool valid = m0 + wr * 64 + i * 16 + r16 < p.M;
        float* crow = Cb + (size_t)mrow[i] * p.ldc + g * 4;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            f32x4 v = acc[j][i] + bv[j];
            if constexpr (ACT) v = gelu_fast4(v, p.gelu_mode);
            v += ex[i][j];
            if (valid) *reinterpret_cast<f32x4*>(crow + colj(j)) = v;
            if constexpr (LNO) {
                acc[j][i] = v;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sm[i] += v[r];
                    sq[i] += v[r] * v[r];
                }
            }
        }
    }
    if constexpr (LNO) {
        // LayerNorm of the finished rows (one-pass variance, whisper_tensor.mojo:249-285).  A row's 384 columns sit in 4 waves x 4
        // lanes: butterfly over the lanes, the four waves meet in LDS (the ring is free once every wave is past its last read).
        constexpr int PITCH = 208;  // bytes per scratch row (192 + 16: 52 dwords — 16 rows' 8-byte stores on distinct bank pairs)
        float* s_stat = reinterpret_cast<float*>(fr_smem + 8 * 64 * PITCH);  // [128 rows][4 waves][2]
        unsigned char* scr = fr_smem + w * 64 * PITCH;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sm[i] += __shfl_xor(sm[i], 16, 64);
            sq[i] += __shfl_xor(sq[i], 16, 64);
            sm[i] += __shfl_xor(sm[i], 32, 64);
            sq[i] += __shfl_xor(sq[i], 32, 64);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (g == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float* st = s_stat + ((wr * 64 + i * 16 + r16) * 4 + wc) * 2;
                st[0] = sm[i];
                st[1] = sq[i];
            }
        }
        __syncthreads();
        typedef __attribute__((ext_vector_type(4))) T t4;
        typedef __attribute__((ext_vector_type(8))) T t8;
        f32x4 gm[6], bt[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            gm[j] = *reinterpret_cast<const f32x4*>(p.lno_g + colj(j) + g * 4);
            bt[j] = *reinterpret_cast<const f32x4*>(p.lno_b + colj(j) + g * 4);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 s01 = *reinterpret_cast<const f32x4*>(s_stat + (wr * 64 + i * 16 + r16) * 8);
            const f32x4 s23 = *reinterpret_cast<const f32x4*>(s_stat + (wr * 64 + i * 16 + r16) * 8 + 4);
            const float s1 = (s01[0] + s01[2]) + (s23[0] + s23[2]), s2 = (s01[1] + s01[3]) + (s23[1] + s23[3]);
            const float mean = s1 / 384.0f;
            const float var = (s2 / 384.0f) - (mean * mean);
            const float inv_std = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const f32x4 v = acc[j][i];
                const t4 o = {from_f32<T>((v[0] - mean) * inv_std * gm[j][0] + bt[j][0]), from_f32<T>((v[1] - mean) * inv_std * gm[j][1] + bt[j][1]),
                              from_f32<T>((v[2] - mean) * inv_std * gm[j][2] + bt[j][2]), from_f32<T>((v[3] - mean) * inv_std * gm[j][3] + bt[j][3])};
                *reinterpret_cast<t4*>(scr + (i * 16 + r16) * PITCH + (j * 16 + g * 4) * 2) = o;  // wave-local column order
            }
        }
        // this wave's 64 rows x 6 column blocks, read back as 16-byte chunks of consecutive columns: 12 chunks per row, 768 in all
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        T* Lb = (T*)p.lno_out + (size_t)bz * p.strideC;
#pragma unroll
        for (int it = 0; it < 12; ++it) {
            const int c = it * 64 + lane, row = c / 12, col = c - row * 12;  // chunk col = half col & 1 of wave-local column block col >> 1
            const t8 v8 = *reinterpret_cast<const t8*>(scr + row * PITCH + col * 16);
            const int m = m0 + wr * 64 + row;
            const int jb = col >> 1;
            const int gcol = (jb < 2 ? (wc * 2 + jb) * 16 : 128 + (wc * 4 + jb - 2) * 16) + (col & 1) * 8;
            if (m < p.M) *reinterpret_cast<t8*>(Lb + (size_t)m * p.ldc + gcol) = v8;
        }
    }
}
static bool fullrow_ok(int operand_bytes, const GemmParams& p) {
    static const bool off = wm_env("WM_GEMM_NO_FULLROW") != nullptr || wm_env("WM_GEMM_DIRECT") != nullptr;
    return operand_bytes == 2 && !off && p.N == 384 && (p.K & 127) == 0 && p.group_n == 0 && !p.ln_g;
}
bool gemm_nt_fuses_layernorm_out(int operand_bytes, const GemmParams& p) {
    static const bool off = wm_env("WM_GEMM_NO_LN_OUT") != nullptr;
    return !off && fullrow_ok(operand_bytes, p);
}
template <typename T, bool LNO, bool ACT> static int launch_fullrow_t(const GemmParams& p, int batch, hipStream_t st) {
    const int lds = 4 * 256 * 64 * (int)sizeof(T);  // 128 KB: one workgroup per CU
    dim3 grid((p.M + 127) / 128, batch);
    if (const hipError_t e = ensure_dyn_lds<&gemm_nt_fullrow_kernel<T, LNO, ACT>>(lds); e != hipSuccess)
        return launch_hip_failed("full-row GEMM: dynamic LDS attribute", e);
    hipLaunchKernelGGL((gemm_nt_fullrow_kernel<T, LNO, ACT>), grid, dim3(512), lds, st, p);
    return WM_LAUNCH_OK;
}
template <typename T> static int launch_fullrow(const GemmParams& p, int batch, hipStream_t st) {
    if (p.lno_out) {
        if (p.act) return launch_fullrow_t<T, true, true>(p, batch, st);
        return launch_fullrow_t<T, true, false>(p, batch, st);
    }
    if (p.act) return launch_fullrow_t<T, false, true>(p, batch, st);
    return launch_fullrow_t<T, false, false>(p, batch, st);
}

// Refusals (nothing is launched, WM_LAUNCH_BAD_SHAPE): a fused LayerNorm asked of a shape whose kernel cannot apply it — the result
// would be a product of un-normalised rows / a missing operand copy — and shapes no kernel tiles (N % 128, K % 32).
template <typename T, typename TO> int launch_gemm_nt(const GemmParams& p, int batch, hipStream_t st) {
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || batch <= 0) return launch_refuse("gemm_nt: empty problem");
    if ((p.N & 127) != 0) return launch_refuse("gemm_nt: N must be a multiple of 128 (the dense kernels tile 128 output columns)");
    if ((p.K & 31) != 0) return launch_refuse("gemm_nt: K must be a multiple of 32 (one MFMA k-step)");
    if (p.ln_g && !rowpanel_ok((int)sizeof(T), p, batch))  // caller bug (see gemm_nt_fuses_layernorm)
        return launch_refuse("gemm_nt: LayerNorm fused into the A load asked of a shape only the plain kernels take");
    if constexpr (sizeof(T) == 2 && sizeof(TO) == 4) {
        if (fullrow_ok(2, p)) return launch_fullrow<T>(p, batch, st);
    }
    if (p.lno_out) return launch_refuse("gemm_nt: LayerNorm of the output rows asked of a shape the full-row kernel does not take");
    dim3 grid(p.N / 128, (p.M + 127) / 128, batch);
    static const bool no_lds = wm_env("WM_GEMM_DIRECT") != nullptr;
    if constexpr (sizeof(T) == 2) {
        if (rowpanel_ok((int)sizeof(T), p, batch)) return launch_rowpanel<T, TO, 12>(p, st);
        if (!no_lds && (p.K & 63) == 0) {
            static const bool no_remap = wm_env("WM_GEMM_NOXCD") != nullptr;
            GemmParams q = p;
            q.xcd_remap = !no_remap;
            hipLaunchKernelGGL((gemm_nt_lds_kernel<T, TO>), grid, dim3(256), 0, st, q);
            return WM_LAUNCH_OK;
        }
    }
    hipLaunchKernelGGL((gemm_nt_kernel<T, TO>), grid, dim3(256), 0, st, p);
    return WM_LAUNCH_OK;
}

// ------------------------------------------------------------------------------------------------------------
// LayerNorm over rows of `cols` fp32 (whisper_tensor.mojo:249-285: one-pass variance E[x²]-mean², eps inside the
// sqrt).  Half a wave per row, butterfly for Σx and Σx².  Writes the GEMM-operand copy (T) and/or an fp32 copy.
// Half a wave per row with 16-byte loads (cols = 128·j, j <= 8): lanes 0-31 take one row, 32-63 the next.
template <typename T, int V4 /* float4 per lane = cols/128 */>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, T* __restrict__ out_t,
                                                             float* __restrict__ out_f, int rows, float eps) {
    constexpr int cols = V4 * 128;
    const int lane = threadIdx.x & 63, half = lane >> 5, l32 = lane & 31;
    const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + half;
    const bool ok = row < rows;
    const float* xr = x + (size_t)(ok ? row : rows - 1) * cols;
    f32x4 v[V4];
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int i = 0; i < V4; ++i) {
        v[i] = *reinterpret_cast<const f32x4*>(xr + 4 * (l32 + 32 * i));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s += v[i][j];
            q += v[i][j] * v[i][j];
        }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {  // stays inside the 32-lane half
        s += __shfl_xor(s, o, 64);
        q += __shfl_xor(q, o, 64);
    }
    const float mean = s / (float)cols;
    const float var = (q / (float)cols) - (mean * mean);
    const float inv_std = 1.0f / sqrtf(var + eps);
    if (!ok) return;
#pragma unroll
    for (int i = 0; i < V4; ++i) {
        const int c = 4 * (l32 + 32 * i);
        const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c), bt = *reinterpret_cast<const f32x4*>(beta + c);
        f32x4 y;
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = (v[i][j] - mean) * inv_std * gm[j] + bt[j];
        if (out_t) {
            typedef __attribute__((ext_vector_type(4))) T t4;
            t4 o = {from_f32<T>(y[0]), from_f32<T>(y[1]), from_f32<T>(y[2]), from_f32<T>(y[3])};
            *reinterpret_cast<t4*>(out_t + (size_t)row * cols + c) = o;
        }
        if (out_f) *reinterpret_cast<f32x4*>(out_f + (size_t)row * cols + c) = y;
    }
}

template <typename T>
void launch_layernorm_rows(const float* x, const float* gamma, const float* beta, void* out_t, float* out_f, int rows,
                           int cols, float eps, hipStream_t st) {
    const dim3 grid((rows + 7) / 8), block(256);
#define WM_LN(V4) hipLaunchKernelGGL((layernorm_rows_kernel<T, V4>), grid, block, 0, st, x, gamma, beta, (T*)out_t, out_f, rows, eps)
    switch (cols / 128) {
        case 1: WM_LN(1); break;
        case 2: WM_LN(2); break;
        case 3: WM_LN(3); break;
        case 4: WM_LN(4); break;
        case 5: WM_LN(5); break;
        case 6: WM_LN(6); break;
        case 7: WM_LN(7); break;
        default: WM_LN(8); break;
    }
#undef WM_LN
}

// ------------------------------------------------------------------------------------------------------------
// Fused encoder self-attention (no mask): O = softmax(Q·Kᵀ·scale)·V per (utterance, head), never materialising the
// n_ctx x n_ctx scores the reference allocates per head (layers.mojo:293).  Scale is applied AFTER the dot product
// as the reference does (layers.mojo:306-308).
//
// Workgroup = 4 waves = 64 query rows (16 per wave) of one (b, h); K/V stream through LDS in 64-key tiles.
//   Sᵀ = K·Qᵀ  : A-operand = K fragment (rows = keys), B-operand = Q fragment (cols = q)  ->  lane (q = lane&15,
//                g = lane>>4) holds S[q][16kb + 4g + r]: the softmax row statistics are per-lane + a 4-lane reduce.
//   Oᵀ = Vᵀ·Pᵀ : the 8 probabilities a lane holds for one 32-key chunk ARE the B-operand fragment (k order permuted:
//                element j of group g <-> key 32c + 16(j>>2) + 4g + (j&3)); V is staged TRANSPOSED in LDS in that
//                same key order so the A-operand fragment is one contiguous 8-element read.  No LDS round trip for P,
//                and the O accumulator has q on the lane too, so rescaling needs no cross-lane traffic.
template <typename T> struct AttnLds {
    static constexpr int PAD = 16 / sizeof(T);
    static constexpr int PITCH = 64 + PAD;  // 144 B (16-bit) / 272 B (fp32): 16-lane b128 reads hit 16 distinct slots
};

template <typename T, bool FAST>
__global__ __launch_bounds__(256) void flash_attn_enc_kernel(const T* __restrict__ qkv, T* __restrict__ out, int n_ctx,
                                                             int d_model, float scale) {
    constexpr int PITCH = AttnLds<T>::PITCH;
    __shared__ __attribute__((aligned(16))) T Ks[64 * PITCH];
    __shared__ __attribute__((aligned(16))) T Vt[64 * PITCH];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const int h = blockIdx.y, b = blockIdx.z;
    const size_t ld = (size_t)3 * d_model;
    const T* base = qkv + (size_t)b * n_ctx * ld;
    const int q_row = blockIdx.x * 64 + wid * 16 + r16;
    const int q_ld = q_row < n_ctx ? q_row : n_ctx - 1;
    Frag<T> qf[2];
#pragma unroll
    for (int dc = 0; dc < 2; ++dc) qf[dc] = load_frag<T>(base + (size_t)q_ld * ld + h * 64 + dc * 32 + g * 8);

    f32x4 o[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) o[db] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -1e30f, l_run = 0.f;

    const T* kbase = base + d_model + h * 64;
    const T* vbase = base + 2 * d_model + h * 64;
    const int n_tiles = (n_ctx + 63) / 64;
    for (int t = 0; t < n_tiles; ++t) {
        const int key0 = t * 64;
        __syncthreads();  // previous tile fully consumed
        // stage K (row-major) and V (transposed, permuted key order): 512 chunks of 8 elements each
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int chunk = threadIdx.x + it * 256;
            const int key = chunk >> 3, dch = chunk & 7;
            int krow = key0 + key;
            krow = krow < n_ctx ? krow : n_ctx - 1;
            Frag<T> kf = load_frag<T>(kbase + (size_t)krow * ld + dch * 8);
            Frag<T> vf = load_frag<T>(vbase + (size_t)krow * ld + dch * 8);
            if constexpr (sizeof(T) == 2) {
                *reinterpret_cast<decltype(kf.v)*>(&Ks[key * PITCH + dch * 8]) = kf.v;
            } else {
                *reinterpret_cast<f32x4*>(&Ks[key * PITCH + dch * 8]) = f32x4{kf.v[0], kf.v[1], kf.v[2], kf.v[3]};
                *reinterpret_cast<f32x4*>(&Ks[key * PITCH + dch * 8 + 4]) = f32x4{kf.v[4], kf.v[5], kf.v[6], kf.v[7]};
            }
            const int c = key >> 5, kk = key & 31;
            const int pos = 32 * c + 8 * ((kk >> 2) & 3) + 4 * (kk >> 4) + (kk & 3);
#pragma unroll
            for (int e = 0; e < 8; ++e) Vt[(dch * 8 + e) * PITCH + pos] = vf.v[e];
        }
        __syncthreads();

        // Sᵀ tile: 4 key blocks x 2 dim chunks
        f32x4 s[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            s[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dc = 0; dc < 2; ++dc) {
                Frag<T> kf = load_frag<T>(&Ks[(kb * 16 + r16) * PITCH + dc * 32 + g * 8]);
                s[kb] = mma32(kf, qf[dc], s[kb]);
            }
        }
        float tmax = -1e30f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = s[kb][r] * scale;
                if (key0 + kb * 16 + g * 4 + r >= n_ctx) v = -1e30f;
                s[kb][r] = v;
                tmax = fmaxf(tmax, v);
            }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = FAST ? __expf(m_run - m_new) : expf(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
        float pv[2][8];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float pe = FAST ? __expf(s[kb][r] - m_new) : expf(s[kb][r] - m_new);
                psum += pe;
                pv[kb >> 1][(kb & 1) * 4 + r] = pe;
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int db = 0; db < 4; ++db) o[db] *= alpha;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            Frag<T> pf = make_frag<T>(pv[c]);
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                Frag<T> vf = load_frag<T>(&Vt[(db * 16 + r16) * PITCH + c * 32 + g * 8]);
                o[db] = mma32(vf, pf, o[db]);
            }
        }
    }
    float l = l_run;
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    if (q_row < n_ctx) {
        T* orow = out + ((size_t)b * n_ctx + q_row) * d_model + h * 64;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            typedef __attribute__((ext_vector_type(4))) T t4;
            t4 ov = {from_f32<T>(o[db][0] * inv), from_f32<T>(o[db][1] * inv), from_f32<T>(o[db][2] * inv),
                     from_f32<T>(o[db][3] * inv)};
            *reinterpret_cast<t4*>(orow + db * 16 + g * 4) = ov;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// v2 of the fused encoder attention for 16-bit operands (same math and operand maps as above):
//   * 128 query rows per workgroup (two 16-row query blocks per wave): every K / V fragment read from LDS feeds two
//     MFMAs, halving LDS traffic per FLOP;
//   * K and V tiles are both staged row-major with 16-byte stores; the Vᵀ operand fragment is produced by the
//     hardware transposing read ds_read_b64_tr_b16 (a 16-lane group reads a 4-key x 16-dim block column-major, which
//     is exactly the permuted-key fragment: keys 32c+4g+q then 32c+16+4g+q) — no scalar transposed writes;
//   * the next tile is fetched global -> registers while this one is consumed and lands in the other LDS buffer:
//     one barrier per tile and no exposed global latency;
//   * softmax in the exp2 domain (scale·log2e folded into one multiply, v_exp_f32 directly), key mask only on the
//     last tile.
__device__ __forceinline__ bf16x4 lds_tr16(const bf16* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p);
}
__device__ __forceinline__ f16x4 lds_tr16(const f16* p) {
    typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 h4;
    const h4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4*)p);
    return __builtin_bit_cast(f16x4, v);
}

template <typename T, int NW, int QB, int WPS = 1, int OPT = 0 /* 1: exact running maximum on every tile (developer A/B) */>
__global__ __launch_bounds__(NW * 64, WPS) void flash_attn_enc_v2_kernel(const T* __restrict__ qkv, T* __restrict__ out, int n_ctx,
                                                                    int d_model, float scale_log2e, int xcd_remap) {
    // 160-byte rows (40 dwords): with the hardware's lane groups (ds_read_b128: {0-3,12-15,20-27}, ...; ds_read_b64_tr:
    // 32 lanes = 8 rows x 32 B) the 16 fragment rows / 8 transposed rows of one access fall on disjoint banks.  (144-byte
    // rows had 2-way conflicts on both: 6 % of the kernel's wave cycles in SQ_LDS_BANK_CONFLICT.)
    constexpr int PITCH = 80;
    __shared__ __attribute__((aligned(16))) T Ks[2][64 * PITCH];
    __shared__ __attribute__((aligned(16))) T Vs[2][64 * PITCH];
    typedef __attribute__((ext_vector_type(8))) T t8;
    typedef __attribute__((ext_vector_type(4))) T t4;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    constexpr int ROWS = NW * QB * 16;  // query rows per workgroup
    int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    if (xcd_remap) {
        // Workgroups are dealt round-robin over the 8 XCDs (private L2s).  Give every XCD whole (utterance, head) pairs
        // so the K/V of a pair is fetched into ONE L2 and re-read there by all of the pair's query tiles.
        const int nq = gridDim.x, npair = gridDim.y * gridDim.z;  // host guarantees npair % 8 == 0
        const int lin = blockIdx.x + nq * (blockIdx.y + gridDim.y * blockIdx.z);
        const int xcd = lin & 7, idx = lin >> 3;
        const int pair = xcd * (npair >> 3) + idx / nq;
        qt = idx % nq;
        h = pair % gridDim.y;
        b = pair / gridDim.y;
    }
    const size_t ld = (size_t)3 * d_model;
    const T* base = qkv + (size_t)b * n_ctx * ld;
    const T* kbase = base + d_model + h * 64;
    const T* vbase = base + 2 * d_model + h * 64;

    Frag<T> qf[QB][2];
    int q_row[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        q_row[qb] = qt * ROWS + wid * (QB * 16) + qb * 16 + r16;
        const int q_ld = q_row[qb] < n_ctx ? q_row[qb] : n_ctx - 1;
#pragma unroll
        for (int dc = 0; dc < 2; ++dc) qf[qb][dc] = load_frag<T>(base + (size_t)q_ld * ld + h * 64 + dc * 32 + g * 8);
    }
    f32x4 o[QB][4];
    float m_run[QB], l_run[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        m_run[qb] = -1e30f;
        l_run[qb] = 0.f;
#pragma unroll
        for (int db = 0; db < 4; ++db) o[qb][db] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // staging map: 512 16-byte chunks per tile and operand; thread t owns chunks t (+ NW*64 when 4 waves)
    constexpr int NIT = 512 / (NW * 64);
    const int skey[2] = {(int)threadIdx.x >> 3, ((int)threadIdx.x + 256) >> 3};
    const int sdch = threadIdx.x & 7;
    t8 kreg[NIT], vreg[NIT];
    // Global addresses as (uniform 64-bit base) + (32-bit byte offset per lane): the offset walks down the utterance by one
    // tile per fetch and is clamped to the last valid key row — one v_add + one v_min per fetch instead of a 64-bit address
    // rebuilt from the row index every tile.
    const char* kb8 = reinterpret_cast<const char*>(kbase);
    const char* vb8 = reinterpret_cast<const char*>(vbase);
    unsigned boff[NIT], bmax;
#pragma unroll
    for (int it = 0; it < NIT; ++it) boff[it] = (unsigned)((size_t)skey[it] * ld + sdch * 8) * (unsigned)sizeof(T);
    bmax = (unsigned)((size_t)(n_ctx - 1) * ld + sdch * 8) * (unsigned)sizeof(T);
    const unsigned bstep = (unsigned)(64 * ld * sizeof(T));
    auto fetch = [&]() {  // the next tile in sequence
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const unsigned o = boff[it] < bmax ? boff[it] : bmax;
            kreg[it] = *reinterpret_cast<const t8*>(kb8 + o);
            vreg[it] = *reinterpret_cast<const t8*>(vb8 + o);
            boff[it] += bstep;
        }
    };
    auto park = [&](int buf) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            *reinterpret_cast<t8*>(&Ks[buf][skey[it] * PITCH + sdch * 8]) = kreg[it];
            *reinterpret_cast<t8*>(&Vs[buf][skey[it] * PITCH + sdch * 8]) = vreg[it];
        }
    };
    const int n_tiles = (n_ctx + 63) / 64;
    fetch();
    park(0);
    __syncthreads();
    // One 64-key tile.  MASKED (keys past n_ctx set to -1e30) only for the LAST tile, as its own copy of the body: written as a
    // run-time predicate the compiler keeps 16 v_cmp + 32 v_cndmask in every tile (48 of 155 VALU issues per tile, and this
    // kernel is VALU-bound).  Scaling, exp2 argument, row sum and accumulator rescale are written on 4-vectors so that they
    // become packed fp32 instructions (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32: two lanes of work per issue).
    auto tile = [&](int t, auto MASKED, auto MORE) {
        const int buf = t & 1;
        if constexpr (decltype(MORE)::value) fetch();
        const T* Kt = Ks[buf];
        const T* Vt = Vs[buf];
        // Sᵀ = K·Qᵀ
        f32x4 sc[QB][4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            Frag<T> kf0 = load_frag<T>(&Kt[(kb * 16 + r16) * PITCH + g * 8]);
            Frag<T> kf1 = load_frag<T>(&Kt[(kb * 16 + r16) * PITCH + 32 + g * 8]);
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
                z = mma32(kf0, qf[qb][0], z);
                sc[qb][kb] = mma32(kf1, qf[qb][1], z);
            }
        }
        Frag<T> pf[QB][2];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            if constexpr (decltype(MASKED)::value) {
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (t * 64 + kb * 16 + g * 4 + r >= n_ctx) sc[qb][kb][r] = -1e30f;
            }
            // Softmax in the exp2 domain against a REFERENCE maximum m_run that is only refreshed when it has to be.  Any reference
            // gives the same softmax as long as nothing overflows (p, the row sum and the accumulator scale together; fp32 and
            // bf16 share their exponent range), so the fast path takes p = exp2(s * scale*log2e - m_run) with the reference as it
            // stands: no row maximum (8 v_max3 + 6 v_max + two cross-lane round trips), no alpha, no accumulator rescale.
            // A wave takes the exact path (refresh the reference to the running maximum, as before) when one of its rows'
            // partial sums leaves [0, 1024]: always on the first tile (m_run = -1e30 gives inf), and whenever a row's scores
            // have outgrown its reference by ~2^6 or more.
            const f32x4 sl4 = f32x4{scale_log2e, scale_log2e, scale_log2e, scale_log2e};
            float pv[2][8];
            auto probs = [&](float m) {  // p of this lane's 16 scores against reference m; returns their sum
                const f32x4 nm4 = f32x4{-m, -m, -m, -m};
                f32x4 ps4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    const f32x4 a = __builtin_elementwise_fma(sc[qb][kb], sl4, nm4);
                    f32x4 pe;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        pe[r] = __builtin_amdgcn_exp2f(a[r]);
                        pv[kb >> 1][(kb & 1) * 4 + r] = pe[r];
                    }
                    ps4 += pe;
                }
                return (ps4[0] + ps4[1]) + (ps4[2] + ps4[3]);
            };
            float psum = (OPT & 1) ? INFINITY : probs(m_run[qb]);
            if (__any(!(psum <= 1024.f))) {
                float tmax = -1e30f;
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) tmax = fmaxf(tmax, sc[qb][kb][r]);
                tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
                const float m_new = fmaxf(m_run[qb], tmax * scale_log2e);
                const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_new);
                m_run[qb] = m_new;
                psum = probs(m_new);
                l_run[qb] *= alpha;
#pragma unroll
                for (int db = 0; db < 4; ++db) o[qb][db] *= alpha;
            }
            l_run[qb] += psum;
            pf[qb][0] = make_frag<T>(pv[0]);
            pf[qb][1] = make_frag<T>(pv[1]);
        }
        // Oᵀ += Vᵀ·Pᵀ : Vᵀ fragments by transposing reads of the row-major V tile
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const T* vp = &Vt[(32 * c + 4 * g + (r16 >> 2)) * PITCH + db * 16 + (r16 & 3) * 4];
                const t4 lo = lds_tr16(vp);
                const t4 hi = lds_tr16(vp + 16 * PITCH);
                Frag<T> vf;
                vf.v = t8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) o[qb][db] = mma32(vf, pf[qb][c], o[qb][db]);
            }
        if constexpr (decltype(MORE)::value) {
            park(buf ^ 1);
            __syncthreads();
        }
    };
    for (int t = 0; t + 1 < n_tiles; ++t) tile(t, std::false_type{}, std::true_type{});
    tile(n_tiles - 1, std::true_type{}, std::false_type{});
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        float l = l_run[qb];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float inv = 1.0f / l;
        if (q_row[qb] < n_ctx) {
            T* orow = out + ((size_t)b * n_ctx + q_row[qb]) * d_model + h * 64;
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                t4 ov = {from_f32<T>(o[qb][db][0] * inv), from_f32<T>(o[qb][db][1] * inv), from_f32<T>(o[qb][db][2] * inv),
                         from_f32<T>(o[qb][db][3] * inv)};
                *reinterpret_cast<t4*>(orow + db * 16 + g * 4) = ov;
            }
        }
    }
}

template <typename T>
void launch_flash_attn_enc(const void* qkv, void* out, int B, int H, int n_ctx, float scale, hipStream_t st) {
    if constexpr (sizeof(T) == 2) {
        // 16-bit operands: 8 waves x one 16-row query block (128 query rows per workgroup), three workgroups per CU
        // (WPS = 6 waves per SIMD caps the kernel at 80 VGPRs — 91 unconstrained, 7 spilled: the third workgroup hides more of
        // the per-tile barrier / LDS latency than the spills cost; 471 -> 447 us per launch at 64 clips).
        // Measured alternatives (tiny, encoder ms per 64 clips at 8 utterances per pass): 8x1 9.73 | 4x1 9.75 | 4x2 10.13 | 8x2 10.31.
        const int remap = (B * H) % 8 == 0 ? 1 : 0;
        const float sl = scale * 1.4426950408889634f;
#ifdef WM_DEV  // A/B variants of the developer build: WM_ATTN_V1 (generic kernel), WM_ATTN_VAR=2..4 (other tilings), WM_ATTN_WPS1, WM_ATTN_NOXCD
        static const bool v1 = wm_env("WM_ATTN_V1") != nullptr;
        static const int var = wm_env("WM_ATTN_VAR") ? atoi(wm_env("WM_ATTN_VAR")) : 1;
        static const bool no_xcd = wm_env("WM_ATTN_NOXCD") != nullptr;
        static const bool wps1 = wm_env("WM_ATTN_WPS1") != nullptr;
        const int rm = no_xcd ? 0 : remap;
        if (v1) {
            hipLaunchKernelGGL((flash_attn_enc_kernel<T, true>), dim3((n_ctx + 63) / 64, H, B), dim3(256), 0, st, (const T*)qkv, (T*)out, n_ctx, H * 64, scale);
            return;
        }
        if (var == 2) {
            hipLaunchKernelGGL((flash_attn_enc_v2_kernel<T, 8, 2>), dim3((n_ctx + 255) / 256, H, B), dim3(512), 0, st, (const T*)qkv, (T*)out, n_ctx, H * 64, sl, rm);
            return;
        } else if (var == 3) {
            hipLaunchKernelGGL((flash_attn_enc_v2_kernel<T, 4, 1>), dim3((n_ctx + 63) / 64, H, B), dim3(256), 0, st, (const T*)qkv, (T*)out, n_ctx, H * 64, sl, rm);
            return;
        } else if (var == 4) {
            hipLaunchKernelGGL((flash_attn_enc_v2_kernel<T, 4, 2>), dim3((n_ctx + 127) / 128, H, B), dim3(256), 0, st, (const T*)qkv, (T*)out, n_ctx, H * 64, sl, rm);
            return;
        } else if (var == 5) {  // exact running maximum on every tile
            hipLaunchKernelGGL((flash_attn_enc_v2_kernel<T, 8, 1, 6, 1>), dim3((n_ctx + 127) / 128, H, B), dim3(512), 0, st, (const T*)qkv, (T*)out, n_ctx, H * 64, sl, rm);
            return;
        } else if (wps1 || no_xcd) {
            hipLaunchKernelGGL((flash_attn_enc_v2_kernel<T, 8, 1, 1>), dim3((n_ctx + 127) / 128, H, B), dim3(512), 0, st, (const T*)qkv, (T*)out, n_ctx, H * 64, sl, rm);
            return;
        }
#endif
        hipLaunchKernelGGL((flash_attn_enc_v2_kernel<T, 8, 1, 6>), dim3((n_ctx + 127) / 128, H, B), dim3(512), 0, st, (const T*)qkv, (T*)out, n_ctx, H * 64, sl, remap);
    } else {  // exact-fp32 operands: the generic kernel
        hipLaunchKernelGGL((flash_attn_enc_kernel<T, false>), dim3((n_ctx + 63) / 64, H, B), dim3(256), 0, st, (const T*)qkv, (T*)out, n_ctx, H * 64, scale);
    }
}

// ---- explicit instantiations --------------------------------------------------------------------------------
#define WM_INST_T(T)                                                                                          \
    template void launch_mel_transpose_pad<T>(const float*, void*, int, int, int, int, hipStream_t);          \
    template void launch_layernorm_rows<T>(const float*, const float*, const float*, void*, float*, int, int, \
                                           float, hipStream_t);                                               \
    template void launch_flash_attn_enc<T>(const void*, void*, int, int, int, float, hipStream_t);
WM_INST_T(float)
WM_INST_T(bf16)
WM_INST_T(f16)
template int launch_gemm_nt<float, float>(const GemmParams&, int, hipStream_t);
template int launch_gemm_nt<bf16, float>(const GemmParams&, int, hipStream_t);
template int launch_gemm_nt<bf16, bf16>(const GemmParams&, int, hipStream_t);
template int launch_gemm_nt<f16, float>(const GemmParams&, int, hipStream_t);
template int launch_gemm_nt<f16, f16>(const GemmParams&, int, hipStream_t);

}  // namespace wm
