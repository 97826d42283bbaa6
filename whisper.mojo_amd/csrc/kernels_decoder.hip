// kernels_decoder.hip — the HBM-bound half of the path: one KV-cached decode step for B utterances.
//
// Replaces (reference file:line): the M<=4 branch of matmul (whisper_tensor.mojo:158-175) for every decoder
// projection incl. the tied 51 865-way logits (whisper.mojo:162-166); layer_norm (whisper_tensor.mojo:249-285) fused
// as a prologue; gelu (:288-308) and the residual adds (layers.mojo:457-461,483-487,513-517) fused as epilogues;
// the KV-cache append (layers.mojo:140-147); the q_len==1 attention path (layers.mojo:186-272); the embedding +
// position add (whisper.mojo:141-149); argmax (whisper_tensor.mojo:431-439).
#include "wm_kernels.h"

#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace wm {

// ------------------------------------------------------------------------------------------------------------
// Developer timeline: ts[0] = entry count, then (tag, 100 MHz clock) pairs; tag = owner id << 8 | kind
// (0 cross-attention start, 1 merge start, 2 logits start, 3 argmax start).  Off (null) outside WM_TRACE_EVENTS=<file>.
__device__ __forceinline__ void ts_put(long long* ts, int id, int kind) {
    const unsigned long long i = atomicAdd((unsigned long long*)ts, 1ull);
    if (i < (1ull << 20)) {
        ts[1 + 2 * i] = ((long long)id << 8) | kind;
        ts[2 + 2 * i] = (long long)wall_clock64();
    }
}

// x[b] = token_emb[tok[b]] + pos_emb[pos[b]]      (whisper.mojo:141-149)
__global__ void dec_embed_kernel(const float* __restrict__ tok_emb, const float* __restrict__ pos_emb,
                                 const int* __restrict__ tok, const int* __restrict__ pos, float* __restrict__ x, int d) {
    const int b = blockIdx.x;
    const float* t = tok_emb + (size_t)tok[b] * d;
    const float* p = pos_emb + (size_t)pos[b] * d;
    for (int j = threadIdx.x; j < d; j += blockDim.x) x[(size_t)b * d + j] = t[j] + p[j];
}
void launch_dec_embed(const float* tok_emb, const float* pos_emb, const int* tok, const int* pos, float* x, int B, int d,
                      hipStream_t st) {
    hipLaunchKernelGGL(dec_embed_kernel, dim3(B), dim3(128), 0, st, tok_emb, pos_emb, tok, pos, x, d);
}

// ------------------------------------------------------------------------------------------------------------
// Skinny linear: out[B,N] = epi( pro(x)[B,K] · W[N,K]ᵀ + bias ).  Every weight byte is read from HBM once per step,
// all B rows sharing the pass — the batched form of the reference's "parallel over n, dot over K" GEMV
// (whisper_tensor.mojo:158-175).
//
// These launches move 0.3-1.2 MB: they are LATENCY bound, and what they wait for is (a) one memory round trip and
// (b) the serial instruction stream of a wave (a lone wave retires ~1 instruction per 4-5 cycles: 1 700 instructions
// = 4 us, measured).  So the kernel minimises the per-wave stream:
//   * workgroup = one 16x16 output tile (16 utterances x 16 columns); its NW <= 16 waves split K, wave w owning
//     k-steps w, w+NW, … (KPW = 1 for K = d_model): a wave's whole job is KPW weight fragments, KPW activation
//     fragments and KPW MFMAs, everything requested up front;
//   * LayerNorm prologue: row statistics (one-pass variance, as the reference) from the SAME activation registers —
//     8·KPW values per lane, a 4-lane butterfly, one LDS exchange between the NW waves;
//   * computed as outᵀ (A-operand = weight fragment): a lane owns 4 consecutive columns of one utterance; the NW
//     K-partials meet in LDS and wave 0 finishes (bias / GELU / residual / KV-cache append), with its epilogue operands
//     requested before anything else so they ride the same round trip.
// dtype-tagged store of one fp32 value / four fp32 values (out_dtype: 0 fp32, 1 bf16, 2 f16)
__device__ __forceinline__ void store_as(void* base, size_t idx, float v, int dt) {
    if (dt == 0)
        ((float*)base)[idx] = v;
    else if (dt == 1)
        ((bf16*)base)[idx] = (bf16)v;
    else
        ((f16*)base)[idx] = (f16)v;
}

template <typename TW, int KPW, bool LN, int NT, bool XT = false>
__global__ __launch_bounds__(1024) void dec_linear_kernel(DecLinearParams p) {
    static_assert(!(LN && XT), "the LayerNorm prologue reads the fp32 residual stream");
#ifdef WM_DEV
#define WM_DL_STAMP(k) do { if (p.dbg && (threadIdx.x & 63) == 0) p.dbg[(((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + (threadIdx.x >> 6)) * 8 + (k)] = (long long)wall_clock64(); } while (0)
#else
#define WM_DL_STAMP(k) do { } while (0)
#endif
    WM_DL_STAMP(0);
    // dynamic LDS, sized by the launcher for the NW waves actually used: [NW][NT][64] f32x4 K-partials, then [NW][16][2]
    // LayerNorm statistics (24.6 + 1.5 KB for the 12-wave QKV / fc1 launches: fits beside two 64-KB GEMM workgroups)
    extern __shared__ __attribute__((aligned(16))) unsigned char dl_smem[];
    f32x4 (*s_red)[NT][64] = reinterpret_cast<f32x4 (*)[NT][64]>(dl_smem);
    float (*s_stat)[16][2] = reinterpret_cast<float (*)[16][2]>(dl_smem + (size_t)(blockDim.x >> 6) * NT * 64 * sizeof(f32x4));
    // LayerNorm gamma | beta of each wave's k-steps, [NW][KPW][32 gamma, 32 beta]: ONE 4-byte load per lane and k-step, handed to
    // the 16 lanes that need each value through LDS.  (Every lane fetching its own 8 + 8 values was four 1-KB load instructions per
    // k-step — 40 % of what an LN1+QKV workgroup put through its CU's address path: 64 B per clock is what bounds these launches.)
    float (*s_gbl)[KPW][64] = reinterpret_cast<float (*)[KPW][64]>(dl_smem + (size_t)(blockDim.x >> 6) * (NT * 64 * sizeof(f32x4) + 16 * 2 * sizeof(float)));
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * (16 * NT), b0 = blockIdx.y * 16;
    int xrow = b0 + r16;
    xrow = xrow < p.B ? xrow : p.B - 1;
    const TW* wp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        int wrow = n0 + t * 16 + r16;
        wrow = wrow < p.N ? wrow : p.N - 1;
        wp[t] = (const TW*)p.W + (size_t)wrow * p.K + g * 8;
    }
    const float* xp = p.x + (size_t)xrow * p.ldx + g * 8;
    const TW* xpt = (const TW*)p.x + (size_t)xrow * p.ldx + g * 8;  // XT: the same row in operand dtype
    // epilogue operands (wave t finishes column tile t)
    const int eb = b0 + r16, en = n0 + (w < NT ? w : 0) * 16 + g * 4;
    // a second column tile that starts past the (16-padded) width does not exist: storing it would land in the next row
    const bool epi = w < NT && eb < p.B && n0 + (w < NT ? w : 0) * 16 < ((p.N + 15) & ~15);
    float bias4[4] = {0.f, 0.f, 0.f, 0.f};
    f32x4 res4 = f32x4{0.f, 0.f, 0.f, 0.f};
    int cache_row = 0;
    if (epi) {
        if (p.bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r) bias4[r] = p.bias[min(en + r, p.N - 1)];
        }
        if (p.residual) res4 = *reinterpret_cast<const f32x4*>(p.residual + (size_t)eb * p.ldr + en);
        // (the address is uniform, so hipcc emits a scalar load whatever the source says — kernarg -> ctl -> len is one more
        //  scalar round trip in the prologue; in-kernel stamps showed the prologue's scalar stages are not what delays the loads)
        if (p.kcache) cache_row = __builtin_nontemporal_load(&p.ctl->len);
    }
    Frag<TW> wf[NT][KPW];
    Frag<TW> xft[XT ? KPW : 1];
    f32x4 xa[XT ? 1 : KPW][2];
    float gbv[LN ? KPW : 1];
    // every weight load first: they come from HBM / MALL (~1 us), the activation and LayerNorm loads behind them are L2 hits,
    // and a wave issues loads only as fast as the CU's address path takes them (24 x 1 KB per wave here)
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
        const int k = (w + nw * i) * 32;
#pragma unroll
        for (int t = 0; t < NT; ++t) wf[t][i] = load_frag<TW>(wp[t] + k);
    }
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
        const int k = (w + nw * i) * 32;
        if constexpr (XT) {
            xft[i] = load_frag<TW>(xpt + k);
        } else {
            xa[i][0] = *reinterpret_cast<const f32x4*>(xp + k);
            xa[i][1] = *reinterpret_cast<const f32x4*>(xp + k + 4);
        }
        if (LN) gbv[i] = (lane < 32 ? p.ln_g : p.ln_b - 32)[k + lane];
    }
    WM_DL_STAMP(1);
    float mean = 0.f, rstd = 1.f;
    if constexpr (LN) {
        float sm = 0.f, sq = 0.f;
#pragma unroll
        for (int i = 0; i < KPW; ++i)
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = xa[i][h2][j];
                    sm += v;
                    sq += v * v;
                }
        sm += __shfl_xor(sm, 16, 64);
        sq += __shfl_xor(sq, 16, 64);
        sm += __shfl_xor(sm, 32, 64);
        sq += __shfl_xor(sq, 32, 64);
        if (g == 0) {
            s_stat[w][r16][0] = sm;
            s_stat[w][r16][1] = sq;
        }
#pragma unroll
        for (int i = 0; i < KPW; ++i) s_gbl[w][i][lane] = gbv[i];
        __syncthreads();
        sm = 0.f;
        sq = 0.f;
        for (int k = 0; k < nw; ++k) {
            sm += s_stat[k][r16][0];
            sq += s_stat[k][r16][1];
        }
        mean = sm / (float)p.K;
        const float var = (sq / (float)p.K) - (mean * mean);
        rstd = 1.0f / sqrtf(var + 1e-5f);
    }
    WM_DL_STAMP(2);
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
        Frag<TW> xf;
        if constexpr (XT) {
            xf = xft[i];
        } else {
            float xv[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xv[j] = xa[i][0][j];
                xv[4 + j] = xa[i][1][j];
            }
            if (LN) {
                const f32x4 gm0 = *reinterpret_cast<const f32x4*>(&s_gbl[w][i][g * 8]), gm1 = *reinterpret_cast<const f32x4*>(&s_gbl[w][i][g * 8 + 4]);
                const f32x4 bt0 = *reinterpret_cast<const f32x4*>(&s_gbl[w][i][32 + g * 8]), bt1 = *reinterpret_cast<const f32x4*>(&s_gbl[w][i][32 + g * 8 + 4]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    xv[j] = (xv[j] - mean) * rstd * gm0[j] + bt0[j];
                    xv[4 + j] = (xv[4 + j] - mean) * rstd * gm1[j] + bt1[j];
                }
            }
            xf = make_frag<TW>(xv);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = mma32(wf[t][i], xf, acc[t]);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) s_red[w][t][lane] = acc[t];
    WM_DL_STAMP(3);
    __syncthreads();
    WM_DL_STAMP(4);
    if (epi) {
        f32x4 v = s_red[0][w][lane];
        for (int k = 1; k < nw; ++k) v += s_red[k][w][lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += bias4[r];
        if (p.act) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = gelu_f(v[r], p.gelu_mode);
        }
        v += res4;
        if (p.kcache && en >= p.d_model) {
            const bool is_v = en >= 2 * p.d_model;
            const int cc = en - (is_v ? 2 : 1) * p.d_model;
            const int ub = p.kv_B > 0 ? eb % p.kv_B : eb, ut = p.kv_B > 0 ? eb / p.kv_B : 0;  // prefill rows: (position, utterance)
            const size_t off = (size_t)ub * p.kv_batch_stride + (size_t)(cache_row + ut) * p.d_model + cc;
            void* basep = is_v ? p.vcache : p.kcache;
            if (p.kv_dtype == 0) {
                *reinterpret_cast<f32x4*>((float*)basep + off) = v;
            } else if (p.kv_dtype == 1) {
                bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                *reinterpret_cast<bf16x4*>((bf16*)basep + off) = o;
            } else {
                f16x4 o = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                *reinterpret_cast<f16x4*>((f16*)basep + off) = o;
            }
        } else if (p.out_is_t && sizeof(TW) == 2) {
            typedef __attribute__((ext_vector_type(4))) TW t4;
            const t4 o = {from_f32<TW>(v[0]), from_f32<TW>(v[1]), from_f32<TW>(v[2]), from_f32<TW>(v[3])};
            *reinterpret_cast<t4*>((TW*)p.out + (size_t)eb * p.ldo + en) = o;
        } else {
            *reinterpret_cast<f32x4*>(p.out + (size_t)eb * p.ldo + en) = v;
        }
    }
    WM_DL_STAMP(5);
}
#undef WM_DL_STAMP
template <typename TW, int KPW> static int launch_dec_linear_t(const DecLinearParams& p, int nw, hipStream_t st) {
    // two column tiles per workgroup for the wide projections (QKV, fc1): half the workgroups, one activation
    // fragment (and one LayerNorm) feeding two MFMAs
    static const int nt_force = wm_env("WM_LIN_NT") ? atoi(wm_env("WM_LIN_NT")) : 0;  // dev A/B
    const bool wide = nt_force ? nt_force == 2 : (p.N >= 1024 && nw >= 2);
    const dim3 grid((p.N + (wide ? 31 : 15)) / (wide ? 32 : 16), (p.B + 15) / 16), block(nw * 64);
    auto smem = [&](int nt) {  // K-partials, LayerNorm statistics, LayerNorm gamma | beta (see the kernel)
        return (size_t)nw * nt * 64 * sizeof(f32x4) + (size_t)nw * 16 * 2 * sizeof(float) + (p.ln_g ? (size_t)nw * KPW * 64 * sizeof(float) : 0);
    };
    if (p.ln_g) {
        if (wide)
            hipLaunchKernelGGL((dec_linear_kernel<TW, KPW, true, 2>), grid, block, smem(2), st, p);
        else
            hipLaunchKernelGGL((dec_linear_kernel<TW, KPW, true, 1>), grid, block, smem(1), st, p);
    } else if (sizeof(TW) == 2 && p.x_is_t) {  // activations already in operand dtype (attention output, MLP hidden)
        if (wide)
            hipLaunchKernelGGL((dec_linear_kernel<TW, KPW, false, 2, true>), grid, block, smem(2), st, p);
        else
            hipLaunchKernelGGL((dec_linear_kernel<TW, KPW, false, 1, true>), grid, block, smem(1), st, p);
    } else {
        if (wide)
            hipLaunchKernelGGL((dec_linear_kernel<TW, KPW, false, 2>), grid, block, smem(2), st, p);
        else
            hipLaunchKernelGGL((dec_linear_kernel<TW, KPW, false, 1>), grid, block, smem(1), st, p);
    }
    return WM_LAUNCH_OK;
}
// K % 32 == 0 and (K/32) must factor as NW * KPW with NW <= 16, KPW <= 4 (true for every K = 128·j, j <= 16).
static int dec_linear_waves(int K, int elem_bytes = 2) {
    if (K <= 0 || (K & 31)) return 0;
    const int ksteps = K >> 5;
    if (elem_bytes == 4) {  // fp32 operands (eight exact-fp32 MFMAs per k-step): one k-step per wave stays fastest (B = 1: 211 vs 223 us per step)
        static const int kpw32 = wm_env("WM_LIN_KPW32") ? atoi(wm_env("WM_LIN_KPW32")) : 0;  // dev A/B: k-steps per wave for fp32 operands
        if (kpw32 > 0) {
            for (int lim = kpw32; lim <= 4; ++lim)
                for (int c = 1; c <= 16; ++c)
                    if (ksteps % c == 0 && ksteps / c <= lim) return c;
        }
        for (int c = 16; c >= 1; --c)
            if (ksteps % c == 0 && ksteps / c <= 4) return c;
        return 0;
    }
    // the FEWEST waves whose k-steps-per-wave stay <= 3 (else <= 4): measured on MI355X (round 2, tiny B = 64) 4 waves x 3 k-steps
    // for K = 384 beat 12 waves x 1 (decode step 266 -> 257 us; pass alone 33.3 -> 32.4 ms) — a 256-thread workgroup is
    // dispatched sooner, exchanges 4 instead of 12 partials, and fits beside a K/V-streaming kernel of another pass; 6 x 2
    // (263 us), 3 x 4 (274 us) and 2 x 6 (363 us) lost
    static const int kpw_max = wm_env("WM_LIN_KPW") ? atoi(wm_env("WM_LIN_KPW")) : 3;
    for (int lim = kpw_max; lim <= 4; ++lim)
        for (int c = 1; c <= 16; ++c)
            if (ksteps % c == 0 && ksteps / c <= lim) return c;
    return 0;
}
bool dec_linear_supports_k(int K) { return dec_linear_waves(K) > 0; }
template <typename TW> int launch_dec_linear(const DecLinearParams& p, hipStream_t st) {
    if (p.B <= 0 || p.N <= 0) return launch_refuse("dec_linear: empty problem");
    const int ksteps = p.K >> 5;
    const int nw = dec_linear_waves(p.K, (int)sizeof(TW));
    if (nw == 0)  // refuse rather than drop k-steps (check_cfg / wm_op_matmul_nt keep model paths away from here)
        return launch_refuse("dec_linear: K must be a multiple of 32 whose k-steps split over <= 16 waves x <= 4 steps (K <= 2048)");
    switch (ksteps / nw) {
        case 1: return launch_dec_linear_t<TW, 1>(p, nw, st);
        case 2: return launch_dec_linear_t<TW, 2>(p, nw, st);
        case 3: return launch_dec_linear_t<TW, 3>(p, nw, st);
        default: return launch_dec_linear_t<TW, 4>(p, nw, st);
    }
}
template int launch_dec_linear<float>(const DecLinearParams&, hipStream_t);
template int launch_dec_linear<bf16>(const DecLinearParams&, hipStream_t);
template int launch_dec_linear<f16>(const DecLinearParams&, hipStream_t);

// Scratch of the logits epilogue (fused argmax, stage 1): static LDS in the 16-bit kernel; in the split-fp32 kernel it overlays the
// activation images once every wave is past its last fragment read.
template <int NRB> struct LogitsScratch {
    float (*av)[NRB * 16];  // [8 waves x 4 lane groups][utterance row]: the 4 lane groups of a row meet in LDS, not by cross-lane
    int (*ai)[NRB * 16];    // shuffles (two ds_bpermute round trips per row block on the kernel's tail)
    float (*tv)[NRB * 16];  // [8 waves][row]: timestamp-side partials (best value, max, sum of exp)
    float (*tm)[NRB * 16];
    float (*ts)[NRB * 16];
    int (*ti)[NRB * 16];
    static constexpr size_t bytes = (size_t)(2 * 32 + 4 * 8) * NRB * 16 * 4;
    __device__ static LogitsScratch carve(unsigned char* base) {
        LogitsScratch s;
        float* f = reinterpret_cast<float*>(base);
        s.av = reinterpret_cast<float (*)[NRB * 16]>(f);
        s.ai = reinterpret_cast<int (*)[NRB * 16]>(f + 32 * NRB * 16);
        s.tv = reinterpret_cast<float (*)[NRB * 16]>(f + 64 * NRB * 16);
        s.tm = reinterpret_cast<float (*)[NRB * 16]>(f + 72 * NRB * 16);
        s.ts = reinterpret_cast<float (*)[NRB * 16]>(f + 80 * NRB * 16);
        s.ti = reinterpret_cast<int (*)[NRB * 16]>(f + 88 * NRB * 16);
        return s;
    }
};
// Shared tail of the logits kernels.  acc[nb][rb][r] = logits[row0 + 16 rb + r16][n0[nb] + 4 g + r]: optional store of the
// logits, then the fused argmax's stage 1 (and the timestamp-side partials when the rules are on).
template <int NRB>
__device__ __forceinline__ void logits_epilogue(const DecLinearParams& p, int CT, const f32x4 (&acc)[2][NRB], const int (&n0)[2],
                                                const bool (&have)[2], const float (&mk)[2][4], int row0, int nrows, int lane, int w,
                                                const LogitsScratch<NRB>& sc) {
    const int r16 = lane & 15, g = lane >> 4;
    // acc[nb][rb][r] = logits[row0 + 16 rb + r16][n0[nb] + 4 g + r]
    if (p.out) {
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            const int lr = rb * 16 + r16;
            if (lr < nrows) {
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const int n = n0[nb] + g * 4;
                    if (have[nb] && n < p.ldo) *reinterpret_cast<f32x4*>(p.out + (size_t)(row0 + lr) * p.ldo + n) = acc[nb][rb];
                }
            }
        }
    }
    if (p.amax_val) {
        // fused argmax, stage 1 (whisper_tensor.mojo:431-439: lowest index wins): this workgroup's best of its CT*16
        // columns per utterance.  Lane-local over 8 columns, 2 butterfly steps over the 4 lanes of a row, LDS over waves.
        // Timestamp rules on (p.ts_state): "best" is the best admissible TEXT id of the utterance's range, and the workgroups
        // that cover timestamp ids also reduce the admissible timestamps to (best value, its id, max, sum of exp(v - max)) —
        // what stage 2 needs to compare the timestamps' probability mass with the best text id (HF rule 5).
        const bool ts_on = p.ts_state != nullptr;
        const bool wg_ts = ts_on && (int)((blockIdx.x + 1) * CT * 16) > p.ts_begin;  // workgroup-uniform
        float (*s_av)[NRB * 16] = sc.av;
        int (*s_ai)[NRB * 16] = sc.ai;
        float (*s_tv)[NRB * 16] = sc.tv, (*s_tm)[NRB * 16] = sc.tm, (*s_ts)[NRB * 16] = sc.ts;
        int (*s_ti)[NRB * 16] = sc.ti;
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            float bv = -INFINITY;
            int bi = 0x7fffffff;
            int t_lo = 0, t_hi = 0x7fffffff, q_lo = 0, q_hi = 0;
            if (ts_on) {
                const TsState* stp = p.ts_state + row0 + min(rb * 16 + r16, nrows - 1);
                t_lo = stp->text_lo;
                t_hi = stp->text_hi;
                q_lo = stp->ts_lo;
                q_hi = stp->ts_hi;
            }
            float cand[2][4];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0[nb] + g * 4 + r;
                    const float v = acc[nb][rb][r] + mk[nb][r];  // 0 or -inf
                    cand[nb][r] = v;
                    if (have[nb] && n < p.N && n >= t_lo && n < t_hi && v > bv) {  // n increases through the loop: strict '>' keeps the lowest index
                        bv = v;
                        bi = n;
                    }
                }
            if (wg_ts) {
                float tv = -INFINITY, tsum = 0.f;
                int ti = 0x7fffffff;
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int n = n0[nb] + g * 4 + r;
                        if (have[nb] && n < p.N && n >= q_lo && n < q_hi && cand[nb][r] > tv) {
                            tv = cand[nb][r];
                            ti = n;
                        }
                    }
                float tm = tv;  // the lane's max IS its best value
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int n = n0[nb] + g * 4 + r;
                        if (have[nb] && n < p.N && n >= q_lo && n < q_hi && cand[nb][r] > -INFINITY) tsum += expf(cand[nb][r] - tm);
                    }
#pragma unroll
                for (int o = 16; o <= 32; o <<= 1) {
                    const float v2 = __shfl_xor(tv, o, 64), m2 = __shfl_xor(tm, o, 64), s2 = __shfl_xor(tsum, o, 64);
                    const int i2 = __shfl_xor(ti, o, 64);
                    if (v2 > tv || (v2 == tv && i2 < ti)) {
                        tv = v2;
                        ti = i2;
                    }
                    const float mn = fmaxf(tm, m2);
                    tsum = (tsum > 0.f ? tsum * expf(tm - mn) : 0.f) + (s2 > 0.f ? s2 * expf(m2 - mn) : 0.f);
                    tm = mn;
                }
                if (g == 0) {
                    s_tv[w][rb * 16 + r16] = tv;
                    s_ti[w][rb * 16 + r16] = ti;
                    s_tm[w][rb * 16 + r16] = tm;
                    s_ts[w][rb * 16 + r16] = tsum;
                }
            }
            s_av[w * 4 + g][rb * 16 + r16] = bv;
            s_ai[w * 4 + g][rb * 16 + r16] = bi;
        }
        __syncthreads();
        if (threadIdx.x < NRB * 16 && (int)threadIdx.x < nrows) {
            float bv = s_av[0][threadIdx.x];
            int bi = s_ai[0][threadIdx.x];
#pragma unroll
            for (int k = 1; k < 8 * 4; ++k) {
                const float v2 = s_av[k][threadIdx.x];
                const int i2 = s_ai[k][threadIdx.x];
                if (v2 > bv || (v2 == bv && i2 < bi)) {
                    bv = v2;
                    bi = i2;
                }
            }
            const size_t o = (size_t)(row0 + threadIdx.x) * p.amax_stride + blockIdx.x;
            p.amax_val[o] = bv;
            p.amax_idx[o] = bi;
            if (wg_ts) {  // the 8 waves' timestamp partials, in wave order (deterministic)
                float tv = s_tv[0][threadIdx.x], tm = s_tm[0][threadIdx.x], tsum = s_ts[0][threadIdx.x];
                int ti = s_ti[0][threadIdx.x];
#pragma unroll
                for (int k = 1; k < 8; ++k) {
                    const float v2 = s_tv[k][threadIdx.x], m2 = s_tm[k][threadIdx.x], s2 = s_ts[k][threadIdx.x];
                    const int i2 = s_ti[k][threadIdx.x];
                    if (v2 > tv || (v2 == tv && i2 < ti)) {
                        tv = v2;
                        ti = i2;
                    }
                    const float mn = fmaxf(tm, m2);
                    tsum = (tsum > 0.f ? tsum * expf(tm - mn) : 0.f) + (s2 > 0.f ? s2 * expf(m2 - mn) : 0.f);
                    tm = mn;
                }
                p.ts_val[o] = tv;
                p.ts_idx[o] = ti;
                p.ts_m[o] = tm;
                p.ts_s[o] = tsum;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Final LayerNorm + tied-embedding logits (whisper.mojo:156-166): logits[B, V] = LN(x)[B, d] · tok_emb[V, d]ᵀ.
// 80 MB (fp32) / 40 MB (16-bit) of weights per step: the one decoder GEMM that is bandwidth- rather than
// latency-bound.  Workgroup = 128 vocabulary rows x up to 64 utterances: the normalised activations are built ONCE per
// workgroup (LN statistics from a 4-thread-per-row sweep, everything in flight at once) and parked in LDS as MFMA
// operands, so L2 sees x once per 128 columns; each wave then streams its 32 embedding rows straight from HBM into
// registers (all k-steps in flight) and needs no cross-wave reduction.
template <typename TW, int KD /* d_model/128 */, int NRB>
__global__ __launch_bounds__(512) void dec_logits_kernel(DecLinearParams p, int CT) {
    // One workgroup = 8 waves = CT <= 16 column tiles of 16 vocabulary rows (wave w owns tiles w and w+8), all NRB*16
    // utterance rows.  CT is chosen by the launcher so that the whole vocabulary is ONE round of <= 256 workgroups
    // (51 865 rows: CT = 13 -> 250 workgroups, 160 KB of embedding rows per CU): the kernel is a single pass over the
    // embedding at one workgroup per CU instead of 406 128-column workgroups in two rounds (16.5 -> ~10 us).
    constexpr int K = KD * 128;
    // row pitch of the LDS activation image: for 16-bit operands ≡ 40 dwords (mod 64) — with the hardware's ds_read_b128
    // lane groups ({0-3,12-15,20-27}, ...) the 16 fragment rows of one read then sit on disjoint banks (a 4-dword pad
    // left one 2-way conflict per group: 3 % of the kernel's wave cycles in SQ_LDS_BANK_CONFLICT)
    constexpr int PAD = sizeof(TW) == 2 ? 80 - (K % 128) : 16 / sizeof(TW);
    constexpr int PITCH = K + PAD;
    constexpr int KS = KD * 4;                         // k-steps of 32
    constexpr int CHK = sizeof(TW) == 2 ? KS : KS / 2;  // k-steps whose weight fragments are in flight together
#ifndef WM_LOGITS_PRE
#define WM_LOGITS_PRE 4
#endif
    constexpr int PRE = WM_LOGITS_PRE < CHK ? WM_LOGITS_PRE : CHK;  // k-steps of the first tile requested before the staging arithmetic
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    TW* xs = reinterpret_cast<TW*>(smem_raw);  // [NRB*16][PITCH]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.y * NRB * 16;
    if (p.ts && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) ts_put(p.ts, p.ts_id, 2);
#ifdef WM_DEV
#define WM_LG_STAMP(k) do { if (p.dbg && (threadIdx.x & 63) == 0) p.dbg[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (k)] = (long long)wall_clock64(); } while (0)
#else
#define WM_LG_STAMP(k) do { } while (0)
#endif
    WM_LG_STAMP(0);
    const int nrows = min(NRB * 16, p.B - row0);
    int n0[2];
    bool have[2];
    const TW* wp[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int t = w + 8 * nb;  // column tile inside this workgroup
        n0[nb] = (blockIdx.x * CT + t) * 16;
        have[nb] = __builtin_amdgcn_readfirstlane((int)(t < CT && n0[nb] < p.N)) != 0;  // wave-uniform, and the compiler knows
        int wr = n0[nb] + r16;
        wr = wr < p.N ? wr : p.N - 1;
        wp[nb] = (const TW*)p.W + (size_t)wr * K + g * 8;
    }
    // a wave without a second tile re-requests its first one (same lines: L1 / L2 hits, no extra HBM traffic) so that the loads
    // below are UNCONDITIONAL: with a branch around them hipcc's s_waitcnt bookkeeping can no longer count them and makes the
    // LayerNorm staging wait for every embedding row (vmcnt(11) … vmcnt(0) in front of the statistics: 5 us of the kernel)
    if (!have[1]) wp[1] = wp[0];
    // The activation rows are requested first; the embedding rows once they are here (see below), so the LayerNorm staging
    // runs while this wave's embedding rows stream from HBM.
    Frag<TW> wf[2][CHK];
    float mk[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};  // additive argmax mask (0 / -inf) of this lane's columns
    __shared__ __attribute__((aligned(16))) float s_gb[2][K];  // LN gamma / beta, fetched ahead of the embedding rows too
    {  // LN + convert -> LDS.  thread t: row t>>3 (+64 per pass), eighth t&7 of the row, float4 index q + 8*i
        for (int rbase = 0; rbase < NRB * 16; rbase += 64) {
            const int lr = rbase + (threadIdx.x >> 3), q = threadIdx.x & 7;
            f32x4 v[KD * 4];
            const bool mine = lr < NRB * 16;
            if (mine) {
                const int row = min(lr, nrows - 1);
                const float* xr = p.x + (size_t)(row0 + row) * p.ldx;
#pragma unroll
                for (int i = 0; i < KD * 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(xr + 4 * (q + 8 * i));
            }
            if (rbase == 0) {
                f32x4 gbv = f32x4{0.f, 0.f, 0.f, 0.f};
                const int gi = threadIdx.x;  // float4 index into [gamma | beta]
                if (gi < K / 2) gbv = *reinterpret_cast<const f32x4*>((gi < K / 4 ? p.ln_g : p.ln_b - K) + 4 * gi);
                if (p.amax_mask) {  // the argmax mask of this lane's 8 columns rides the same round trip (the epilogue used to fetch it)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) mk[nb][r] = p.amax_mask[min(n0[nb] + g * 4 + r, p.N - 1)];
                }
                if (gi < K / 2) *reinterpret_cast<f32x4*>(&s_gb[0][0] + 4 * gi) = gbv;
                WM_LG_STAMP(5);
                __syncthreads();
                WM_LG_STAMP(6);
                // The embedding rows are requested only now, when this workgroup's activation rows have arrived: issued together
                // with them (round 1) the 40 MB burst of all 250 workgroups filled the memory pipeline first and the 98 KB of
                // activations every workgroup needs — the same L2 lines for all of them — came back last: the LayerNorm staging
                // ended 7.8 us after kernel entry, 1.7 us before the MFMAs did (in-kernel stamps, tools/logits_phases.py).
#pragma unroll
                for (int i = 0; i < PRE; ++i) wf[0][i] = load_frag<TW>(wp[0] + i * 32);
            }
            if (mine) {
                float sm = 0.f, sq = 0.f;
#pragma unroll
                for (int i = 0; i < KD * 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        sm += v[i][j];
                        sq += v[i][j] * v[i][j];
                    }
#pragma unroll
                for (int o = 1; o <= 4; o <<= 1) {
                    sm += __shfl_xor(sm, o, 64);
                    sq += __shfl_xor(sq, o, 64);
                }
                const float mean = sm / (float)K;
                const float var = (sq / (float)K) - (mean * mean);
                const float rstd = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
                for (int i = 0; i < KD * 4; ++i) {
                    const int k = 4 * (q + 8 * i);
                    const f32x4 gm = *reinterpret_cast<const f32x4*>(&s_gb[0][k]), bt = *reinterpret_cast<const f32x4*>(&s_gb[1][k]);
                    typedef __attribute__((ext_vector_type(4))) TW t4;
                    t4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = from_f32<TW>((v[i][j] - mean) * rstd * gm[j] + bt[j]);
                    *reinterpret_cast<t4*>(&xs[lr * PITCH + k]) = o;
                }
            }
        }
    }
    WM_LG_STAMP(1);
    __syncthreads();
    WM_LG_STAMP(2);
    // the rest of the embedding rows, behind the staging arithmetic AND the barrier: a wave cannot issue past a vector load the
    // memory pipeline has no room for, so with all 24 KB per wave requested up front the staging stood behind the loads' ISSUE
    // for ~5 us (in-kernel stamps), and a barrier after the issue made every wave wait for the slowest issuer
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int i = (nb == 0 ? PRE : 0); i < CHK; ++i) wf[nb][i] = load_frag<TW>(wp[nb] + i * 32);

    f32x4 acc[2][NRB];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) acc[nb][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
    // straight-line MFMA bodies for "both tiles" / "first tile only" (have[1] implies have[0]); a per-MFMA predicate
    // would put a branch and a wait around every instruction
    auto body = [&](auto NB) {
        constexpr int nbn = decltype(NB)::value;
#pragma unroll
        for (int c = 0; c < KS / CHK; ++c) {
            if (c > 0) {
#pragma unroll
                for (int nb = 0; nb < nbn; ++nb)
#pragma unroll
                    for (int i = 0; i < CHK; ++i) wf[nb][i] = load_frag<TW>(wp[nb] + (c * CHK + i) * 32);
            }
#pragma unroll
            for (int i = 0; i < CHK; ++i) {
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) {
                    Frag<TW> xf = load_frag<TW>(&xs[(rb * 16 + r16) * PITCH + (c * CHK + i) * 32 + g * 8]);
#pragma unroll
                    for (int nb = 0; nb < nbn; ++nb) acc[nb][rb] = mma32(wf[nb][i], xf, acc[nb][rb]);
                }
            }
        }
    };
    if (have[1])
        body(std::integral_constant<int, 2>{});
    else if (have[0])
        body(std::integral_constant<int, 1>{});
    WM_LG_STAMP(3);
    if constexpr (NRB <= 4) {
        __shared__ float s_av[8 * 4][NRB * 16];
        __shared__ int s_ai[8 * 4][NRB * 16];
        __shared__ float s_tv[8][NRB * 16], s_tm[8][NRB * 16], s_ts[8][NRB * 16];
        __shared__ int s_ti[8][NRB * 16];
        LogitsScratch<NRB> sc;
        sc.av = s_av;
        sc.ai = s_ai;
        sc.tv = s_tv;
        sc.tm = s_tm;
        sc.ts = s_ts;
        sc.ti = s_ti;
        logits_epilogue<NRB>(p, CT, acc, n0, have, mk, row0, nrows, lane, w, sc);
    } else {  // 128 rows: the 48 KB of scratch do not fit beside the activation image — overlay it once every wave is done reading
        __syncthreads();
        static_assert(LogitsScratch<NRB>::bytes <= (size_t)NRB * 16 * PITCH * sizeof(TW), "epilogue scratch must fit the activation image");
        logits_epilogue<NRB>(p, CT, acc, n0, have, mk, row0, nrows, lane, w, LogitsScratch<NRB>::carve(smem_raw));
    }
    WM_LG_STAMP(4);
}
#undef WM_LG_STAMP
int dec_logits_tiles_per_wg(int N);
int dec_logits_parts(int N);
// ------------------------------------------------------------------------------------------------------------
// The same projection for FP32 weights (decoder_fp32 / all-fp32 models) on the bf16 MFMA pipe.
// With fp32 operands the kernel above is MFMA-bound, not HBM-bound: 2·64·51 865·384 = 2.55 GFLOP per step against the 157 TFLOP/s
// of v_mfma_f32_16x16x4_f32 (32 cycles per SIMD for a 16x16x4 block) is 16 us at best, 20.5 us for the busiest SIMD's four
// column tiles — measured 38 us per launch for an 80 MB stream that HBM delivers in 13.  gfx950 has no reduced-precision fp32
// path, but every fp32 value is EXACTLY the sum of three bf16 values (8 + 8 + 8 significant bits, taken by truncation:
// x = h + m + l with l = x - h - m exact), and a bf16 x bf16 product is exact in the MFMA's fp32 accumulator.  So
//     w·x = (wh + wm + wl)(xh + xm + xl) ≈ wh·xh + wh·xm + wm·xh + wh·xl + wm·xm + wl·xh
// — six v_mfma_f32_16x16x32_bf16 (16 cycles each, K = 32) instead of eight fp32 MFMAs of K = 4 (32 cycles each): 96 against 256
// cycles per 16x16x32 block.  The three dropped terms (wm·xl, wl·xm, wl·xl) are < 2^-21 |w·x| together in the worst case
// (truncation leaves |m| < 2^-7 |x|, |l| < 2^-15 |x|), 2^-24 typically; accumulation stays fp32.  Measured over 5 clips x 196
// positions x 51 865 logits: largest deviation from the fp32 CPU restatement 5.3e-6, against 6.3e-6 for the exact-fp32 MFMA kernel (tests/test_split3.py
// restates the arithmetic; tests/test_gpu_long_parity.py prints the figure).  The weights are split in registers as they arrive (44 VALU operations per fragment, hidden in
// the MFMAs' issue gaps); the normalised activations are split ONCE per workgroup and parked in LDS as three bf16 images.
struct Split3 {
    bf16x8 h, m, l;
};
__device__ __forceinline__ Split3 split3(const f32x8& x) {
    typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
    u16x8 h, m, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const unsigned u = __float_as_uint(x[j]);
        const unsigned uh = u & 0xffff0000u;  // top 8 significant bits (truncation keeps the sign: the remainders share it)
        const float r1 = x[j] - __uint_as_float(uh);  // exact
        const unsigned um = __float_as_uint(r1) & 0xffff0000u;
        const float r2 = r1 - __uint_as_float(um);  // exact, <= 8 significant bits left: a bf16 value
        h[j] = (unsigned short)(uh >> 16);
        m[j] = (unsigned short)(um >> 16);
        l[j] = (unsigned short)(__float_as_uint(r2) >> 16);
    }
    Split3 s;
    s.h = __builtin_bit_cast(bf16x8, h);
    s.m = __builtin_bit_cast(bf16x8, m);
    s.l = __builtin_bit_cast(bf16x8, l);
    return s;
}
__device__ __forceinline__ f32x4 mma_bf16(const bf16x8& a, const bf16x8& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

template <int KD /* d_model/128 */, int NRB>
__global__ __launch_bounds__(512) void dec_logits_split_kernel(DecLinearParams p, int CT) {
    // Geometry as dec_logits_kernel: 8 waves, wave w owns column tiles w and w + 8 of the workgroup's CT <= 16, all NRB*16 rows.
    constexpr int K = KD * 128;
    // row pitch of one bf16 activation image: ≡ 8 dwords (mod 64) — the 16 fragment rows of a ds_read_b128 lane group
    // ({rows 0-3, 12-15 at k-group g} ∪ {rows 4-11 at g + 1}) then sit on 16 disjoint 4-bank groups, like the ≡ 40 pitch above
    constexpr int PITCH = K + 16;
    constexpr int KS = KD * 4;      // k-steps of 32
    constexpr int CHK = KD == 1 ? 2 : KS / 2;  // k-steps of a tile whose fp32 weight fragments are in flight together (8 VGPRs each)
    constexpr int IMG = NRB * 16 * PITCH;  // elements per image
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* xs = reinterpret_cast<bf16*>(smem_raw);                                        // [3][NRB*16][PITCH]: h, m, l
    float* s_gb = reinterpret_cast<float*>(smem_raw + (size_t)3 * IMG * sizeof(bf16));  // [2][K] LN gamma / beta
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.y * NRB * 16;
    if (p.ts && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) ts_put(p.ts, p.ts_id, 2);
#ifdef WM_DEV
#define WM_LG_STAMP(k) do { if (p.dbg && (threadIdx.x & 63) == 0) p.dbg[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (k)] = (long long)wall_clock64(); } while (0)
#else
#define WM_LG_STAMP(k) do { } while (0)
#endif
    WM_LG_STAMP(0);
    const int nrows = min(NRB * 16, p.B - row0);
    int n0[2];
    bool have[2];
    const float* wp[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int t = w + 8 * nb;
        n0[nb] = (blockIdx.x * CT + t) * 16;
        have[nb] = __builtin_amdgcn_readfirstlane((int)(t < CT && n0[nb] < p.N)) != 0;
        int wr = n0[nb] + r16;
        wr = wr < p.N ? wr : p.N - 1;
        wp[nb] = (const float*)p.W + (size_t)wr * K + g * 8;
    }
    if (!have[1]) wp[1] = wp[0];  // unconditional loads (see dec_logits_kernel): a wave without a second tile re-requests its first
    f32x8 wf[2][CHK];
    auto wload = [&](int nb, int ks) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(wp[nb] + ks * 32), b = *reinterpret_cast<const f32x4*>(wp[nb] + ks * 32 + 4);
        return f32x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    };
    float mk[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    constexpr int PRE = CHK < 4 ? CHK : 4;  // k-steps of the first tile requested before the staging arithmetic
    {  // LN + three-way split -> LDS.  thread t: row t>>3 (+64 per pass), eighth t&7 of the row, float4 index q + 8*i
        for (int rbase = 0; rbase < NRB * 16; rbase += 64) {
            const int lr = rbase + (threadIdx.x >> 3), q = threadIdx.x & 7;
            f32x4 v[KD * 4];
            const bool mine = lr < NRB * 16;
            if (mine) {
                const int row = min(lr, nrows - 1);
                const float* xr = p.x + (size_t)(row0 + row) * p.ldx;
#pragma unroll
                for (int i = 0; i < KD * 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(xr + 4 * (q + 8 * i));
            }
            if (rbase == 0) {
                f32x4 gbv = f32x4{0.f, 0.f, 0.f, 0.f};
                const int gi = threadIdx.x;  // float4 index into [gamma | beta]
                if (gi < K / 2) gbv = *reinterpret_cast<const f32x4*>((gi < K / 4 ? p.ln_g : p.ln_b - K) + 4 * gi);
                if (p.amax_mask) {
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) mk[nb][r] = p.amax_mask[min(n0[nb] + g * 4 + r, p.N - 1)];
                }
                if (gi < K / 2) *reinterpret_cast<f32x4*>(s_gb + 4 * gi) = gbv;
                WM_LG_STAMP(5);
                __syncthreads();
                WM_LG_STAMP(6);
#pragma unroll
                for (int i = 0; i < PRE; ++i) wf[0][i] = wload(0, i);
            }
            if (mine) {
                float sm = 0.f, sq = 0.f;
#pragma unroll
                for (int i = 0; i < KD * 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        sm += v[i][j];
                        sq += v[i][j] * v[i][j];
                    }
#pragma unroll
                for (int o = 1; o <= 4; o <<= 1) {
                    sm += __shfl_xor(sm, o, 64);
                    sq += __shfl_xor(sq, o, 64);
                }
                const float mean = sm / (float)K;
                const float var = (sq / (float)K) - (mean * mean);
                const float rstd = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
                for (int i = 0; i < KD * 4; ++i) {
                    const int k = 4 * (q + 8 * i);
                    const f32x4 gm = *reinterpret_cast<const f32x4*>(s_gb + k), bt = *reinterpret_cast<const f32x4*>(s_gb + K + k);
                    typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
                    u16x4 oh, om, ol;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float y = (v[i][j] - mean) * rstd * gm[j] + bt[j];  // the fp32 operand the exact kernel multiplies
                        const unsigned uh = __float_as_uint(y) & 0xffff0000u;
                        const float r1 = y - __uint_as_float(uh);
                        const unsigned um = __float_as_uint(r1) & 0xffff0000u;
                        const float r2 = r1 - __uint_as_float(um);
                        oh[j] = (unsigned short)(uh >> 16);
                        om[j] = (unsigned short)(um >> 16);
                        ol[j] = (unsigned short)(__float_as_uint(r2) >> 16);
                    }
                    *reinterpret_cast<u16x4*>(&xs[lr * PITCH + k]) = oh;
                    *reinterpret_cast<u16x4*>(&xs[IMG + lr * PITCH + k]) = om;
                    *reinterpret_cast<u16x4*>(&xs[2 * IMG + lr * PITCH + k]) = ol;
                }
            }
        }
    }
    WM_LG_STAMP(1);
    __syncthreads();
    WM_LG_STAMP(2);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int i = (nb == 0 ? PRE : 0); i < CHK; ++i) wf[nb][i] = wload(nb, i);

    f32x4 acc[2][NRB];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) acc[nb][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto body = [&](auto NB) {
        constexpr int nbn = decltype(NB)::value;
#pragma unroll
        for (int c = 0; c < KS / CHK; ++c) {
#pragma unroll
            for (int i = 0; i < CHK; ++i) {
                Split3 ws[nbn];
#pragma unroll
                for (int nb = 0; nb < nbn; ++nb) {
                    ws[nb] = split3(wf[nb][i]);
                    // rolling prefetch: the register set just split takes the same k-step of the next chunk
                    if (c + 1 < KS / CHK) wf[nb][i] = wload(nb, (c + 1) * CHK + i);
                }
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) {
                    const int xo = (rb * 16 + r16) * PITCH + (c * CHK + i) * 32 + g * 8;
                    const bf16x8 xh = *reinterpret_cast<const bf16x8*>(&xs[xo]);
                    const bf16x8 xm = *reinterpret_cast<const bf16x8*>(&xs[IMG + xo]);
                    const bf16x8 xl = *reinterpret_cast<const bf16x8*>(&xs[2 * IMG + xo]);
#pragma unroll
                    for (int nb = 0; nb < nbn; ++nb) {  // smallest terms first
                        f32x4 a = acc[nb][rb];
                        a = mma_bf16(ws[nb].l, xh, a);
                        a = mma_bf16(ws[nb].h, xl, a);
                        a = mma_bf16(ws[nb].m, xm, a);
                        a = mma_bf16(ws[nb].m, xh, a);
                        a = mma_bf16(ws[nb].h, xm, a);
                        a = mma_bf16(ws[nb].h, xh, a);
                        acc[nb][rb] = a;
                    }
                }
            }
        }
    };
    if (have[1])
        body(std::integral_constant<int, 2>{});
    else if (have[0])
        body(std::integral_constant<int, 1>{});
    WM_LG_STAMP(3);
    __syncthreads();  // every wave is past its last fragment read: the epilogue's scratch overlays the activation images
    static_assert(LogitsScratch<NRB>::bytes <= (size_t)3 * IMG * sizeof(bf16), "epilogue scratch must fit the activation images");
    logits_epilogue<NRB>(p, CT, acc, n0, have, mk, row0, nrows, lane, w, LogitsScratch<NRB>::carve(smem_raw));
    WM_LG_STAMP(4);
}
#undef WM_LG_STAMP
// The same kernel for 128 rows per workgroup — two coalesced 64-utterance batches on one decode state (wm_config.coalesce): with 64
// rows per workgroup a 128-row state streamed the 80 MB embedding twice per step.  Three bf16 images of 128 x 384 do not fit the
// 160 KB of LDS, so K goes in TWO passes of K/2 columns: stage the first half of every row, multiply it with k-steps 0 .. K/64-1 of
// the workgroup's embedding rows (the second half's weight fragments are requested meanwhile, register set by register set), barrier,
// re-stage, multiply the second half into the same accumulators.  Every row's arithmetic is exactly the 64-row kernel's — the
// LayerNorm statistics are summed in the same order (a thread carries the two partial sums of the two "virtual" threads the
// 8-threads-per-row scheme gives its elements to), the k-steps accumulate in the same order — so the ids of a coalesced pass equal
// the uncoalesced ones bit for bit.
template <int KD /* d_model/128 */>
__global__ __launch_bounds__(512) void dec_logits_split128_kernel(DecLinearParams p, int CT) {
    constexpr int NRB = 8, ROWS = 128;
    constexpr int K = KD * 128, KH = K / 2;  // columns per pass
    constexpr int PITCH = KH + 16;           // ≡ 8 dwords (mod 16): conflict-free fragment reads (see dec_logits_split_kernel)
    constexpr int KSH = KD * 2;              // k-steps of 32 per pass
    constexpr int IMG = ROWS * PITCH;        // elements per image
    constexpr int F4 = K / 16;               // float4 per thread and row (4 threads per row); the first F4/2 belong to pass 0
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* xs = reinterpret_cast<bf16*>(smem_raw);                                        // [3][128][PITCH]: h, m, l of the current K half
    float* s_gb = reinterpret_cast<float*>(smem_raw + (size_t)3 * IMG * sizeof(bf16));  // [2][K] LN gamma / beta
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.y * ROWS;
    if (p.ts && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) ts_put(p.ts, p.ts_id, 2);
#ifdef WM_DEV  // phase stamps (wm_bench_kernel id 40): 0 entry, 1 first half staged, 2 past its barrier, 5 first half multiplied, 6 second half staged, 3 all multiplied, 4 end
#define WM_LG_STAMP(k) do { if (p.dbg && (threadIdx.x & 63) == 0) p.dbg[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (k)] = (long long)wall_clock64(); } while (0)
#else
#define WM_LG_STAMP(k) do { } while (0)
#endif
    WM_LG_STAMP(0);
    const int nrows = min(ROWS, p.B - row0);
    int n0[2];
    bool have[2];
    const float* wp[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int t = w + 8 * nb;
        n0[nb] = (blockIdx.x * CT + t) * 16;
        have[nb] = __builtin_amdgcn_readfirstlane((int)(t < CT && n0[nb] < p.N)) != 0;
        int wr = n0[nb] + r16;
        wr = wr < p.N ? wr : p.N - 1;
        wp[nb] = (const float*)p.W + (size_t)wr * K + g * 8;
    }
    if (!have[1]) wp[1] = wp[0];
    constexpr int NSL = 3 < 2 * KSH ? 3 : 2 * KSH;  // weight-fragment register sets per tile: k-step kk lives in set kk % NSL and, once
    f32x8 wf[2][NSL];                                // split, the set is refilled with k-step kk + NSL (across the pass boundary too)
    auto wload = [&](int nb, int ks) __attribute__((always_inline)) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(wp[nb] + ks * 32), b = *reinterpret_cast<const f32x4*>(wp[nb] + ks * 32 + 4);
        return f32x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    };
    float mk[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    // staging role: row lr = thread / 4, quarter q4 = thread % 4; the thread owns float4 indices q4 + 4 j (j < F4) of its row
    const int lr = threadIdx.x >> 2, q4 = threadIdx.x & 3;
    const float* xr = p.x + (size_t)(row0 + min(lr, nrows - 1)) * p.ldx;
    float mean, rstd;
    auto stage = [&](const f32x4 (&v)[F4 / 2], int pass) __attribute__((always_inline)) {  // normalise + split the pass's F4/2 float4 of this thread -> LDS
#pragma unroll
        for (int j = 0; j < F4 / 2; ++j) {
            const int k = 4 * (q4 + 4 * (pass * (F4 / 2) + j));  // column of the row
            const f32x4 gm = *reinterpret_cast<const f32x4*>(s_gb + k), bt = *reinterpret_cast<const f32x4*>(s_gb + K + k);
            typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
            u16x4 oh, om, ol;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float y = (v[j][e] - mean) * rstd * gm[e] + bt[e];
                const unsigned uh = __float_as_uint(y) & 0xffff0000u;
                const float r1 = y - __uint_as_float(uh);
                const unsigned um = __float_as_uint(r1) & 0xffff0000u;
                const float r2 = r1 - __uint_as_float(um);
                oh[e] = (unsigned short)(uh >> 16);
                om[e] = (unsigned short)(um >> 16);
                ol[e] = (unsigned short)(__float_as_uint(r2) >> 16);
            }
            const int kc = k - pass * KH;  // column inside the pass's image
            *reinterpret_cast<u16x4*>(&xs[lr * PITCH + kc]) = oh;
            *reinterpret_cast<u16x4*>(&xs[IMG + lr * PITCH + kc]) = om;
            *reinterpret_cast<u16x4*>(&xs[2 * IMG + lr * PITCH + kc]) = ol;
        }
    };
    {  // pass 0: the whole row for the statistics, its first half staged
        f32x4 v0[F4 / 2], v1[F4 / 2];
#pragma unroll
        for (int j = 0; j < F4 / 2; ++j) {
            v0[j] = *reinterpret_cast<const f32x4*>(xr + 4 * (q4 + 4 * j));
            v1[j] = *reinterpret_cast<const f32x4*>(xr + 4 * (q4 + 4 * (F4 / 2 + j)));
        }
        f32x4 gbv = f32x4{0.f, 0.f, 0.f, 0.f};
        const int gi = threadIdx.x;  // float4 index into [gamma | beta]
        if (gi < K / 2) gbv = *reinterpret_cast<const f32x4*>((gi < K / 4 ? p.ln_g : p.ln_b - K) + 4 * gi);
        if (p.amax_mask) {
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) mk[nb][r] = p.amax_mask[min(n0[nb] + g * 4 + r, p.N - 1)];
        }
        if (gi < K / 2) *reinterpret_cast<f32x4*>(s_gb + 4 * gi) = gbv;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NSL; ++i) wf[0][i] = wload(0, i);
        // one-pass statistics in the order of the 8-threads-per-row scheme: virtual thread q owns float4 q + 8 i.  This thread's
        // float4 j is index q4 + 4 j: even j -> virtual thread q4 (i = j / 2), odd j -> virtual thread q4 + 4.
        float sa = 0.f, qa = 0.f, sb = 0.f, qb = 0.f;
        static_assert((F4 / 2) % 2 == 0, "the parity of a float4's index is the same in both halves");
        auto accum = [&](const f32x4 (&v)[F4 / 2]) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < F4 / 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (j & 1) {
                        sb += v[j][e];
                        qb += v[j][e] * v[j][e];
                    } else {
                        sa += v[j][e];
                        qa += v[j][e] * v[j][e];
                    }
                }
        };
        accum(v0);
        accum(v1);
#pragma unroll
        for (int o = 1; o <= 2; o <<= 1) {  // the butterfly's first two levels, inside each half of the 8 virtual threads
            sa += __shfl_xor(sa, o, 64);
            qa += __shfl_xor(qa, o, 64);
            sb += __shfl_xor(sb, o, 64);
            qb += __shfl_xor(qb, o, 64);
        }
        const float sm = sa + sb, sq = qa + qb;  // its last level
        mean = sm / (float)K;
        const float var = (sq / (float)K) - (mean * mean);
        rstd = 1.0f / sqrtf(var + 1e-5f);
        stage(v0, 0);
    }
    WM_LG_STAMP(1);
    __syncthreads();
    WM_LG_STAMP(2);
#pragma unroll
    for (int i = 0; i < NSL; ++i) wf[1][i] = wload(1, i);
    f32x4 acc[2][NRB];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) acc[nb][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto restage = [&]() __attribute__((always_inline)) {  // every wave, tiles or not: the second K half of the rows
        __syncthreads();  // all fragment reads of the first half are done
        f32x4 v1[F4 / 2];
#pragma unroll
        for (int j = 0; j < F4 / 2; ++j) v1[j] = *reinterpret_cast<const f32x4*>(xr + 4 * (q4 + 4 * (F4 / 2 + j)));
        stage(v1, 1);
        __syncthreads();
    };
    auto run = [&](auto NB) __attribute__((always_inline)) {
        constexpr int nbn = decltype(NB)::value;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1) {
                WM_LG_STAMP(5);
                restage();
                WM_LG_STAMP(6);
            }
            if constexpr (nbn > 0) {
#pragma unroll
                for (int i = 0; i < KSH; ++i) {
                    Split3 ws[nbn];
#pragma unroll
                    for (int nb = 0; nb < nbn; ++nb) {
                        constexpr int dummy = 0;
                        (void)dummy;
                        const int kk = pass * KSH + i;  // compile-time after unrolling
                        ws[nb] = split3(wf[nb][kk % NSL]);
                        if (kk + NSL < 2 * KSH) wf[nb][kk % NSL] = wload(nb, kk + NSL);  // rolling: the set just split takes k-step kk + NSL
                    }
                    // Row blocks in pairs, the six products of a k-step interleaved over the pair's (and both tiles') accumulators: 128
                    // rows make this kernel MFMA-bound, and six back-to-back MFMAs on ONE accumulator wait for each other's results.
                    // Per accumulator the order of the six terms is the 64-row kernel's (smallest first).
#pragma unroll
                    for (int rp = 0; rp < NRB; rp += 2) {
                        bf16x8 xh[2], xm[2], xl[2];
#pragma unroll
                        for (int r = 0; r < 2; ++r) {
                            const int xo = ((rp + r) * 16 + r16) * PITCH + i * 32 + g * 8;
                            xh[r] = *reinterpret_cast<const bf16x8*>(&xs[xo]);
                            xm[r] = *reinterpret_cast<const bf16x8*>(&xs[IMG + xo]);
                            xl[r] = *reinterpret_cast<const bf16x8*>(&xs[2 * IMG + xo]);
                        }
#define WM_TERM(WPART, XPART)                                                                            \
    _Pragma("unroll") for (int r = 0; r < 2; ++r) _Pragma("unroll") for (int nb = 0; nb < nbn; ++nb)     \
        acc[nb][rp + r] = mma_bf16(ws[nb].WPART, XPART[r], acc[nb][rp + r]);
                        WM_TERM(l, xh)
                        WM_TERM(h, xl)
                        WM_TERM(m, xm)
                        WM_TERM(m, xh)
                        WM_TERM(h, xm)
                        WM_TERM(h, xh)
#undef WM_TERM
                    }
                }
            }
        }
    };
    if (have[1])
        run(std::integral_constant<int, 2>{});
    else if (have[0])
        run(std::integral_constant<int, 1>{});
    else
        run(std::integral_constant<int, 0>{});
    WM_LG_STAMP(3);
    __syncthreads();  // the epilogue's scratch overlays the activation images
    static_assert(LogitsScratch<NRB>::bytes <= (size_t)3 * IMG * sizeof(bf16), "epilogue scratch must fit the activation images");
    logits_epilogue<NRB>(p, CT, acc, n0, have, mk, row0, nrows, lane, w, LogitsScratch<NRB>::carve(smem_raw));
    WM_LG_STAMP(4);
}
#undef WM_LG_STAMP
template <int KD> static int launch_dec_logits_split128_t(const DecLinearParams& p, hipStream_t st) {
    const size_t lds = (size_t)3 * 128 * (KD * 64 + 16) * sizeof(bf16) + (size_t)2 * KD * 128 * sizeof(float);
    if (lds > 48 * 1024)
        if (const hipError_t e = ensure_dyn_lds<&dec_logits_split128_kernel<KD>>((int)lds); e != hipSuccess)
            return launch_hip_failed("logits kernel (split fp32, 128 rows): dynamic LDS attribute", e);
    const int ct = dec_logits_tiles_per_wg(p.N);
    dim3 grid(dec_logits_parts(p.N), (p.B + 127) / 128);
    hipLaunchKernelGGL((dec_logits_split128_kernel<KD>), grid, dim3(512), lds, st, p, ct);
    return WM_LAUNCH_OK;
}
template <int KD, int NRB> static int launch_dec_logits_split_t(const DecLinearParams& p, hipStream_t st) {
    const size_t lds = (size_t)3 * NRB * 16 * (KD * 128 + 16) * sizeof(bf16) + (size_t)2 * KD * 128 * sizeof(float);
    if (lds > 48 * 1024)
        if (const hipError_t e = ensure_dyn_lds<&dec_logits_split_kernel<KD, NRB>>((int)lds); e != hipSuccess)
            return launch_hip_failed("logits kernel (split fp32): dynamic LDS attribute", e);
    const int ct = dec_logits_tiles_per_wg(p.N);
    dim3 grid(dec_logits_parts(p.N), (p.B + NRB * 16 - 1) / (NRB * 16));
    hipLaunchKernelGGL((dec_logits_split_kernel<KD, NRB>), grid, dim3(512), lds, st, p, ct);
    return WM_LAUNCH_OK;
}

// column tiles (of 16) per workgroup: the smallest count that covers the vocabulary with <= 256 workgroups, at most 16
int dec_logits_tiles_per_wg(int N) {
    const int tiles = (N + 15) / 16;
    // (developer A/B WM_LOGITS_CT: 14 / 16 tiles per workgroup — every wave two tiles at 16 — measured the same launch time as 13:
    //  the stream's end is set by the chip-wide rate, not by the busiest wave)
    static const int ct_env = wm_env("WM_LOGITS_CT") ? atoi(wm_env("WM_LOGITS_CT")) : 0;
    if (ct_env > 0 && tiles > 16) return std::max((tiles + 255) / 256, std::min(16, ct_env));
    return std::max(1, std::min(16, (tiles + 255) / 256));
}
int dec_logits_ids_per_part(int N) { return dec_logits_tiles_per_wg(N) * 16; }
int dec_logits_parts(int N) {
    const int tiles = (N + 15) / 16, ct = dec_logits_tiles_per_wg(N);
    return (tiles + ct - 1) / ct;
}
template <typename TW, int KD, int NRB> static int launch_dec_logits_t(const DecLinearParams& p, hipStream_t st) {
    const size_t lds = (size_t)NRB * 16 * (KD * 128 + (sizeof(TW) == 2 ? 80 : 16 / sizeof(TW))) * sizeof(TW);
    if (lds > 48 * 1024)
        if (const hipError_t e = ensure_dyn_lds<&dec_logits_kernel<TW, KD, NRB>>((int)lds); e != hipSuccess)
            return launch_hip_failed("logits kernel: dynamic LDS attribute", e);
    const int ct = dec_logits_tiles_per_wg(p.N);
    dim3 grid(dec_logits_parts(p.N), (p.B + NRB * 16 - 1) / (NRB * 16));
    hipLaunchKernelGGL((dec_logits_kernel<TW, KD, NRB>), grid, dim3(512), lds, st, p, ct);
    return WM_LAUNCH_OK;
}
// requires ln_g/ln_b, no bias/act/residual, K in {128, 384, 512}, ldo % 4 == 0; amax_stride >= dec_logits_parts(N)
template <typename TW> int launch_dec_logits(const DecLinearParams& p, hipStream_t st) {
    const int kd = p.K >> 7;
    if (p.B <= 0 || p.N <= 0) return launch_refuse("dec_logits: empty problem");
    if ((p.K & 127) != 0 || (kd != 1 && kd != 3 && kd != 4)) return launch_refuse("dec_logits: d_model must be 128, 384 or 512");
    if (!p.ln_g || !p.ln_b) return launch_refuse("dec_logits: the final LayerNorm's gamma / beta are required");
    if (p.amax_val && p.amax_stride < dec_logits_parts(p.N)) return launch_refuse("dec_logits: amax_stride is smaller than the partials per utterance");
    if constexpr (sizeof(TW) == 4) {
        // fp32 weights: the three-way bf16 split (d_model 128 / 384: three activation images fit the 160 KB of LDS); d_model 512
        // keeps the exact-fp32 MFMA kernel.  WM_LOGITS_EXACT (developer build) forces the exact kernel for A/B runs.
        static const bool exact = wm_env("WM_LOGITS_EXACT") != nullptr;
        if (!exact && (kd == 1 || kd == 3)) {
            if (p.B <= 16) return kd == 1 ? launch_dec_logits_split_t<1, 1>(p, st) : launch_dec_logits_split_t<3, 1>(p, st);
            // more than 64 rows (coalesced batches): 128 rows per workgroup, the embedding streamed once per 128 (WM_LOGITS_NO128: A/B)
            static const bool no128 = wm_env("WM_LOGITS_NO128") != nullptr;
            if (p.B > 64 && !no128) return kd == 1 ? launch_dec_logits_split128_t<1>(p, st) : launch_dec_logits_split128_t<3>(p, st);
            return kd == 1 ? launch_dec_logits_split_t<1, 4>(p, st) : launch_dec_logits_split_t<3, 4>(p, st);
        }
    }
    if (p.B <= 16) {
        if (kd == 1) return launch_dec_logits_t<TW, 1, 1>(p, st);
        if (kd == 3) return launch_dec_logits_t<TW, 3, 1>(p, st);
        return launch_dec_logits_t<TW, 4, 1>(p, st);
    }
#ifdef WM_DEV
    if constexpr (sizeof(TW) == 2) {
        // Developer A/B (WM_LOGITS_128_16BIT): 128 rows per workgroup for 16-bit weights too.  Measured SLOWER than two 64-row
        // launches (tiny bf16, coalesced pairs: 16.5 vs 15.8 ms per pass; base f16: equal): the second launch's 40 MB embedding
        // stream is served by the Infinity Cache, so there is no HBM traffic to save, and the 128-row form stages and reduces twice
        // the rows behind one stream.  The fp32 form (80 MB per stream, MFMA-heavy) does gain: dec_logits_split128_kernel.
        static const bool on128 = wm_env("WM_LOGITS_128_16BIT") != nullptr;
        if (p.B > 64 && on128) {
            if (kd == 1) return launch_dec_logits_t<TW, 1, 8>(p, st);
            if (kd == 3) return launch_dec_logits_t<TW, 3, 8>(p, st);
            return launch_dec_logits_t<TW, 4, 8>(p, st);
        }
    }
#endif
    if (kd == 1) return launch_dec_logits_t<TW, 1, 4>(p, st);
    if (kd == 3) return launch_dec_logits_t<TW, 3, 4>(p, st);
    return launch_dec_logits_t<TW, 4, 4>(p, st);
}
template int launch_dec_logits<float>(const DecLinearParams&, hipStream_t);
template int launch_dec_logits<bf16>(const DecLinearParams&, hipStream_t);
template int launch_dec_logits<f16>(const DecLinearParams&, hipStream_t);

// ------------------------------------------------------------------------------------------------------------
// Single-query attention over the KV cache (layers.mojo:186-272), all heads of one utterance per workgroup so that
// whole token-major cache rows (H*64 elements, the reference's layout: layers.mojo:140-147) stream fully coalesced.
// grid = (nsplit key chunks, B).  Lane map: LPH lanes (16 B each) cover one head's 64 dims of one key; LPR = H*LPH
// lanes cover a row; the block sweeps RPS rows per step, U steps per iteration.
// ONE pass: K and V rows of an iteration are requested together (and the next iteration's before this one is
// consumed), scores reduce inside an aligned LPH-lane group with DPP only, and the softmax is the online form
// (running max / sum per lane, one rescale per iteration) — every K and V byte is read once, nothing is parked in LDS
// except the final merge of the RPS row slots.  Emits un-normalised partials (o, max, sum) per chunk for
// attn_combine, or the normalised output directly when the chunk is the whole sequence.
// Scale after the dot product and max initialised to -1e10 follow layers.mojo:196,212 (the mask branch at :213 is a
// no-op for j <= len-1 and is omitted).
template <typename TKV, int LPH, bool FAST, bool NT, int U, int NQ = 1>
__global__ __launch_bounds__(512) void attn_decode_kernel(AttnDecParams p) {
    // NQ > 1 (prompt prefill, cross-attention): the workgroup of utterance b serves the NQ query rows t * B + b from ONE
    // sweep of that utterance's K/V chunk — each cache row is read once for all prompt positions.
    constexpr int EPL = 64 / LPH;  // elements per lane
    __shared__ float s_ml[512][2];
    __shared__ float s_red[512 * EPL];
    const int b = blockIdx.y, split = blockIdx.x;
    if (p.ts && NT && b == 0 && split == 0 && threadIdx.x == 0) ts_put(p.ts, p.ts_id, 0);
    const int LPR = p.H * LPH;
    const int RPS = p.rps;  // key rows swept per step = active threads / LPR
    const int bk = (NQ == 1 && p.q_B > 0) ? b % p.q_B : b;  // utterance whose K/V this workgroup reads
    const int len = p.n_keys >= 0 ? p.n_keys : p.ctl->len + 1 + ((NQ == 1 && p.q_B > 0) ? b / p.q_B : 0);  // (scalar chain kernarg -> ctl -> len: the loop bounds need it before any K/V row is requested)
    const int qstride = NQ > 1 ? p.q_B : 0;  // query / output row of position t: b + t * qstride
    const int chunk = (len + p.nsplit - 1) / p.nsplit;
    const int j0 = split * chunk;
    const int j1 = min(len, j0 + chunk);
    const int rslot = threadIdx.x / LPR, c = threadIdx.x % LPR;
    const int h = c / LPH, e0 = (c % LPH) * EPL;
    const bool active = rslot < RPS;

    float qv[NQ][EPL];
#pragma unroll
    for (int t = 0; t < NQ; ++t)
#pragma unroll
        for (int e = 0; e < EPL; ++e) qv[t][e] = p.q[(size_t)(b + t * qstride) * p.d + h * 64 + e0 + e] * p.scale;
    const TKV* Kb = (const TKV*)p.K + (size_t)bk * p.batch_stride + h * 64 + e0;
    const TKV* Vb = (const TKV*)p.V + (size_t)bk * p.batch_stride + h * 64 + e0;
    typedef __attribute__((ext_vector_type(EPL))) TKV kvec;

    float m_run[NQ], l_run[NQ];
    float acc[NQ][EPL];
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
        m_run[t] = -1e10f;
        l_run[t] = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[t][e] = 0.f;
    }
    const int step = RPS * U;

    auto load = [&](kvec (&kk)[U], kvec (&vv)[U], int j) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int jj = max(min(j + u * RPS, j1 - 1), 0);
            if (NT) {  // read-once stream: keep it out of L2 / Infinity Cache so the weights stay resident there
                kk[u] = __builtin_nontemporal_load(reinterpret_cast<const kvec*>(Kb + (size_t)jj * p.d));
                vv[u] = __builtin_nontemporal_load(reinterpret_cast<const kvec*>(Vb + (size_t)jj * p.d));
            } else {
                kk[u] = *reinterpret_cast<const kvec*>(Kb + (size_t)jj * p.d);
                vv[u] = *reinterpret_cast<const kvec*>(Vb + (size_t)jj * p.d);
            }
        }
    };
    auto consume = [&](const kvec (&kk)[U], const kvec (&vv)[U], int j) {
#pragma unroll
        for (int t = 0; t < NQ; ++t) {
            float sc[U];
            float bm = -1e30f;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float dot = 0.f;
#pragma unroll
                for (int e = 0; e < EPL; ++e) dot += qv[t][e] * (float)kk[u][e];
                dot = LPH == 16 ? group_sum16(dot) : group_sum8(dot);
                sc[u] = (active && j + u * RPS < j1) ? dot : -1e30f;
                bm = fmaxf(bm, sc[u]);
            }
            const float m_new = fmaxf(m_run[t], bm);
            const float alpha = FAST ? __expf(m_run[t] - m_new) : expf(m_run[t] - m_new);
            float ps = 0.f;
#pragma unroll
            for (int e = 0; e < EPL; ++e) acc[t][e] *= alpha;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float pe = FAST ? __expf(sc[u] - m_new) : expf(sc[u] - m_new);
                ps += pe;
#pragma unroll
                for (int e = 0; e < EPL; ++e) acc[t][e] += pe * (float)vv[u][e];
            }
            l_run[t] = l_run[t] * alpha + ps;
            m_run[t] = m_new;
        }
    };

    {
        kvec ka[U], va[U], kb[U], vb[U];
        int j = j0 + rslot;
        if (j0 < j1) {
            load(ka, va, j);
            while (true) {
                const bool more1 = j + step < j1 + RPS;  // some row slot still has keys in the next iteration
                if (more1) load(kb, vb, j + step);
                consume(ka, va, j);
                j += step;
                if (!more1) break;
                const bool more2 = j + step < j1 + RPS;
                if (more2) load(ka, va, j + step);
                consume(kb, vb, j);
                j += step;
                if (!more2) break;
            }
        }
    }
    // merge the RPS row slots (lanes of one LPH group carry identical m, l), one query position after the other
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
        if (t > 0) __syncthreads();  // everyone is done reading the previous position's partials
        s_ml[threadIdx.x][0] = m_run[t];
        s_ml[threadIdx.x][1] = l_run[t];
#pragma unroll
        for (int e = 0; e < EPL; ++e) s_red[threadIdx.x * EPL + e] = acc[t][e];
        __syncthreads();
        if (rslot == 0) {
            float M = -1e10f;
            for (int r = 0; r < RPS; ++r) M = fmaxf(M, s_ml[r * LPR + c][0]);
            float L = 0.f, o[EPL];
#pragma unroll
            for (int e = 0; e < EPL; ++e) o[e] = 0.f;
            for (int r = 0; r < RPS; ++r) {
                const int tt = r * LPR + c;
                const float wgt = s_ml[tt][1] > 0.f ? (FAST ? __expf(s_ml[tt][0] - M) : expf(s_ml[tt][0] - M)) : 0.f;
                L += wgt * s_ml[tt][1];
#pragma unroll
                for (int e = 0; e < EPL; ++e) o[e] += wgt * s_red[tt * EPL + e];
            }
            const size_t orow = (size_t)b + (size_t)t * qstride;
            if (p.direct_out) {
                const float norm = 1.0f / L;
#pragma unroll
                for (int e = 0; e < EPL; ++e) store_as(p.direct_out, orow * p.d + h * 64 + e0 + e, o[e] * norm, p.out_dtype);
            } else {
                float* po = p.part_o + (orow * p.nsplit + split) * p.d + h * 64 + e0;
#pragma unroll
                for (int e = 0; e < EPL; ++e) po[e] = o[e];
            }
            if (!p.direct_out && (c % LPH) == 0) {
                float* pml = p.part_ml + ((orow * p.nsplit + split) * p.H + h) * 2;
                pml[0] = M;
                pml[1] = L;
            }
        }
    }
}
template <typename TKV> int launch_attn_decode(const AttnDecParams& p, hipStream_t st) {
    constexpr int LPH = sizeof(TKV) == 4 ? 16 : 8;
    constexpr bool FAST = sizeof(TKV) == 2;
    const int LPR = p.H * LPH;
    if (p.B <= 0 || p.H <= 0 || LPR > 256 || p.d != p.H * 64) return launch_refuse("attn_decode: needs 1 <= heads <= 16 of 64 dims (d = 64 heads)");
    if (p.nsplit < 1 || (!p.direct_out && (!p.part_o || !p.part_ml)) || (p.direct_out && p.nsplit != 1))
        return launch_refuse("attn_decode: partial buffers / direct output do not match nsplit");
    AttnDecParams q = p;
    // rows swept per step.  16-bit K/V: 256 threads (512 measured no better for the short self-attention: 5.3 vs 5.0 us).
    // fp32 K/V rows are twice as wide (96-128 lanes per row): the latency-bound self-attention takes 512 threads so a
    // lane's serial key loop stays short (61 keys: 16 -> 7 iterations)
    q.rps = ((sizeof(TKV) == 4 && p.n_keys < 0) ? 512 : 256) / LPR;
    static const int thr_self = wm_env("WM_SELF_THREADS") ? atoi(wm_env("WM_SELF_THREADS")) : 0;  // A/B: self-attention block size
    if (thr_self && p.n_keys < 0) q.rps = std::max(1, std::min(512, thr_self) / LPR);
    static const int thr_cross = wm_env("WM_ATTN_THREADS") ? atoi(wm_env("WM_ATTN_THREADS")) : 0;  // A/B: cross-attention block size
    if (thr_cross && p.n_keys >= 0) q.rps = std::max(1, std::min(512, thr_cross) / LPR);
    // block rounded up to whole waves: the spare lanes take no rows (rslot >= RPS) but stay in the DPP groups
    const dim3 grid(p.nsplit, p.B), block((q.rps * LPR + 63) / 64 * 64);
    static const bool nt_off = wm_env("WM_NO_NT") != nullptr;
    static const int u_cross = wm_env("WM_ATTN_U") ? atoi(wm_env("WM_ATTN_U")) : 4;
    if (p.n_keys >= 0 && p.nq == 4) {  // prompt prefill, four positions per utterance from one K/V sweep (q_B = utterances)
        const dim3 grid4(p.nsplit, p.q_B);
        hipLaunchKernelGGL((attn_decode_kernel<TKV, LPH, FAST, true, 4, 4>), grid4, block, 0, st, q);  // U = 4 as in the step kernel: same arithmetic per query
        return WM_LAUNCH_OK;
    }
    if (p.n_keys >= 0 && !nt_off) {  // the cross-attention K/V stream (1500 rows per utterance, read once per step)
        // p.lds_pad: unused dynamic LDS that caps the workgroups per CU (160 KB / (20 KB static + pad)) — see AttnDecParams
        static const int pad_env = wm_env("WM_ATTN_LDS_PAD") ? atoi(wm_env("WM_ATTN_LDS_PAD")) : -1;  // dev A/B override
        const int lds_pad = pad_env >= 0 ? pad_env : p.lds_pad;
        if (lds_pad > 40 * 1024)
            if (const hipError_t e = ensure_dyn_lds<&attn_decode_kernel<TKV, LPH, FAST, true, 4>>(lds_pad); e != hipSuccess)
                return launch_hip_failed("attn_decode: dynamic LDS attribute", e);
#ifdef WM_DEV
        if (u_cross == 8) {
            hipLaunchKernelGGL((attn_decode_kernel<TKV, LPH, FAST, true, 8>), grid, block, 0, st, q);
            return WM_LAUNCH_OK;
        } else if (u_cross == 2) {
            hipLaunchKernelGGL((attn_decode_kernel<TKV, LPH, FAST, true, 2>), grid, block, 0, st, q);
            return WM_LAUNCH_OK;
        }
#endif
        (void)u_cross;
        hipLaunchKernelGGL((attn_decode_kernel<TKV, LPH, FAST, true, 4>), grid, block, lds_pad, st, q);
    } else {
        // self-attention: 4 rows per lane per iteration (measured per 64-clip pass alone: U = 1 / 2 34.6 ms, U = 4 34.3 ms — the
        // serial iteration count matters as the cache grows to 104 rows).  WM_SELF_U=2 for A/B.
#ifdef WM_DEV
        static const int u_self = wm_env("WM_SELF_U") ? atoi(wm_env("WM_SELF_U")) : 4;
        if (u_self == 2) {
            hipLaunchKernelGGL((attn_decode_kernel<TKV, LPH, FAST, false, 2>), grid, block, 0, st, q);
            return WM_LAUNCH_OK;
        }
#endif
        hipLaunchKernelGGL((attn_decode_kernel<TKV, LPH, FAST, false, 4>), grid, block, 0, st, q);
    }
    return WM_LAUNCH_OK;
}
template int launch_attn_decode<float>(const AttnDecParams&, hipStream_t);
template int launch_attn_decode<bf16>(const AttnDecParams&, hipStream_t);
template int launch_attn_decode<f16>(const AttnDecParams&, hipStream_t);

// merge the key-chunk partials: out[b][h*64+e] = Σ_s w_s·o_s / Σ_s w_s·l_s,  w_s = exp(m_s − max m).
// One wave per (utterance, head): lane s owns chunk s's (m, l) — one exp per chunk, not per element — and the
// weights reach the 64 output lanes by wave broadcast.  nsplit <= 64.
__global__ void attn_combine_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml,
                                    void* __restrict__ out, int out_dtype, int nsplit, int H, int d, long long* ts, int ts_id) {
    const int b = blockIdx.x, lane = threadIdx.x & 63, h = threadIdx.x >> 6;
    if (ts && b == 0 && threadIdx.x == 0) ts_put(ts, ts_id, 1);
    float m = -1e30f, l = 0.f;
    if (lane < nsplit) {
        const float* ml = part_ml + (((size_t)b * nsplit + lane) * H + h) * 2;
        m = ml[0];
        l = ml[1];
    }
    const float M = wave_max(l > 0.f ? m : -1e30f);
    const float wgt = l > 0.f ? expf(m - M) : 0.f;
    const float L = wave_sum(wgt * l);
    const float* po = part_o + (size_t)b * nsplit * d + h * 64 + lane;
    float o = 0.f;
    int s = 0;
    for (; s + 8 <= nsplit; s += 8) {  // 8 independent loads in flight
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = po[(size_t)(s + j) * d];
#pragma unroll
        for (int j = 0; j < 8; ++j) o += __shfl(wgt, s + j, 64) * v[j];
    }
    for (; s < nsplit; ++s) o += __shfl(wgt, s, 64) * po[(size_t)s * d];
    store_as(out, (size_t)b * d + h * 64 + lane, o * (1.0f / L), out_dtype);
}
void launch_attn_combine(const float* part_o, const float* part_ml, void* out, int out_dtype, int B, int nsplit, int H, int d,
                         hipStream_t st, long long* ts, int ts_id) {
    hipLaunchKernelGGL(attn_combine_kernel, dim3(B), dim3(64 * H), 0, st, part_o, part_ml, out, out_dtype, nsplit, H, d, ts, ts_id);
}

// ------------------------------------------------------------------------------------------------------------
// argmax with the reference's tie rule (strict '>' scanning upward => lowest index wins, whisper_tensor.mojo:436)
// + the greedy loop's bookkeeping (whisper.mojo:200-221): append the id, stop an utterance after its eot.
// block-wide (value, index) argmax of row[0..V): 16-byte loads (row 16-byte aligned, V rounded up inside the padded
// row), every load independent; ties resolve to the LOWEST index at every level.
__device__ __forceinline__ void argmax_block(const float* row, int V, float& best, int& bidx) {
    __shared__ float s_v[16];
    __shared__ int s_i[16];
    float mv = -INFINITY;
    int mi = 0x7fffffff;
    const int nv = V >> 2;
    for (int i = threadIdx.x; i < nv; i += blockDim.x) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(row + 4 * i);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (v[j] > mv) {  // increasing index within a thread: strict '>' keeps the lowest
                mv = v[j];
                mi = 4 * i + j;
            }
    }
    for (int i = (nv << 2) + threadIdx.x; i < V; i += blockDim.x) {
        const float v = row[i];
        if (v > mv || (v == mv && i < mi)) {
            mv = v;
            mi = i;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float v2 = __shfl_xor(mv, o, 64);
        const int i2 = __shfl_xor(mi, o, 64);
        if (v2 > mv || (v2 == mv && i2 < mi)) {
            mv = v2;
            mi = i2;
        }
    }
    const int wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_v[wid] = mv;
        s_i[wid] = mi;
    }
    __syncthreads();
    mv = s_v[0];
    mi = s_i[0];
    for (int k = 1; k < nw; ++k)
        if (s_v[k] > mv || (s_v[k] == mv && s_i[k] < mi)) {
            mv = s_v[k];
            mi = s_i[k];
        }
    best = mv;
    bidx = mi;
}
// stage 2 of the fused argmax: best of the per-workgroup partials of one utterance (ties -> lowest index)
__device__ __forceinline__ void argmax_partials(const float* pv, const int* pi, int n, float& best, int& bidx) {
    __shared__ float s_v[16];
    __shared__ int s_i[16];
    float mv = -INFINITY;
    int mi = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float v = pv[i];
        const int ix = pi[i];
        if (v > mv || (v == mv && ix < mi)) {
            mv = v;
            mi = ix;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float v2 = __shfl_xor(mv, o, 64);
        const int i2 = __shfl_xor(mi, o, 64);
        if (v2 > mv || (v2 == mv && i2 < mi)) {
            mv = v2;
            mi = i2;
        }
    }
    const int wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_v[wid] = mv;
        s_i[wid] = mi;
    }
    __syncthreads();
    mv = s_v[0];
    mi = s_i[0];
    for (int k = 1; k < nw; ++k)
        if (s_v[k] > mv || (s_v[k] == mv && s_i[k] < mi)) {
            mv = s_v[k];
            mi = s_i[k];
        }
    best = mv;
    bidx = mi;
}
__global__ __launch_bounds__(1024) void argmax_step_kernel(ArgmaxParams p) {
    const int b = blockIdx.x;
    if (p.ts && b == 0 && threadIdx.x == 0) ts_put(p.ts, p.ts_id, 3);
    const int pos0 = p.emb_out ? p.pos[b] : 0;  // read before thread 0 advances it (the reductions below synchronise)
    if (p.host_progress && b == 0 && threadIdx.x == 0) {
        // finishes of EARLIER steps are complete (their launches ended); this step's may or may not be counted yet: a lower bound.
        // System-scope stores to pinned host memory: visible to the polling host without any stream synchronisation.
        const int fin = __hip_atomic_load(&p.ctl->n_finished, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&p.host_progress[0], fin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&p.host_progress[1], p.ctl->len, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    float best;
    int idx;
    if (p.pval)
        argmax_partials(p.pval + (size_t)b * p.npart, p.pidx + (size_t)b * p.npart, p.npart, best, idx);
    else
        argmax_block(p.logits + (size_t)b * p.ldl, p.V, best, idx);
    if (p.ts_state) {
        // Timestamp rules, the decision (HF WhisperTimeStampLogitsProcessor rule 5 + the final argmax): (best, idx) is the best
        // admissible text id; if the admissible timestamps' probability mass exceeds it — logsumexp(ts) > best, the softmax
        // normaliser being common — the id is the best admissible timestamp.  (Without that the argmax over both sets would
        // still be a text id: logsumexp(ts) >= max(ts).)  Then the history and the NEXT argmax's ranges.
        __shared__ int s_pick;
        if (threadIdx.x == 0) {
            float tv = -INFINITY, tm = -INFINITY, tsum = 0.f;
            int ti = 0x7fffffff;
            for (int k = p.ts_part0; k < p.npart; ++k) {  // <= a dozen parts, fixed order
                const size_t o = (size_t)b * p.npart + k;
                const float v2 = p.ts_val[o], m2 = p.ts_m[o], s2 = p.ts_s[o];
                const int i2 = p.ts_idx[o];
                if (v2 > tv || (v2 == tv && i2 < ti)) {
                    tv = v2;
                    ti = i2;
                }
                const float mn = fmaxf(tm, m2);
                tsum = (tsum > 0.f ? tsum * expf(tm - mn) : 0.f) + (s2 > 0.f ? s2 * expf(m2 - mn) : 0.f);
                tm = mn;
            }
            int pick = idx;
            const bool have_text = (unsigned)idx < (unsigned)p.V;
            if (tsum > 0.f && (!have_text || tm + logf(tsum) > best)) pick = ti;
            if ((unsigned)pick >= (unsigned)p.V) pick = 0;
            TsState st = p.ts_state[b];
            const int is_ts = pick >= p.rules.tb;
            st.pen_ts = st.n_gen + 1 < 2 ? 1 : st.last_ts;
            st.last_ts = is_ts;
            st.n_gen += 1;
            if (is_ts) st.t_last = pick;
            ts_next_ranges(st, p.rules);
            p.ts_state[b] = st;
            s_pick = pick;
        }
        __syncthreads();
        idx = s_pick;
    }
    // every candidate NaN / -inf (16-bit overflow, bad weights, a mask over the whole vocabulary): no comparison succeeded and
    // idx is still the sentinel — it must not become an embedding row index.  The reference's scan (whisper_tensor.mojo:431-439:
    // start at t[0], strict '>') returns index 0 in exactly these cases, and so does this.
    if ((unsigned)idx >= (unsigned)p.V) idx = 0;
    if (threadIdx.x == 0) {
        p.next[b] = idx;
        if (p.advance) {  // current_len += 1 (layers.mojo:143), position += 1: nothing else in this launch reads them
            p.pos[b] += 1;
            if (b == 0) p.ctl->len += 1;
        }
        if (p.out_tokens && !p.finished[b]) {
            p.out_tokens[(size_t)b * p.out_stride + p.n_tokens[b]] = idx;
            p.n_tokens[b] += 1;
            if (!p.ignore_eot && idx == p.eot) {
                p.finished[b] = 1;
                atomicAdd(&p.ctl->n_finished, 1);
            }
        }
    }
    if (p.emb_out) {  // every thread holds idx: next step's embedding row (whisper.mojo:141-149)
        const float* te = p.emb_tok + (size_t)idx * p.d;
        const float* pe = p.emb_pos + (size_t)min(pos0 + (p.advance ? 1 : 0), p.max_pos) * p.d;
        for (int j = threadIdx.x; j < p.d; j += blockDim.x) p.emb_out[(size_t)b * p.d + j] = te[j] + pe[j];
    }
}
void launch_argmax_step(const ArgmaxParams& p, hipStream_t st) {
    hipLaunchKernelGGL(argmax_step_kernel, dim3(p.B), dim3(p.pval ? 256 : 1024), 0, st, p);
}
__global__ __launch_bounds__(1024) void argmax_plain_kernel(const float* t, int n, int* idx) {
    float best;
    int i;
    argmax_block(t, n, best, i);
    if ((unsigned)i >= (unsigned)n) i = 0;  // all-NaN / all -inf input: the reference's scan (whisper_tensor.mojo:431-439) returns 0
    if (threadIdx.x == 0) *idx = i;
}
void launch_argmax_plain(const float* t, int n, int* idx, hipStream_t st) {
    hipLaunchKernelGGL(argmax_plain_kernel, dim3(1), dim3(1024), 0, st, t, n, idx);
}

// start of a greedy run: tokens = prompt (whisper.mojo:187-191, 200-202), nothing finished, control block zeroed
__global__ void init_tokens_kernel(InitTokensParams p) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < p.B) {
        for (int i = 0; i < p.n_prompt; ++i) p.out_tokens[(size_t)b * p.out_stride + i] = p.prompt[i];
        p.n_tokens[b] = p.n_prompt;
        p.finished[b] = 0;
        if (p.tok_rows) {
            for (int i = 0; i < p.n_prompt; ++i) {
                p.tok_rows[i * p.B + b] = p.prompt[i];
                p.pos_rows[i * p.B + b] = i;
            }
        }
        if (p.ts_state) {
            TsState st;
            st.n_gen = 0;
            st.last_ts = 0;
            st.pen_ts = 1;
            st.t_last = -1;
            ts_next_ranges(st, p.rules);
            p.ts_state[b] = st;
        }
    }
    if (b == 0) {
        p.ctl->len = 0;
        p.ctl->n_finished = 0;
    }
}
void launch_init_tokens(const InitTokensParams& p, hipStream_t st) {
    hipLaunchKernelGGL(init_tokens_kernel, dim3((p.B + 255) / 256), dim3(256), 0, st, p);
}

__global__ void pack_tokens_kernel(const int* __restrict__ out_tokens, const int* __restrict__ n_tokens, int out_stride, int rows, int stride,
                                   int* __restrict__ dst) {
    const int r = blockIdx.x;
    int* d = dst + (size_t)r * (1 + stride);
    const int n = r < rows ? min(n_tokens[r], stride) : 0;
    if (threadIdx.x == 0) d[0] = n;
    for (int j = threadIdx.x; j < stride; j += blockDim.x) d[1 + j] = j < n ? out_tokens[(size_t)r * out_stride + j] : 0;
}
void launch_pack_tokens(const int* out_tokens, const int* n_tokens, int out_stride, int rows, int rows_cap, int stride, int* dst, hipStream_t st) {
    hipLaunchKernelGGL(pack_tokens_kernel, dim3(rows_cap), dim3(256), 0, st, out_tokens, n_tokens, out_stride, rows, stride, dst);
}

__global__ void set_step_kernel(StepCtl* ctl, int len, int set_len, int* pos, int pos_value, int* tok, int tok_value, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) {
        if (pos) pos[i] = pos_value;
        if (tok) tok[i] = tok_value;
    }
    if (i == 0 && set_len) ctl->len = len;
}
void launch_set_step(StepCtl* ctl, int len, int set_len, int* pos, int pos_value, int* tok, int tok_value, int B,
                     hipStream_t st) {
    hipLaunchKernelGGL(set_step_kernel, dim3((B + 255) / 256), dim3(256), 0, st, ctl, len, set_len, pos, pos_value, tok,
                       tok_value, B);
}

// ------------------------------------------------------------------------------------------------------------
// op-level helpers (known-answer tests of the C-ABI)
__global__ void gelu_kernel(float* t, size_t n, int mode) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) t[i] = gelu_f(t[i], mode);
}
void launch_gelu(float* t, size_t n, int mode, hipStream_t st) {
    // the reference leaves the tail t.size % width untouched (whisper_tensor.mojo:308); width = 8 (x86 AVX2 lanes)
    size_t nb = (n / 8) * 8;
    if (nb) hipLaunchKernelGGL(gelu_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st, t, nb, mode);
}
// 3-pass row softmax (whisper_tensor.mojo:311-355), one wave per row
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* t, int rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* r = t + (size_t)row * cols;
    float mx = -INFINITY;
    for (int j = lane; j < cols; j += 64) mx = fmaxf(mx, r[j]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int j = lane; j < cols; j += 64) {
        float e = expf(r[j] - mx);
        r[j] = e;
        s += e;
    }
    s = wave_sum(s);
    for (int j = lane; j < cols; j += 64) r[j] = r[j] / s;
}
void launch_softmax_rows(float* t, int rows, int cols, hipStream_t st) {
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, t, rows, cols);
}
template <typename T> __global__ void convert_kernel(const float* in, T* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = from_f32<T>(in[i]);
}
template <typename T> void launch_convert(const float* in, void* out, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(convert_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, (T*)out, n);
}
template void launch_convert<float>(const float*, void*, size_t, hipStream_t);
template void launch_convert<bf16>(const float*, void*, size_t, hipStream_t);
template void launch_convert<f16>(const float*, void*, size_t, hipStream_t);
__global__ void transpose_f32_kernel(const float* in, float* out, int rows, int cols) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (size_t)rows * cols) {
        int r = (int)(i / cols), c = (int)(i % cols);
        out[(size_t)c * rows + r] = in[i];
    }
}
void launch_transpose_f32(const float* in, float* out, int rows, int cols, hipStream_t st) {
    size_t n = (size_t)rows * cols;
    hipLaunchKernelGGL(transpose_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out, rows, cols);
}

}  // namespace wm
