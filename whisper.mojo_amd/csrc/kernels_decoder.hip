// kernels_decoder.hip — the HBM-bound half of the path: one KV-cached decode step for B utterances.
//
// Replaces (reference file:line): the M<=4 branch of matmul (whisper_tensor.mojo:158-175) for every decoder
// projection incl. the tied 51 865-way logits (whisper.mojo:162-166); layer_norm (whisper_tensor.mojo:249-285) fused
// as a prologue; gelu (:288-308) and the residual adds (layers.mojo:457-461,483-487,513-517) fused as epilogues;
// the KV-cache append (layers.mojo:140-147); the q_len==1 attention path (layers.mojo:186-272); the embedding +
// position add (whisper.mojo:141-149); argmax (whisper_tensor.mojo:431-439).
#include "wm_kernels.h"

namespace wm {

// ------------------------------------------------------------------------------------------------------------
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
// Skinny linear: out[B,N] = epi( pro(x)[B,K] · W[N,K]ᵀ + bias ).  Every weight byte is read from HBM exactly once
// per step, all B rows sharing the pass — the batched form of the reference's "parallel over n, dot over K" GEMV.
//
// Workgroup = 16 output columns; its 4 waves split K (wave w takes k-steps w, w+4, …) and each wave sweeps all
// row blocks of 16 utterances against the ONE weight fragment it loaded (MFMA 16x16; exact fp32 or 16-bit operands).
// Computed as outᵀ so a lane owns 4 consecutive columns of one utterance; the 4 K-partials meet in LDS, then wave w
// finishes row block w (bias / GELU / residual / KV-cache scatter).
// LayerNorm prologue: the row statistics (one-pass variance, as the reference) are computed by the workgroup for all
// B rows and applied while the activation fragments are formed, so the normalised vector never exists in memory.
template <typename TW>
__global__ __launch_bounds__(256) void dec_linear_kernel(DecLinearParams p) {
    __shared__ float s_mean[64], s_rstd[64];
    __shared__ __attribute__((aligned(16))) f32x4 s_red[4][4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16;
    int wrow = n0 + r16;
    wrow = wrow < p.N ? wrow : p.N - 1;
    const TW* wp = (const TW*)p.W + (size_t)wrow * p.K + g * 8;
    const int ksteps = p.K >> 5;
    const int nrb = (p.B + 15) >> 4;

    for (int rb0 = 0; rb0 < nrb; rb0 += 4) {
        const int rows_here = min(64, p.B - rb0 * 16);
        if (p.ln_g) {
            __syncthreads();
            for (int r = w; r < rows_here; r += 4) {
                const float* xr = p.x + (size_t)(rb0 * 16 + r) * p.ldx;
                float s = 0.f, q = 0.f;
                for (int k = lane; k < p.K; k += 64) {
                    float v = xr[k];
                    s += v;
                    q += v * v;
                }
                s = wave_sum(s);
                q = wave_sum(q);
                const float mean = s / (float)p.K;
                const float var = (q / (float)p.K) - (mean * mean);
                if (lane == 0) {
                    s_mean[r] = mean;
                    s_rstd[r] = 1.0f / sqrtf(var + 1e-5f);
                }
            }
            __syncthreads();
        }
        f32x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int nrb_here = min(4, nrb - rb0);
        for (int ks = w; ks < ksteps; ks += 4) {
            const int k = ks * 32 + g * 8;
            Frag<TW> wf = load_frag<TW>(wp + ks * 32);
            float gam[8], bet[8];
            if (p.ln_g) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    gam[j] = p.ln_g[k + j];
                    bet[j] = p.ln_b[k + j];
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i < nrb_here) {
                    int lr = i * 16 + r16;  // row within this pass
                    lr = lr < rows_here ? lr : rows_here - 1;
                    const float* xr = p.x + (size_t)(rb0 * 16 + lr) * p.ldx + k;
                    f32x4 a = *reinterpret_cast<const f32x4*>(xr), b = *reinterpret_cast<const f32x4*>(xr + 4);
                    float xv[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
                    if (p.ln_g) {
                        const float mean = s_mean[lr], rstd = s_rstd[lr];
#pragma unroll
                        for (int j = 0; j < 8; ++j) xv[j] = (xv[j] - mean) * rstd * gam[j] + bet[j];
                    }
                    Frag<TW> xf = make_frag<TW>(xv);
                    acc[i] = mma32(wf, xf, acc[i]);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) s_red[w][i][lane] = acc[i];
        __syncthreads();
        if (w < nrb_here) {
            f32x4 v = s_red[0][w][lane] + s_red[1][w][lane] + s_red[2][w][lane] + s_red[3][w][lane];
            const int lr = w * 16 + r16;
            if (lr < rows_here) {
                const int b = rb0 * 16 + lr;
                const int n = n0 + g * 4;
                if (p.bias) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += p.bias[min(n + r, p.N - 1)];
                }
                if (p.act) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = gelu_f(v[r], p.gelu_mode);
                }
                if (p.residual) v += *reinterpret_cast<const f32x4*>(p.residual + (size_t)b * p.ldr + n);
                if (p.kcache && n >= p.d_model) {
                    const bool is_v = n >= 2 * p.d_model;
                    const int c = n - (is_v ? 2 : 1) * p.d_model;
                    const size_t off = (size_t)b * p.kv_batch_stride + (size_t)p.ctl->len * p.d_model + c;
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
                } else {
                    *reinterpret_cast<f32x4*>(p.out + (size_t)b * p.ldo + n) = v;
                }
            }
        }
    }
}
template <typename TW> void launch_dec_linear(const DecLinearParams& p, hipStream_t st) {
    hipLaunchKernelGGL(dec_linear_kernel<TW>, dim3((p.N + 15) / 16), dim3(256), 0, st, p);
}
template void launch_dec_linear<float>(const DecLinearParams&, hipStream_t);
template void launch_dec_linear<bf16>(const DecLinearParams&, hipStream_t);
template void launch_dec_linear<f16>(const DecLinearParams&, hipStream_t);

// ------------------------------------------------------------------------------------------------------------
// Single-query attention over the KV cache (layers.mojo:186-272), all heads of one utterance per workgroup so that
// whole token-major cache rows (H*64 elements, the reference's layout: layers.mojo:140-147) stream fully coalesced.
// grid = (nsplit key chunks, B).  Lane map: LPH lanes (16 B each) cover one head's 64 dims of one key; LPR = H*LPH
// lanes cover a row; the block sweeps RPS rows per step.  Dot products reduce inside an aligned LPH-lane group with
// DPP only.  Two passes over the chunk (K, then V) with scores parked in LDS; every K and V byte is read once.
// Emits un-normalised partials (o, max, sum) per chunk; attn_combine merges chunks.
// Scale after the dot product and max initialised to -1e10 follow layers.mojo:196,212 (the mask branch at :213 is a
// no-op for j <= len-1 and is omitted).
template <typename TKV, int LPH>
__global__ __launch_bounds__(256) void attn_decode_kernel(AttnDecParams p) {
    constexpr int EPL = 64 / LPH;  // elements per lane
    __shared__ float s_scores[512 * 8];
    __shared__ float s_m[8], s_l[8];
    __shared__ float s_red[256 * EPL];
    const int b = blockIdx.y, split = blockIdx.x;
    const int LPR = p.H * LPH;
    const int RPS = blockDim.x / LPR;
    const int len = p.n_keys >= 0 ? p.n_keys : p.ctl->len + 1;
    const int chunk = (len + p.nsplit - 1) / p.nsplit;
    const int j0 = split * chunk;
    const int j1 = min(len, j0 + chunk);
    const int rslot = threadIdx.x / LPR, c = threadIdx.x % LPR;
    const int h = c / LPH, e0 = (c % LPH) * EPL;
    const bool active = rslot < RPS;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nwaves = (blockDim.x + 63) >> 6;

    float qv[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) qv[e] = p.q[(size_t)b * p.d + h * 64 + e0 + e];
    const TKV* Kb = (const TKV*)p.K + (size_t)b * p.batch_stride + h * 64 + e0;
    const TKV* Vb = (const TKV*)p.V + (size_t)b * p.batch_stride + h * 64 + e0;
    typedef __attribute__((ext_vector_type(EPL))) TKV kvec;

    // pass 1: scores
    constexpr int U = 4;
    for (int j = j0 + rslot; j < j1; j += RPS * U) {
        kvec kv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int jj = min(j + u * RPS, j1 - 1);
            kv[u] = *reinterpret_cast<const kvec*>(Kb + (size_t)jj * p.d);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float dot = 0.f;
#pragma unroll
            for (int e = 0; e < EPL; ++e) dot += qv[e] * (float)kv[u][e];
            dot = LPH == 16 ? group_sum16(dot) : group_sum8(dot);
            const int jj = j + u * RPS;
            if (active && jj < j1 && (c % LPH) == 0) s_scores[(jj - j0) * p.H + h] = dot * p.scale;
        }
    }
    __syncthreads();
    // per-head max / exp / sum over the chunk
    const int nk = j1 - j0;
    for (int hh = wid; hh < p.H; hh += nwaves) {
        float mx = -1e10f;
        for (int j = lane; j < nk; j += 64) mx = fmaxf(mx, s_scores[j * p.H + hh]);
        mx = wave_max(mx);
        float sum = 0.f;
        for (int j = lane; j < nk; j += 64) {
            float e = expf(s_scores[j * p.H + hh] - mx);
            s_scores[j * p.H + hh] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        if (lane == 0) {
            s_m[hh] = mx;
            s_l[hh] = sum;
        }
    }
    __syncthreads();
    // pass 2: weighted sum of V
    float acc[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[e] = 0.f;
    for (int j = j0 + rslot; j < j1; j += RPS * U) {
        kvec vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int jj = min(j + u * RPS, j1 - 1);
            vv[u] = *reinterpret_cast<const kvec*>(Vb + (size_t)jj * p.d);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int jj = j + u * RPS;
            const float pj = (active && jj < j1) ? s_scores[(jj - j0) * p.H + h] : 0.f;
#pragma unroll
            for (int e = 0; e < EPL; ++e) acc[e] += pj * (float)vv[u][e];
        }
    }
#pragma unroll
    for (int e = 0; e < EPL; ++e) s_red[threadIdx.x * EPL + e] = active ? acc[e] : 0.f;
    __syncthreads();
    if (rslot == 0) {
        float* po = p.direct_out ? p.direct_out + (size_t)b * p.d + h * 64 + e0
                                 : p.part_o + ((size_t)b * p.nsplit + split) * p.d + h * 64 + e0;
        const float norm = p.direct_out ? 1.0f / s_l[h] : 1.0f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            float v = 0.f;
            for (int r = 0; r < RPS; ++r) v += s_red[(r * LPR + c) * EPL + e];
            po[e] = v * norm;
        }
        if (!p.direct_out && (c % LPH) == 0) {
            float* pml = p.part_ml + (((size_t)b * p.nsplit + split) * p.H + h) * 2;
            pml[0] = s_m[h];
            pml[1] = s_l[h];
        }
    }
}
template <typename TKV> void launch_attn_decode(const AttnDecParams& p, hipStream_t st) {
    constexpr int LPH = sizeof(TKV) == 4 ? 16 : 8;
    const int LPR = p.H * LPH;
    const int RPS = 256 / LPR;
    // block rounded up to whole waves: the spare lanes take no rows (rslot >= RPS) but join the wave-wide reductions
    hipLaunchKernelGGL((attn_decode_kernel<TKV, LPH>), dim3(p.nsplit, p.B), dim3((RPS * LPR + 63) / 64 * 64), 0, st, p);
}
template void launch_attn_decode<float>(const AttnDecParams&, hipStream_t);
template void launch_attn_decode<bf16>(const AttnDecParams&, hipStream_t);
template void launch_attn_decode<f16>(const AttnDecParams&, hipStream_t);

// merge the key-chunk partials: out[b][h*64+e] = Σ_s w_s·o_s / Σ_s w_s·l_s,  w_s = exp(m_s − max m)
__global__ void attn_combine_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml,
                                    float* __restrict__ out, int nsplit, int H, int d) {
    const int b = blockIdx.x;
    for (int t = threadIdx.x; t < d; t += blockDim.x) {
        const int h = t >> 6;
        float M = -1e30f;
        for (int s = 0; s < nsplit; ++s) M = fmaxf(M, part_ml[(((size_t)b * nsplit + s) * H + h) * 2]);
        float L = 0.f, o = 0.f;
        for (int s = 0; s < nsplit; ++s) {
            const float* ml = part_ml + (((size_t)b * nsplit + s) * H + h) * 2;
            const float wgt = ml[1] > 0.f ? expf(ml[0] - M) : 0.f;
            L += wgt * ml[1];
            o += wgt * part_o[((size_t)b * nsplit + s) * d + t];
        }
        out[(size_t)b * d + t] = o * (1.0f / L);
    }
}
void launch_attn_combine(const float* part_o, const float* part_ml, float* out, int B, int nsplit, int H, int d,
                         hipStream_t st) {
    hipLaunchKernelGGL(attn_combine_kernel, dim3(B), dim3(d < 256 ? 128 : 256), 0, st, part_o, part_ml, out, nsplit, H, d);
}

// ------------------------------------------------------------------------------------------------------------
// argmax with the reference's tie rule (strict '>' scanning upward => lowest index wins, whisper_tensor.mojo:436)
// + the greedy loop's bookkeeping (whisper.mojo:200-221): append the id, stop an utterance after its eot.
__device__ __forceinline__ void argmax_block(const float* row, int V, float& best, int& bidx) {
    __shared__ float s_v[256];
    __shared__ int s_i[256];
    float mv = -INFINITY;
    int mi = 0x7fffffff;
    for (int i = threadIdx.x; i < V; i += blockDim.x) {
        float v = row[i];
        if (v > mv || (v == mv && i < mi)) {
            mv = v;
            mi = i;
        }
    }
    s_v[threadIdx.x] = mv;
    s_i[threadIdx.x] = mi;
    __syncthreads();
    for (int o = blockDim.x >> 1; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            float v2 = s_v[threadIdx.x + o];
            int i2 = s_i[threadIdx.x + o];
            if (v2 > s_v[threadIdx.x] || (v2 == s_v[threadIdx.x] && i2 < s_i[threadIdx.x])) {
                s_v[threadIdx.x] = v2;
                s_i[threadIdx.x] = i2;
            }
        }
        __syncthreads();
    }
    best = s_v[0];
    bidx = s_i[0];
}
__global__ __launch_bounds__(256) void argmax_step_kernel(ArgmaxParams p) {
    const int b = blockIdx.x;
    float best;
    int idx;
    argmax_block(p.logits + (size_t)b * p.ldl, p.V, best, idx);
    if (threadIdx.x == 0) {
        p.next[b] = idx;
        if (p.out_tokens && !p.finished[b]) {
            p.out_tokens[(size_t)b * p.out_stride + p.n_tokens[b]] = idx;
            p.n_tokens[b] += 1;
            if (!p.ignore_eot && idx == p.eot) {
                p.finished[b] = 1;
                atomicAdd(&p.ctl->n_finished, 1);
            }
        }
    }
}
void launch_argmax_step(const ArgmaxParams& p, hipStream_t st) {
    hipLaunchKernelGGL(argmax_step_kernel, dim3(p.B), dim3(256), 0, st, p);
}
__global__ __launch_bounds__(256) void argmax_plain_kernel(const float* t, int n, int* idx) {
    float best;
    int i;
    argmax_block(t, n, best, i);
    if (threadIdx.x == 0) *idx = i;
}
void launch_argmax_plain(const float* t, int n, int* idx, hipStream_t st) {
    hipLaunchKernelGGL(argmax_plain_kernel, dim3(1), dim3(256), 0, st, t, n, idx);
}

// current_len += 1 (layers.mojo:143) and every utterance's position += 1
__global__ void advance_kernel(StepCtl* ctl, int* pos, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) pos[i] += 1;
    if (i == 0) ctl->len += 1;
}
void launch_advance(StepCtl* ctl, int* pos, int B, hipStream_t st) {
    hipLaunchKernelGGL(advance_kernel, dim3((B + 255) / 256), dim3(256), 0, st, ctl, pos, B);
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
// in [rows][cols] fp32 -> out [rows][cols_pad] T, zero padded columns
template <typename T> __global__ void pad_rows_kernel(const float* in, T* out, int rows, int cols, int cols_pad) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (size_t)rows * cols_pad) {
        int r = (int)(i / cols_pad), c = (int)(i % cols_pad);
        out[i] = from_f32<T>(c < cols ? in[(size_t)r * cols + c] : 0.f);
    }
}
template <typename T> void launch_pad_rows(const float* in, void* out, int rows, int cols, int cols_pad, hipStream_t st) {
    size_t n = (size_t)rows * cols_pad;
    hipLaunchKernelGGL(pad_rows_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, (T*)out, rows, cols,
                       cols_pad);
}
template void launch_pad_rows<float>(const float*, void*, int, int, int, hipStream_t);
template void launch_pad_rows<bf16>(const float*, void*, int, int, int, hipStream_t);
template void launch_pad_rows<f16>(const float*, void*, int, int, int, hipStream_t);
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
