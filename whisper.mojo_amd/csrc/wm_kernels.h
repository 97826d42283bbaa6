// wm_kernels.h — launcher declarations shared by the kernel translation units and the C-ABI host code.
#pragma once
#include "wm_device.h"

#include <atomic>
#include <cstdlib>

// Developer switches (A/B knobs, timelines, debug chains) exist only in the -DWM_DEV build (libwhispermi_dev.so, built by
// `python whisper.mojo_amd/build.py --dev`).  In the product library wm_env() is a constant nullptr, so every switch folds
// to its measured-best default at compile time and no launch path reads the environment.
#ifdef WM_DEV
static inline const char* wm_env(const char* name) { return std::getenv(name); }
#else
static constexpr const char* wm_env(const char*) { return nullptr; }
#endif

namespace wm {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE property of ONE kernel: the record is keyed on the kernel itself
// (a non-type template parameter — keyed on the function TYPE, every instantiation of a kernel template with the same signature
// shared one flag) and holds, per device, the largest byte count set so far.  Returns the HIP status.
template <auto Kernel> static inline hipError_t ensure_dyn_lds(int bytes) {
    static std::atomic<int> set_bytes[64];  // zero-initialised; index = device
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::atomic<int>& have = set_bytes[dev & 63];
    if (have.load(std::memory_order_acquire) >= bytes) return hipSuccess;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) {
        int cur = have.load(std::memory_order_relaxed);
        while (cur < bytes && !have.compare_exchange_weak(cur, bytes, std::memory_order_release)) {
        }
    }
    return e;
}

// Launcher status: a launcher that cannot serve a shape REFUSES it (nothing is launched) and says so; the C-ABI entry above turns
// that into WM_E_ARG with wm_last_error set.  0 = launched.
enum { WM_LAUNCH_OK = 0, WM_LAUNCH_BAD_SHAPE = 1, WM_LAUNCH_HIP = 2 };
// message of the last refusal on this thread (static text)
const char* launch_last_refusal();
int launch_refuse(const char* why);  // records `why`, returns WM_LAUNCH_BAD_SHAPE
int launch_hip_failed(const char* what, hipError_t e);

// Per-state decode control block, resident in HBM so that a captured decode step can be replayed unchanged.
struct StepCtl {
    int len;         // LayerCache.current_len BEFORE this step (= cache row the new K/V is written to)
    int n_finished;  // utterances that have emitted eot
    int pad0, pad1;
};

// ---- encoder ------------------------------------------------------------------------------------------------
struct GemmParams {
    const void* A;
    const void* W;
    void* C;
    int M, N, K;  // per batch
    long lda, ldw, ldc;
    long strideA, strideC;  // per-batch element strides (grid.z)
    const float* bias;      // [N] or null
    const float* residual;  // fp32, indexed [m*ldr + n] (+ z*strideR), or null
    long ldr, strideR;
    const float* pos;  // fp32 [M][N] added after the activation (conv2 + pos_emb, whisper.mojo:83-89), or null
    int act;           // 0 none, 1 gelu
    int gelu_mode;
    int group_n;  // >0: column n goes to C + (n/group_n)*group_stride + m*ldc + n%group_n  (cross-K/V scatter)
    long group_stride;
    int xcd_remap;  // set by the launcher (LDS-staged kernel only)
    // LayerNorm fused into the A load (row-panel kernel only — ask gemm_nt_fuses_layernorm first): A is then the fp32 residual
    // stream [M][lda] and the operand is LN(A) * ln_g + ln_b rounded to T, exactly what layernorm_rows + a plain GEMM compute
    const float* ln_g;
    const float* ln_b;
    // LayerNorm of the OUTPUT rows fused into the epilogue (full-row kernel only — ask gemm_nt_fuses_layernorm_out first): besides
    // C the kernel writes LN(C) * lno_g + lno_b as 16-bit operands to lno_out, laid out like C (ldc, strideC): the next GEMM's A
    const float* lno_g;
    const float* lno_b;
    void* lno_out;
    long long* dbg;  // developer build: phase stamps of the row-panel kernel (wm_bench_kernel id 43), else null
};
bool gemm_nt_fuses_layernorm_out(int operand_bytes, const GemmParams& p);
// true when launch_gemm_nt<T, *> takes the A-stationary row-panel kernel for these parameters (the only one that can fuse a LayerNorm)
bool gemm_nt_fuses_layernorm(int operand_bytes, const GemmParams& p, int batch);
template <typename T> void launch_mel_transpose_pad(const float* mel, void* out, int B, int C, int L, int Cp, hipStream_t st);
template <typename T, typename TO> int launch_gemm_nt(const GemmParams& p, int batch, hipStream_t st);  // WM_LAUNCH_*
template <typename T>
void launch_layernorm_rows(const float* x, const float* gamma, const float* beta, void* out_t, float* out_f, int rows,
                           int cols, float eps, hipStream_t st);
template <typename T> void launch_flash_attn_enc(const void* qkv, void* out, int B, int H, int n_ctx, float scale, hipStream_t st);

// ---- decoder ------------------------------------------------------------------------------------------------
// Timestamp rules of the fused argmax (semantics of HF's WhisperTimeStampLogitsProcessor, absent from the
// reference).  Per utterance: what the rules need of the history, and the id ranges the NEXT argmax may choose from — written
// by stage 2 of one step (argmax_step), read by stage 1 of the next (dec_logits).
struct TsState {
    int n_gen;    // ids generated so far (after the prompt)
    int last_ts;  // the last one was a timestamp
    int pen_ts;   // the one before was a timestamp (or there is none)
    int t_last;   // the latest timestamp id emitted, -1 = none
    int text_lo, text_hi;  // admissible "text" ids (everything below timestamp_begin): [text_lo, text_hi)
    int ts_lo, ts_hi;      // admissible timestamp ids: [ts_lo, ts_hi)
};
struct TsRules {
    int tb;        // timestamp_begin: first timestamp id (<|0.00|>); <= 0 = rules off
    int eos;       // ids below it are "normal text" (masked after the first timestamp of a pair)
    int max_init;  // max_initial_timestamp_index, < 0 = none
    int vocab;
};
__host__ __device__ inline void ts_next_ranges(TsState& st, const TsRules& r) {  // ranges for the next argmax from the history
    st.text_lo = 0;
    st.text_hi = r.tb;
    st.ts_lo = r.tb;
    st.ts_hi = r.vocab;
    if (st.n_gen == 0) {  // the first id is a timestamp, at most <|max_init * 0.02|>
        st.text_hi = 0;
        if (r.max_init >= 0 && r.tb + r.max_init + 1 < st.ts_hi) st.ts_hi = r.tb + r.max_init + 1;
        return;
    }
    if (st.last_ts) {
        if (st.pen_ts)
            st.ts_hi = st.ts_lo;  // two in a row: the next one is not a timestamp
        else
            st.text_lo = r.eos;  // first of a pair: no normal text
    }
    if (st.t_last >= 0) {  // never decrease; do not re-emit a closed pair's value
        const int ts_end = (st.last_ts && !st.pen_ts) ? st.t_last : st.t_last + 1;
        if (ts_end > st.ts_lo) st.ts_lo = ts_end;
    }
}

struct DecLinearParams {
    const float* x;  // [B][ldx] fp32 activations
    int ldx;
    int x_is_t;    // != 0 (no LayerNorm prologue): x holds operand-dtype elements (TW) written by the producer kernel — exactly the
                   // values the MFMA fragment conversion would produce, so results are unchanged; half the bytes, no conversion
    int out_is_t;  // != 0 (plain output, no residual / cache append): out is stored as TW for such a consumer
    const float* ln_g;  // LayerNorm prologue (null = none)
    const float* ln_b;
    const void* W;  // [N][K], operand dtype
    int N, K, B;
    const float* bias;  // [N] or null
    int act, gelu_mode;
    const float* residual;  // [B][ldr] fp32 or null (may alias out)
    int ldr;
    float* out;  // [B][ldo] fp32
    int ldo;
    // QKV mode (kcache != null): columns [0,d) -> out (q), [d,2d) -> kcache, [2d,3d) -> vcache at row ctl->len
    void* kcache;
    void* vcache;
    long kv_batch_stride;  // elements between utterances in the cache
    int d_model;
    int kv_dtype;  // WM_F32 / WM_BF16 / WM_F16
    int kv_B;      // > 0: prompt prefill, rows are position-major (row = t * kv_B + b): row's K/V go to utterance b, cache row len + t
    const StepCtl* ctl;
    // dec_logits only: fused argmax stage 1 — per-utterance best (value, column) of each 128-column workgroup
    float* amax_val;  // [B][amax_stride] or null
    int* amax_idx;
    int amax_stride;
    const float* amax_mask;  // [N] additive mask (0 / -inf) applied to the argmax candidates only, or null
    // timestamp rules (null = off): candidates are split into text ids and timestamp ids by the utterance's TsState ranges; the
    // workgroups that cover ids >= ts_begin also emit the best admissible timestamp and (max, sum exp) of the admissible ones
    const TsState* ts_state;  // [B]
    int ts_begin;
    float* ts_val;  // [B][amax_stride] like amax_val; written for parts >= ts_begin / (ids per part) only
    int* ts_idx;
    float* ts_m;
    float* ts_s;
    long long* ts;  // developer timeline (dec_logits only)
    int ts_id;
    long long* dbg;  // developer build: per-(workgroup, wave) phase stamps of dec_logits (100 MHz clock), null = off
};
template <typename TW> int launch_dec_linear(const DecLinearParams& p, hipStream_t st);  // WM_LAUNCH_*
bool dec_linear_supports_k(int K);  // K/32 k-steps must split into NW <= 16 waves x KPW <= 4 steps (checked at model load)
template <typename TW> int launch_dec_logits(const DecLinearParams& p, hipStream_t st);  // WM_LAUNCH_*
int dec_logits_parts(int N);  // fused-argmax partials per utterance that launch_dec_logits writes (amax_stride must cover them)
int dec_logits_ids_per_part(int N);  // vocabulary ids one part covers

struct AttnDecParams {
    const float* q;  // [B][d] fp32
    const void* K;   // [B][rows][d] cache of this layer
    const void* V;
    long batch_stride;
    int n_keys;  // >= 0: fixed key count (cross);  < 0: ctl->len + 1 (self, includes the row just written)
    int q_B;     // > 0: prompt prefill, query rows are position-major (row = t * q_B + b): K/V of utterance b, self length + t
    int nq;      // 4 (cross-attention, with q_B): one workgroup per (utterance, chunk) serves the 4 positions; else 0/1
    const StepCtl* ctl;
    int nsplit;
    float scale;
    float* part_o;      // [B][nsplit][d]
    float* part_ml;     // [B][nsplit][H][2]
    float* direct_out;  // non-null (nsplit must be 1): write the normalised output [B][d] here, skip the partials
    int out_dtype;      // element type of direct_out: WM_F32 / WM_BF16 / WM_F16 (the operand dtype of the projection that reads it)
    int H, d, B;
    int rps;  // filled by the launcher
    // cross-attention only: bytes of UNUSED dynamic LDS per workgroup.  34 KB (+ 20 KB static) admits two workgroups per CU
    // instead of four and leaves 52 KB of LDS and half the register file free: when several passes share the chip
    // (wm_transcribe_submit), the other passes' latency-bound launches then start beside this K/V stream instead of queueing
    // behind it — measured four passes in flight, tiny B = 64: 19.6 -> 19.1 ms per pass, although the kernel alone is
    // 1 us slower (26.3 vs 25.3 us).  0 for a pass that has the chip to itself.
    int lds_pad;
    long long* ts;  // developer timeline (null = off): see ts_put in kernels_decoder.hip
    int ts_id;
};
template <typename TKV> int launch_attn_decode(const AttnDecParams& p, hipStream_t st);  // WM_LAUNCH_*
void launch_attn_combine(const float* part_o, const float* part_ml, void* out, int out_dtype, int B, int nsplit, int H, int d,
                         hipStream_t st, long long* ts = nullptr, int ts_id = 0);

void launch_dec_embed(const float* tok_emb, const float* pos_emb, const int* tok, const int* pos, float* x, int B, int d,
                      hipStream_t st);
// argmax over logits rows (lowest index wins) + greedy-loop bookkeeping
struct ArgmaxParams {
    const float* logits;
    int ldl, V, B;
    const float* pval;  // non-null: reduce the fused-argmax partials [B][npart] instead of scanning the logits
    const int* pidx;
    int npart;
    int* next;        // [B] next token (also the next step's input)
    int* out_tokens;  // [B][out_stride] or null
    int out_stride;
    int* n_tokens;  // [B]
    int* finished;  // [B]
    StepCtl* ctl;
    int eot, ignore_eot;
    int advance;  // != 0: also do the end-of-step bookkeeping (len += 1, pos[b] += 1)
    int* pos;
    // non-null: a host-visible (pinned, device-mapped) pair the launch updates for the host's early-exit check, no stream
    // synchronisation needed: [0] = utterances finished BEFORE this step (never more than the truth), [1] = loop steps completed
    int* host_progress;
    long long* ts;  // developer timeline (null = off)
    int ts_id;
    // non-null: also write the NEXT step's input row x[b] = tok_emb[chosen id] + pos_emb[pos[b] (+1 when advancing)] — the
    // embedding launch of the following step folded into this one
    const float* emb_tok;
    const float* emb_pos;
    float* emb_out;
    int d, max_pos;
    // timestamp rules (ts_state null = off): pval / pidx then hold the best admissible TEXT id of each part, ts_* the timestamp side
    TsState* ts_state;
    TsRules rules;
    const float* ts_val;
    const int* ts_idx;
    const float* ts_m;
    const float* ts_s;
    int ts_part0;  // first part that covers timestamp ids
};
void launch_argmax_step(const ArgmaxParams& p, hipStream_t st);
struct InitTokensParams {
    int* out_tokens;
    int out_stride;
    int* n_tokens;
    int* finished;
    StepCtl* ctl;
    int B, n_prompt;
    int prompt[16];
    int* tok_rows;  // non-null: also fill the prefill's position-major token / position rows [n_prompt][B]
    int* pos_rows;
    TsState* ts_state;  // non-null: start state of the timestamp rules
    TsRules rules;
};
void launch_init_tokens(const InitTokensParams& p, hipStream_t st);
// rows of the gather buffer of SURVEY §8e: dst[r] = [n_tokens[r], ids of row r zero-padded to `stride`] for r < rows; rows in
// [rows, rows_cap) are zeroed (ragged shards gather a fixed row count per rank)
void launch_pack_tokens(const int* out_tokens, const int* n_tokens, int out_stride, int rows, int rows_cap, int stride, int* dst, hipStream_t st);
void launch_set_step(StepCtl* ctl, int len, int set_len, int* pos, int pos_value, int* tok, int tok_value, int B,
                     hipStream_t st);

// ---- log-mel front end (kernels_frontend.hip) ------------------------------------------------------------------------
void launch_zero_tails(float* pcm, const int* len, int B, int N, hipStream_t st);
void launch_frames(const float* pcm, float* F, const float* window, int B, int N, int n_frames, int hop, hipStream_t st);
void launch_mel_log(const float* spec, const float* fb, const int* band, float* logmel, int B, int n_frames, int n_mels, hipStream_t st);
void launch_mel_norm(const float* logmel, float* out, int B, int n, hipStream_t st);

// ---- small ops for the op-level C-ABI ----------------------------------------------------------------------------
void launch_gelu(float* t, size_t n, int mode, hipStream_t st);
void launch_softmax_rows(float* t, int rows, int cols, hipStream_t st);
void launch_argmax_plain(const float* t, int n, int* idx, hipStream_t st);
template <typename T> void launch_convert(const float* in, void* out, size_t n, hipStream_t st);
void launch_transpose_f32(const float* in, float* out, int rows, int cols, hipStream_t st);

}  // namespace wm
