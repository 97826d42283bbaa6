/* whisper_mi.h — C-ABI of libwhispermi.so: the MI355X (gfx950) replacement for the hot path of
 * antonvice/whisper.Mojo.  Plain C, no torch / C++ types cross this boundary.
 *
 * The reference has no FFI of its own (it is one Mojo program); the surface below is what a Mojo
 * `sys.ffi.DLHandle` (or Python ctypes) binding of that program would bind, one entry per reference
 * function it replaces (file:line relative to the reference repo).  INTEGRATION.md shows the Mojo stub.
 *
 * Conventions: every function returns 0 on success or a negative WM_E_* code and never throws;
 * wm_last_error() gives the message (thread-local).  Opaque handles own all device memory; every host
 * buffer is caller-owned and caller-sized.  One wm_model lives on one GPU; calls on one handle must be
 * serialised by the caller (the reference is single-caller too: whisper.mojo:184 takes self immutably and
 * builds its cache locally).  Different handles (other GPUs / processes) are independent.
 */
#ifndef WHISPER_MI_H
#define WHISPER_MI_H
#include <stddef.h>
#include <stdint.h>
#include "wm_synth.h" /* wm_dims */

#ifdef __cplusplus
extern "C" {
#endif

enum { WM_OK = 0, WM_E_ARG = -1, WM_E_SIZE = -2, WM_E_IO = -3, WM_E_HIP = -4, WM_E_STATE = -5 };
enum { WM_F32 = 0, WM_BF16 = 1, WM_F16 = 2 };
enum { WM_GELU_TANH = 0 /* whisper_tensor.mojo:288-308 */, WM_GELU_ERF = 1 /* HF */ };
enum { WM_POS_REF = 0 /* start_pos = current_len-1, whisper.mojo:217 */, WM_POS_HF = 1 /* = current_len */ };

/* Replaces WhisperConfig (whisper.mojo:15-31) + config.mojo:4-17.  dims.d_model / dims.n_heads must be 64
 * (layers.mojo:190-198 hard-codes a 64-wide head). */
typedef struct {
    wm_dims dims;
    int gelu_mode;     /* WM_GELU_* */
    int compute_dtype; /* WM_F32: exact fp32 MFMA everywhere.  WM_BF16 / WM_F16: GEMM operands (weights and
                          the activations fed to them) are 16-bit, accumulation / LayerNorm / softmax /
                          residual stream stay fp32 */
    int kv_dtype;      /* storage type of the self- and cross-attention K/V cache: WM_F32 or compute_dtype */
    int max_batch;     /* largest B any later call will pass (arena sizing) */
    int decoder_fp32;  /* != 0: compute_dtype applies to the ENCODER only (conv stem, encoder blocks, cross-K/V projection); the
                          decoder's weights and MFMA operands stay fp32 — BASELINE config 3 read literally ("bf16 encoder GEMMs").
                          0 (a zero-filled tail): one dtype for both, as before */
    int coalesce;      /* 2: two consecutive wm_transcribe_submit calls with the same batch size and options share ONE decode state of
                          2·B rows (the latency-bound launches of a decode step cost the same for 128 rows as for 64): the first call of
                          a pair is held until its partner arrives — or until it is waited for, then it runs alone — and every call
                          still gets exactly its own ids, bit-identical to an uncoalesced run (nothing in an utterance's arithmetic
                          depends on the batch it rides in).  A host-side mel buffer must then stay valid until the matching wait.
                          0 / 1: off */
} wm_config;

/* Replaces the literals in Whisper.transcribe (whisper.mojo:187-191 prompt, :206 eot, :205 loop bound). */
typedef struct {
    const int32_t* prompt;
    int n_prompt;
    int eot;
    int max_loop;   /* reference: 195 -> at most n_prompt + 1 + 195 ids per utterance */
    int pos_mode;   /* WM_POS_* */
    int ignore_eot; /* != 0: "fixed" mode — never stop early (bench) */
    /* SURVEY §8f rank 4 — logit masks the reference lacks (whisper.mojo:198,219 take the raw argmax), with the semantics
     * of HF generate's SuppressTokensLogitsProcessor / SuppressTokensAtBeginLogitsProcessor; applied inside the fused
     * argmax, never to the logits wm_decode_step returns.  NULL / 0 = none (the reference's behaviour). */
    const int32_t* suppress_tokens;       /* ids that can never be emitted */
    int n_suppress;
    const int32_t* begin_suppress_tokens; /* ids that cannot be the FIRST generated token */
    int n_begin_suppress;
    /* Timestamp rules, same row: the semantics of HF generate's WhisperTimeStampLogitsProcessor, applied after the two suppress
     * masks inside the fused argmax — timestamps come in pairs, never decrease, the first generated id is a timestamp no
     * later than timestamp_begin + max_initial_timestamp_index, <|notimestamps|> is never emitted, and whenever the
     * probability mass of all admissible timestamps exceeds the most likely text id the id is a timestamp.  `eot` doubles as
     * the processor's eos_token_id (ids below it are "normal text").  timestamp_begin <= 0 = off (the reference's behaviour;
     * a zero-filled struct tail keeps it off). */
    int timestamp_begin;             /* id of <|0.00|> (50364 for the multilingual vocabulary) */
    int no_timestamps_token;         /* id of <|notimestamps|> (50363), or < 0 */
    int max_initial_timestamp_index; /* HF generation config default 50 (= 1.00 s); < 0 = unlimited */
} wm_decode_opts;

typedef struct wm_model wm_model;
typedef struct wm_state wm_state;

const char* wm_last_error(void);

/* wm_config and wm_decode_opts have grown round by round (decoder_fp32, the timestamp fields) and carry no size field: a host
 * built against an older header would hand the library structs whose tail it never wrote.  WM_ABI_VERSION is bumped with every
 * such change; a host binding compares it with wm_abi_version() once after loading the library and refuses a mismatch (the
 * Python and C++ mirrors do). */
#define WM_ABI_VERSION 4
int wm_abi_version(void);

/* ---- WeightLoader(filename) + Whisper() + Whisper.load(loader)   loader.mojo:10-27, whisper.mojo:175-182 ----
 * path: the reference's headerless little-endian fp32 file (order of export_weights.py:19-90).  Unlike the
 * reference (loader.mojo:21-27 reads past the end silently) the byte size is validated: WM_E_SIZE. */
int wm_model_load(const char* path, const wm_config* cfg, int device, wm_model** out);
/* Same from a host image of n_floats fp32 values. */
int wm_model_load_memory(const float* weights, size_t n_floats, const wm_config* cfg, int device, wm_model** out);
void wm_model_free(wm_model* m);
size_t wm_weight_count(const wm_dims* dims);

/* ---- weight file format v2 ------------------------------------------------------------------------
 * v1 = the reference's headerless fp32 dump.  v2 = 64-byte header (magic "WMIWGT2", version, matrix dtype, dims, flags,
 * payload size) + the same tensors in the same order, conv / linear matrices in `dtype` (WM_F32 / WM_BF16 / WM_F16), all
 * vectors and positional tables in fp32; the token embedding stays fp32 when emb_f32 != 0 (then a v2 file in the model's
 * compute dtype loads to exactly the weights the v1 file gives).  wm_model_load accepts either format and validates
 * size / dims.  wm_weights_read expands either format to the fp32 image (wm_weight_count(dims) floats).  Host-only. */
int wm_weights_convert_v2(const char* v1_path, const char* v2_path, const wm_dims* dims, int dtype, int emb_f32);
int wm_weights_read(const char* path, const wm_dims* dims, float* out);

/* ---- KVCache(n_layers, d_model, 448)   layers.mojo:55-63 — one per batch of B utterances ------------------- */
int wm_state_new(wm_model* m, int batch, wm_state** out);
int wm_state_reset(wm_state* s); /* current_len = 0, has_cross = false */
void wm_state_free(wm_state* s);
int wm_state_len(const wm_state* s); /* LayerCache.current_len (layers.mojo:18) */

/* ---- WhisperEncoder.forward(mel) -> Tensor   whisper.mojo:71-99 ---------------------------------------------
 * mel: [B, n_mels, 2*n_audio_ctx] fp32 row-major (sample_input.bin layout, main.mojo:23-27); host pointer, or
 * a device pointer on the model's GPU when mel_on_device != 0.  Resets `s`, keeps the encoder output inside it
 * for the decoder, and (if enc_out != NULL) copies it to host as [B, n_audio_ctx, d_model] fp32. */
int wm_encode(wm_model* m, wm_state* s, const float* mel, int mel_on_device, int B, float* enc_out);
/* Stage-level tests: inject an encoder output [B, n_audio_ctx, d_model] (host) instead of running the encoder. */
int wm_state_set_encoder_output(wm_model* m, wm_state* s, const float* enc_out, int B);

/* ---- WhisperDecoder.forward(tokens, enc_out, cache, use_cache=True, start_pos) -> logits  whisper.mojo:130-167 --
 * tokens [B, q_len] (host), start_pos [B] (host; the reference passes one Int — per-utterance here; it stays a
 * CALLER argument so that the position quirk of whisper.mojo:217 lives in the host loop).  Appends q_len
 * positions to the cache (layers.mojo:140-143); computes the cross K/V on first use (layers.mojo:150-154).
 * logits: NULL or host [B, vocab] for the last position; next: NULL or host [B] = argmax (lowest index wins,
 * whisper_tensor.mojo:431-439). */
int wm_decode_step(wm_model* m, wm_state* s, const int32_t* tokens, int q_len, const int32_t* start_pos,
                   float* logits, int32_t* next);

/* ---- Whisper.transcribe(mel) -> List[Int]   whisper.mojo:184-223 ------------------------------------------
 * tokens_out: host [B, n_prompt + 1 + max_loop]; row b holds prompt + generated ids (incl. the trailing eot when
 * hit), exactly the reference's list; n_tokens[b] = its length.  The whole greedy loop runs on the GPU (token
 * feedback never visits the host); the host only reads a device-written "utterances finished" word between sub-chunks of 8 steps
 * and stops enqueueing once every utterance has emitted eot (see wm_transcribe_submit). */
int wm_transcribe(wm_model* m, const float* mel, int mel_on_device, int B, const wm_decode_opts* opts,
                  int32_t* tokens_out, int32_t* n_tokens);

/* Pipelined form for back-to-back batches (serving / bench): wm_transcribe_submit enqueues the encoder, the prompt prefill and
 * the greedy loop on the slot's own HIP stream and returns immediately; wm_transcribe_wait blocks until that slot's ids are
 * ready and copies them out (same layout as wm_transcribe).  Eight slots (0..7): submitting batch i+1 before waiting for batch i
 * overlaps its MFMA-bound encoder with batch i's latency/HBM-bound decode; four passes in flight is the measured optimum (the chip
 * runs four hardware queues at a time, and ROCm must be allowed that many per-process queues: GPU_MAX_HW_QUEUES >= 8 in the
 * environment before HIP initialises — see INTEGRATION.md).
 * The loop stops like the reference's (whisper.mojo:206-207: break at eot — here: once EVERY utterance of the batch has emitted
 * eot): with ignore_eot == 0 it is enqueued in sub-chunks of 8 steps, two sub-chunks ahead of the GPU, by a host thread of the
 * library that reads a device-written "utterances finished" word between sub-chunks (no stream synchronisation); it therefore runs
 * at most 16 steps (+ the sub-chunk in progress) past the longest utterance, and the ids are identical to wm_transcribe's.  With
 * ignore_eot != 0 ("fixed" mode) all max_loop steps are enqueued at submit.  mel must stay valid until the matching wait when it
 * is a device pointer. */
int wm_transcribe_submit(wm_model* m, int slot, const float* mel, int mel_on_device, int B, const wm_decode_opts* opts);
int wm_transcribe_wait(wm_model* m, int slot, int32_t* tokens_out, int32_t* n_tokens);
/* wm_transcribe_wait for the multi-GPU path: the slot's result stays ON THE DEVICE, packed as the gather buffer the one
 * collective of a batch moves — dev_packed [rows_cap, 1 + stride] int32 in the caller's DEVICE memory (e.g. a torch tensor handed to
 * RCCL's all-gather): row r = [n_tokens[r], ids zero-padded to stride]; rows past the batch are zero (ragged shards gather a fixed
 * row count per rank).  rows_cap >= B, stride >= n_prompt + 1 + max_loop.  Returns when the buffer is complete. */
int wm_transcribe_wait_device(wm_model* m, int slot, int32_t* dev_packed, int rows_cap, int stride);
/* Loop iterations that were enqueued for the slot's most recent completed pass (slot 0 also serves wm_transcribe): max_loop when
 * the pass ran to its bound, less when the early exit cut it.  -1: no pass yet / bad slot.  Diagnostics and tests. */
int wm_transcribe_steps(wm_model* m, int slot);

/* ---- log-mel front end --------------------------------------------------------------------------
 * Replaces the reference's call to HF WhisperProcessor (export_weights.py:100-116): 16 kHz mono PCM -> pad / trim to the
 * 30 s window -> 400-point Hann STFT (hop 160, reflect padding) -> 80 slaney mel bands -> log10 -> clamp to max-8 ->
 * (x+4)/4.  pcm: host [B, stride] fp32, n_samples[b] <= stride valid samples each.  mel_out: NULL or host
 * [B, n_mels, 2*n_audio_ctx] (the sample_input.bin layout the encoder takes). */
int wm_log_mel(wm_model* m, const float* pcm, const int32_t* n_samples, int B, int stride, float* mel_out);
/* PCM in, token ids out: front end + wm_transcribe without the mel ever leaving the GPU. */
int wm_transcribe_pcm(wm_model* m, const float* pcm, const int32_t* n_samples, int B, int stride, const wm_decode_opts* opts,
                      int32_t* tokens_out, int32_t* n_tokens);

/* ---- op-level entry points (host pointers; known-answer tests only) ------------------------------------------
 * Same argument meaning as the reference ops: out-param first, caller-allocated. */
/* matmul(C, A, B, bias)  whisper_tensor.mojo:151-246 : C[M,N] = A[M,K]·B[N,K]ᵀ (+bias[N], may be NULL).
 * dtype selects the operand rounding (WM_F32 exact).  Requires K % 32 == 0. */
int wm_op_matmul_nt(float* C, const float* A, const float* B, const float* bias, int M, int N, int K, int dtype);
/* C = layer_norm(A, ln_g, ln_b, 1e-5) · Bᵀ (+ bias): the LayerNorm -> projection pair of ResidualAttentionBlock.forward
 * (layers.mojo:449-455, 489-497) on the encoder's kernels.  require_fused != 0 demands the one-kernel form (the LayerNorm applied while
 * the GEMM loads its fp32 A rows; 16-bit dtypes, K = 384, N % 128 == 0, N <= 3072): any other shape is REFUSED by the kernel launcher —
 * WM_E_ARG, wm_last_error says why, nothing is launched and C is untouched.  require_fused == 0: fused where possible, else a
 * LayerNorm launch + a plain GEMM.  N % 128 == 0, K % 128 == 0, K <= 1024.  Known-answer and negative tests. */
int wm_op_ln_matmul_nt(float* C, const float* A, const float* ln_g, const float* ln_b, const float* B, const float* bias, int M, int N,
                       int K, int dtype, int require_fused);
/* layer_norm(out, inp, gamma, beta, eps)  whisper_tensor.mojo:249-285 (one-pass variance). cols % 128 == 0, <= 1024 */
int wm_op_layer_norm(float* out, const float* inp, const float* gamma, const float* beta, int rows, int cols,
                     float eps);
/* The q_len == 1 path of MultiHeadAttention.forward over cached keys / values (layers.mojo:186-272): per utterance and head
 * s_j = (q·K_j)·0.125 (scale AFTER the dot product), running max from -1e10, p = exp(s - max), o = Σ p_j V_j / Σ p_j.
 * q, out [B, 64·n_heads] fp32; k, v [B, t, 64·n_heads] fp32 host rows, rounded to kv_dtype as the cache holds them.
 * n_chunks > 1: the keys of an utterance are swept by n_chunks workgroups and merged (the cross-attention form; needs
 * t/512 <= n_chunks <= t/32 rounded up);  n_chunks == 1: one workgroup per utterance, the key count read from the decode
 * control block (the self-attention form).  Known-answer tests. */
int wm_op_attention_cached(float* out, const float* q, const float* k, const float* v, int B, int t, int n_heads, int kv_dtype,
                           int n_chunks);
/* The MLP half of ResidualAttentionBlock.forward  layers.mojo:489-517 :  x += fc2(gelu(fc1(layer_norm(x, ln_g, ln_b)))),
 * x [M, d] in place; fc1_w [ffn, d], fc2_w [d, ffn] (HF [out, in]).  With next_g / next_b / xn_out non-NULL also returns
 * layer_norm(x_new, next_g, next_b) rounded to the operand dtype (what the next projection is fed) in xn_out [M, d].
 * Runs on the encoder's own kernels: for 16-bit dtypes with d = 384 the LayerNorm rides the fc1 GEMM's A load, and the residual
 * add and the next LayerNorm ride fc2's epilogue.  d % 128 == 0, ffn % 128 == 0. */
int wm_op_mlp_block(float* x, const float* ln_g, const float* ln_b, const float* fc1_w, const float* fc1_b, const float* fc2_w,
                    const float* fc2_b, const float* next_g, const float* next_b, float* xn_out, int M, int d, int ffn, int dtype,
                    int gelu_mode);
/* The block attention path of MultiHeadAttention.forward without cache and without mask (the encoder's: layers.mojo:273-342:
 * per head gather, S = q_h·k_hᵀ, scale 1/8 after the product, row softmax, O = S·v_h, scatter).  q, k, v, out [n_ctx, 64·n_heads]
 * fp32 host rows; dtype = operand rounding of q, k, v and of the probabilities (WM_F32: exact).  Runs the encoder's fused
 * attention kernel (never materialises S).  Known-answer tests. */
int wm_op_attention(float* out, const float* q, const float* k, const float* v, int n_ctx, int n_heads, int dtype);
/* gelu(t) in place  whisper_tensor.mojo:288-308 (mode WM_GELU_TANH) */
int wm_op_gelu(float* t, size_t n, int mode);
/* softmax(t) rows in place  whisper_tensor.mojo:311-355 */
int wm_op_softmax_rows(float* t, int rows, int cols);
/* conv1d(out, inp, weight, bias, stride, padding, out_T)  whisper_tensor.mojo:367-428; K=3, padding=1.
 * inp [C_in, L_in]; weight in the ORIGINAL [C_out, C_in, 3] file layout (the transpose of
 * whisper_tensor.mojo:358-364 is internal); out [C_out, L_out] or [L_out, C_out] when out_T. */
int wm_op_conv1d_k3(float* out, const float* inp, const float* weight, const float* bias, int C_in, int L_in,
                    int C_out, int stride, int out_T, int dtype);
/* argmax(t)  whisper_tensor.mojo:431-439 */
int wm_op_argmax(const float* t, int n, int32_t* idx);

/* ---- measurement helpers (bench.py) ----------------------------------------------------------------------------- */
enum { WM_KERNEL_CROSS_ATTN = 0, WM_KERNEL_DECODE_STEP = 1, WM_KERNEL_ENCODER = 2,
       WM_KERNEL_DECODE_STEP_SHARED = 3 /* the step as wm_transcribe_submit's passes run it: K/V stream at two workgroups per CU */ };
/* Launches `reps` instances of the named kernel / stage on the library's stream between two HIP events and
 * returns the average duration in microseconds.  State must have been encoded (cross K/V present). */
int wm_bench_kernel(wm_model* m, wm_state* s, int which, int reps, float* avg_us);
/* Algorithmic HBM bytes one launch of `which` moves for state `s` (SURVEY §8d formulas). */
int wm_bench_bytes(wm_model* m, wm_state* s, int which, double* bytes);

/* Synthetic weights / mels (include/wm_synth.h) exported for hosts that cannot include the header. */
size_t wm_synth_weights(const wm_dims* dims, uint64_t seed, float* out);
void wm_synth_mel_host(uint64_t seed, int n_mels, int n_frames, float* out);

#ifdef __cplusplus
}
#endif
#endif
