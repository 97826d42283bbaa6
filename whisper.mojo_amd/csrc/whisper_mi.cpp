// whisper_mi.cpp — host runtime behind the C-ABI of include/whisper_mi.h (compiled with hipcc).
//
// One wm_model per GPU: weights resident in HBM in kernel-friendly layouts, one HIP stream, no per-op allocation
// (the reference zero-fills a fresh heap Tensor for every intermediate, whisper_tensor.mojo:17-23; here every
// buffer belongs to a preallocated per-state arena).  The greedy loop of Whisper.transcribe (whisper.mojo:184-223)
// runs with the token feedback entirely on the device.
#include "../../include/whisper_mi.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <string>
#include <unordered_set>
#include <vector>

#include "wm_kernels.h"

using namespace wm;

// ------------------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCHK(expr)                                                                                     \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) return fail(WM_E_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define WMCHK(expr)            \
    do {                       \
        int rc_ = (expr);      \
        if (rc_ != 0) return rc_; \
    } while (0)

extern "C" const char* wm_last_error(void) { return g_err.c_str(); }
extern "C" int wm_abi_version(void) { return WM_ABI_VERSION; }

// a launcher's status (wm_kernels.h WM_LAUNCH_*) as a C-ABI status: a refused shape is the caller's argument error
static int launch_rc(int st) {
    if (st == wm::WM_LAUNCH_OK) return 0;
    return fail(st == wm::WM_LAUNCH_HIP ? WM_E_HIP : WM_E_ARG, "%s", wm::launch_last_refusal());
}
#define LCHK(expr) WMCHK(launch_rc(expr))

// Developer timeline (WM_TRACE_EVENTS=1): HIP events recorded on the library's streams at pass / phase boundaries and
// printed, sorted on the GPU clock, when the model is freed.  The profiler serialises concurrent queues; events do not.
struct TraceMark {
    hipEvent_t ev;
    std::string label;
};
static std::vector<TraceMark> g_trace;
static bool trace_events_on() {
    static const bool on = wm_env("WM_TRACE_EVENTS") != nullptr;
    return on;
}
static void trace_mark(hipStream_t st, const char* fmt, ...) {
    if (!trace_events_on() || g_trace.size() > 200000) return;
    char buf[128];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, st);
    g_trace.push_back({e, buf});
}
static void trace_dump() {
    if (g_trace.empty()) return;
    (void)hipDeviceSynchronize();
    std::vector<std::pair<float, std::string>> rows;
    for (auto& t : g_trace) {
        float ms = 0.f;
        const hipError_t e = hipEventElapsedTime(&ms, g_trace[0].ev, t.ev);
        if (e == hipSuccess)
            rows.push_back({ms, t.label});
        else
            fprintf(stderr, "[wm-trace] %s: %s\n", t.label.c_str(), hipGetErrorString(e));
    }
    for (auto& t : g_trace) (void)hipEventDestroy(t.ev);
    g_trace.clear();
    std::sort(rows.begin(), rows.end());
    for (auto& r : rows) fprintf(stderr, "[wm-trace] %10.3f ms  %s\n", r.first, r.second.c_str());
}

static size_t dt_size(int dt) { return dt == WM_F32 ? 4 : 2; }
static int dec_dtype(const wm_config& c) { return c.decoder_fp32 ? WM_F32 : c.compute_dtype; }  // operand dtype of the decoder
static inline uint16_t f32_to_bf16_host(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline uint16_t f32_to_f16_host(float f) {
    _Float16 h = (_Float16)f;
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int alloc(size_t n, bool zero = false) {
        release();  // re-allocation never leaks the previous block
        bytes = n ? n : 16;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) return fail(WM_E_HIP, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
        if (zero) {
            // hipMemset on device memory is asynchronous to the host and runs on the null stream, which the library's
            // non-blocking streams do not wait for: finish it here, or the first kernels on a fresh buffer can race the
            // zeroing (seen once in ~6 runs of the eight-slot test as a wrong token row)
            e = hipMemset(p, 0, bytes);
            if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
            if (e != hipSuccess) return fail(WM_E_HIP, "hipMemset: %s", hipGetErrorString(e));
        }
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
    }
    template <typename T> T* as() const { return (T*)p; }
};

// host fp32 -> device buffer in dtype dt
static int upload(DevBuf& b, const float* h, size_t n, int dt) {
    WMCHK(b.alloc(n * dt_size(dt)));
    if (dt == WM_F32) {
        HIPCHK(hipMemcpy(b.p, h, n * 4, hipMemcpyHostToDevice));
    } else {
        std::vector<uint16_t> tmp(n);
        if (dt == WM_BF16)
            for (size_t i = 0; i < n; ++i) tmp[i] = f32_to_bf16_host(h[i]);
        else
            for (size_t i = 0; i < n; ++i) tmp[i] = f32_to_f16_host(h[i]);
        HIPCHK(hipMemcpy(b.p, tmp.data(), n * 2, hipMemcpyHostToDevice));
    }
    return 0;
}

// Every live wm_state: a handle that is not in here (already freed — e.g. by wm_model_free, which owns the states created
// on it — or never valid) is rejected instead of dereferenced.
static std::mutex g_states_mu;
static std::unordered_set<const void*> g_live_states;
static bool state_is_live(const void* s) {
    std::lock_guard<std::mutex> lk(g_states_mu);
    return s && g_live_states.count(s) != 0;
}

struct EncLayer {
    DevBuf qkv_w, qkv_b, o_w, o_b, ln1_g, ln1_b, fc1_w, fc1_b, fc2_w, fc2_b, ln2_g, ln2_b;
};
struct DecLayer {
    DevBuf sqkv_w, sqkv_b, so_w, so_b, ln1_g, ln1_b, cq_w, cq_b, co_w, co_b, lnx_g, lnx_b, fc1_w, fc1_b, fc2_w, fc2_b, ln2_g,
        ln2_b;
};

struct wm_model {
    wm_config cfg;
    int device = 0;
    hipStream_t stream = nullptr;
    int Cp = 0;  // mel channels padded to a multiple of 32 (implicit-GEMM K)
    int K1 = 0;  // conv1 implicit-GEMM K (3*Cp rounded up to 64)
    int Vpad = 0;
    // utterances per encoder pass.  Measured (tiny, B=64, ms per 64 clips): 4 -> 12.0, 8 -> 9.0, 16 -> 7.6, 32 -> 6.8,
    // 64 -> 6.7: filling the chip (>= 4 tiles per CU per launch) matters more than keeping activations in the 256 MB L3
    int enc_chunk = 64;  // (re-measured with the row-panel GEMMs: 16 -> 6.1, 32 -> 5.0, 64 -> 4.8 ms)
    DevBuf conv1_w, conv1_b, conv2_w, conv2_b, enc_pos;
    std::vector<EncLayer> enc;
    DevBuf enc_ln_g, enc_ln_b;
    DevBuf tok_emb_f, tok_emb_t, dec_pos;
    std::vector<DecLayer> dec;
    DevBuf dec_ln_g, dec_ln_b;
    DevBuf cross_kv_w, cross_kv_b;  // [L*2*d][d] rows: layer-major, K then V
    DevBuf ts_buf;     // developer timeline (WM_TRACE_EVENTS=<file>): count + (tag, clock) pairs
    // log-mel front end (lazy): constants + scratch sized for max_batch utterances
    struct Frontend {
        bool ready = false;
        int chunk = 16;  // utterances per DFT GEMM
        DevBuf window, dft, fb, band, pcm, lens, frames, spec, logtmp, mel;
    } fe;
    static const int NSLOT = 8;
    wm_state* cached = nullptr;           // slot 0: the state behind wm_transcribe / wm_transcribe_submit(slot 0)
    wm_state* slots[NSLOT - 1] = {};  // slots 1..7: further pipeline stages (wm_transcribe_submit)
    std::unordered_set<wm_state*> states;  // every state created on this model (wm_state_new), freed with it
    // Coalescing (wm_config.coalesce == 2): two consecutive wm_transcribe_submit calls of the same batch size and options share ONE
    // decode state of 2·B rows — the 33 latency-bound launches of a decode step cost the same for 128 rows as for 64.  The first
    // call of a pair is held (`held`) until its partner arrives (or until it is waited for: then it runs alone); each call still
    // returns exactly its own ids.
    struct Held {
        bool active = false;
        int slot = 0, B = 0, on_dev = 0;
        const float* mel = nullptr;
        wm_decode_opts o{};
        std::vector<int32_t> prompt, sup, bsup;  // deep copies: the caller's option arrays need not outlive the call
    } held;
    struct SlotRef {  // where a submitted slot's rows live
        bool pending = false;
        wm_state* st = nullptr;  // null while the slot is only held
        int row0 = 0, rows = 0, total = 0;
    } slot_ref[8];
    wm_state* pairs[4] = {};  // 2·B-row states of coalesced pairs
    int last_steps[8] = {-1, -1, -1, -1, -1, -1, -1, -1};  // loop iterations enqueued for each slot's last collected pass
    // Loop pump (natural-stop passes of wm_transcribe_submit): one host thread per model that keeps each pending pass's greedy loop
    // two sub-chunks ahead of the GPU and stops enqueueing once the device reports every utterance finished (whisper.mojo:206-207).
    std::thread pump;
    std::mutex pump_mu;
    std::condition_variable pump_cv;
    std::vector<wm_state*> pump_work;  // passes whose loop is not fully enqueued yet
    bool pump_quit = false;
};

struct wm_state {
    wm_model* m = nullptr;
    int B = 0;
    int Bc = 0;  // encoder chunk
    int nsplit = 1;
    int out_stride = 0;
    bool has_enc = false, has_cross = false;
    int host_len = 0;
    // Decode lanes: the batch is cut into independent sub-batches, each decoding on its own HIP stream with its own
    // control block and captured step graph, so one lane's latency-bound launches overlap another's K/V streaming.
    struct Lane {
        int b0 = 0, nb = 0;
        hipStream_t st = nullptr;
        StepCtl* ctl = nullptr;
        static const int NEXEC = 3;  // instances of the same captured step, launched round-robin
        hipGraphExec_t graph[3] = {nullptr, nullptr, nullptr};
        hipEvent_t done = nullptr;
    };
    std::vector<Lane> lanes;
    hipEvent_t enc_done = nullptr;
    hipStream_t enc_stream = nullptr;  // stream the pending pass's encoder was enqueued on
    int graph_eot = 0, graph_ignore = 0;
    bool graphs_valid = false;
    bool pending = false;   // a submitted pass has not been waited for yet
    int halves_left = 0;    // coalesced pair: slots that have not collected their rows yet
    bool synced = false;    // the pending pass's completion has been waited for (second half of a pair does not wait again)
    bool shares_chip = false;  // this state's passes run beside other passes (pipelined entry): K/V stream at two workgroups per CU
    bool graph_shares = false;
    int trace_id = 1;       // slot + 1: tags this state's entries in the developer timeline
    int pend_total = 0;     // ids per utterance of the pending pass
    // Greedy loop of the pending pass, enqueued in sub-chunks of LOOP_CHUNK steps with at most two sub-chunks queued ahead of the GPU
    // (natural stop only; a fixed-length pass is enqueued whole).  h_prog: pinned, device-mapped [finished utterances, cache length],
    // written by every step's argmax launch.
    volatile int* h_prog = nullptr;
    int* d_prog = nullptr;
    int loop_total = 0, loop_enq = 0;  // steps the loop may run / steps enqueued so far
    int chunk_k = 0;                   // sub-chunks enqueued
    hipEvent_t chunk_ev[2] = {nullptr, nullptr};
    std::atomic<bool> enq_done{true};  // the pending pass is fully enqueued (its `done` events are recorded)
    int enq_rc = 0;                    // status of the pump's enqueues
    std::string enq_err;
    int last_steps = -1;               // loop steps enqueued for the most recent completed pass (wm_transcribe_steps)
    struct LoopOpts {
        int eot = 0, ignore_eot = 0;
        TsRules rules{};
    } loop_opts;
    const float* last_mel = nullptr;  // device pointer of the last encoded batch (bench replays the encoder on it)
    // encoder arena (sized for Bc utterances)
    DevBuf mel_dev, mel_t, h1, x, xn, qkv, ao, hid, enc_t;
    DevBuf enc_f;            // [B*n_ctx][d] fp32
    DevBuf cross_kv;         // [L][2][B][n_ctx][d] kv dtype
    DevBuf self_kv;          // [L][2][B][n_text_ctx][d]
    // decode arena
    DevBuf mask_steady, mask_begin;  // [Vpad] additive logit masks (0 / -inf) for the fused argmax
    std::vector<int32_t> sup_cached, bsup_cached;
    bool masks_valid = false;
    static const int PREFILL_MAX = 16;  // prompt positions decoded in one pass (= the n_prompt bound of wm_decode_opts)
    DevBuf ts_state, ts_val, ts_idx, ts_m, ts_s;  // timestamp rules: per-utterance history / ranges, per-part timestamp partials
    TsRules graph_rules{};                         // rules baked into the captured step graph
    int no_ts_cached = -1;
    DevBuf dx, dq, dattn, dhid, part_o, part_ml, logits, amax_val, amax_idx, tok, pos, tok_rows, pos_rows, ctl, out_tokens, n_tokens, finished;
    int npart = 0;  // fused-argmax partials per utterance = workgroups per row block of the logits kernel
};

// ------------------------------------------------------------------------------------------------------------
#define DISPATCH_DT(dt, T, ...)              \
    do {                                     \
        if ((dt) == WM_F32) {                \
            typedef float T;                 \
            __VA_ARGS__;                     \
        } else if ((dt) == WM_BF16) {        \
            typedef wm::bf16 T;              \
            __VA_ARGS__;                     \
        } else {                             \
            typedef wm::f16 T;               \
            __VA_ARGS__;                     \
        }                                    \
    } while (0)

static int gemm_dispatch(int dt_in, int dt_out, const GemmParams& p, int batch, hipStream_t st);
// C = LN(x) W^T (+ epilogue of p): p.A names the scratch for the normalised rows.  The A-stationary GEMM normalises the fp32 rows
// while it loads them (no LayerNorm launch, no 16-bit copy through HBM); every other shape runs layernorm_rows first.
static int ln_then_gemm(int dt_in, int dt_out, GemmParams p, const float* x, const float* g, const float* b, hipStream_t st) {
    if (gemm_nt_fuses_layernorm(dt_in == WM_F32 ? 4 : 2, p, 1)) {
        p.A = x;
        p.ln_g = g;
        p.ln_b = b;
    } else {
        DISPATCH_DT(dt_in, TT, launch_layernorm_rows<TT>(x, g, b, const_cast<void*>(p.A), nullptr, p.M, p.K, 1e-5f, st));
    }
    return gemm_dispatch(dt_in, dt_out, p, 1, st);
}
// 0 or a WM_E_* status with wm_last_error set (a shape the kernels refuse: nothing was launched)
static int gemm_dispatch(int dt_in, int dt_out, const GemmParams& p, int batch, hipStream_t st) {
    int rc;
    if (dt_in == WM_F32) {
        rc = launch_gemm_nt<float, float>(p, batch, st);
    } else if (dt_in == WM_BF16) {
        rc = dt_out == WM_F32 ? launch_gemm_nt<bf16, float>(p, batch, st) : launch_gemm_nt<bf16, bf16>(p, batch, st);
    } else {
        rc = dt_out == WM_F32 ? launch_gemm_nt<f16, float>(p, batch, st) : launch_gemm_nt<f16, f16>(p, batch, st);
    }
    return launch_rc(rc);
}

extern "C" size_t wm_weight_count(const wm_dims* c) { return wm_synth_count(c); }
extern "C" size_t wm_synth_weights(const wm_dims* dims, uint64_t seed, float* out) { return wm_synth_fill(dims, seed, out); }
extern "C" void wm_synth_mel_host(uint64_t seed, int n_mels, int n_frames, float* out) {
    wm_synth_mel(seed, n_mels, n_frames, out);
}

static int check_cfg(const wm_config* c) {
    const wm_dims& d = c->dims;
    if (d.d_model <= 0 || d.n_heads <= 0 || d.d_model != d.n_heads * 64)
        return fail(WM_E_ARG, "d_model must equal n_heads*64 (got %d, %d)", d.d_model, d.n_heads);
    if (d.n_heads > 8) return fail(WM_E_ARG, "n_heads > 8 not supported");
    if (d.d_model % 128 || d.ffn % 128) return fail(WM_E_ARG, "d_model and ffn must be multiples of 128");
    // what the decode kernels are instantiated for: the logits kernel keeps d_model/128 in {1, 3, 4} k-panels
    // (kernels_decoder.hip launch_dec_logits), the skinny linears split K/32 k-steps over <= 16 waves x <= 4 steps
    if (d.d_model != 128 && d.d_model != 384 && d.d_model != 512)
        return fail(WM_E_ARG, "d_model %d not supported (128, 384 or 512)", d.d_model);
    if (!dec_linear_supports_k(d.d_model) || !dec_linear_supports_k(d.ffn))
        return fail(WM_E_ARG, "ffn %d not supported (ffn/32 must split into <= 16 waves x <= 4 k-steps, ffn <= 2048)", d.ffn);
    if ((d.n_layers * 2 * d.d_model) % 128) return fail(WM_E_ARG, "n_layers*2*d_model must be a multiple of 128");
    if (d.n_text_ctx > 512) return fail(WM_E_ARG, "n_text_ctx > 512 not supported");
    if (d.n_mels <= 0 || d.n_audio_ctx <= 0 || d.vocab <= 0 || d.n_layers <= 0) return fail(WM_E_ARG, "bad dims");
    if (c->compute_dtype < 0 || c->compute_dtype > 2) return fail(WM_E_ARG, "bad compute_dtype");
    if (!(c->kv_dtype == WM_F32 || c->kv_dtype == c->compute_dtype))
        return fail(WM_E_ARG, "kv_dtype must be WM_F32 or equal compute_dtype");
    if (c->decoder_fp32 != 0 && c->decoder_fp32 != 1) return fail(WM_E_ARG, "decoder_fp32 must be 0 or 1");
    if (c->coalesce < 0 || c->coalesce > 2) return fail(WM_E_ARG, "coalesce must be 0, 1 (both: off) or 2");
    if (c->max_batch <= 0) return fail(WM_E_ARG, "max_batch must be > 0");
    if (c->gelu_mode != 0 && c->gelu_mode != 1) return fail(WM_E_ARG, "bad gelu_mode");
    return 0;
}

// ---- model load: loader.mojo:10-27 + whisper.mojo:60-69,122-128 + layers.mojo:96-103,418-433 -------------------
struct Reader {
    const float* p;
    size_t off = 0;
    const float* take(size_t n) {
        const float* r = p + off;
        off += n;
        return r;
    }
};
struct AttnW {
    const float *q_w, *q_b, *k_w, *v_w, *v_b, *o_w, *o_b;
};
static AttnW read_attn(Reader& r, size_t d) {
    AttnW a;
    a.q_w = r.take(d * d);
    a.q_b = r.take(d);
    a.k_w = r.take(d * d);
    a.v_w = r.take(d * d);
    a.v_b = r.take(d);
    a.o_w = r.take(d * d);
    a.o_b = r.take(d);
    return a;
}
// [co][ci][3] -> [co][3][ci_pad]   (transpose_conv_weights, whisper_tensor.mojo:358-364, plus channel padding)
static std::vector<float> conv_relayout(const float* w, int co, int ci, int cip) {
    std::vector<float> o((size_t)co * 3 * cip, 0.f);
    for (int a = 0; a < co; ++a)
        for (int c = 0; c < ci; ++c)
            for (int k = 0; k < 3; ++k) o[((size_t)a * 3 + k) * cip + c] = w[(size_t)a * ci * 3 + (size_t)c * 3 + k];
    return o;
}

extern "C" void wm_state_free(wm_state* s);

extern "C" void wm_model_free(wm_model* m) {
    if (m && trace_events_on()) {
        trace_dump();
        {
            const char* path = wm_env("WM_TRACE_EVENTS");
            if (m->ts_buf.p && path) {
                std::vector<long long> h((2u << 20) + 1);
                if (hipMemcpy(h.data(), m->ts_buf.p, h.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
                    if (FILE* f = fopen(path, "w")) {
                        const long long n = std::min<long long>(h[0], 1ll << 20);
                        for (long long i = 0; i < n; ++i) fprintf(f, "%lld %lld %lld\n", h[1 + 2 * i] >> 8, h[1 + 2 * i] & 255, h[2 + 2 * i]);
                        fclose(f);
                    }
                }
            }
        }
    }
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->pump.joinable()) {
        {
            std::lock_guard<std::mutex> lk(m->pump_mu);
            m->pump_quit = true;
        }
        m->pump_cv.notify_all();
        m->pump.join();
    }
    // the model owns every state created on it (pipeline slots and the caller's KVCaches): a handle the caller still holds
    // becomes stale — wm_state_free / wm_decode_step on it is a checked no-op / error, not a use after free
    while (!m->states.empty()) wm_state_free(*m->states.begin());
    {
        DevBuf* fb[] = {&m->fe.window, &m->fe.dft, &m->fe.fb, &m->fe.band, &m->fe.pcm, &m->fe.lens, &m->fe.frames, &m->fe.spec, &m->fe.logtmp, &m->fe.mel};
        for (DevBuf* b : fb) b->release();
    }
    DevBuf* top[] = {&m->conv1_w, &m->conv1_b, &m->conv2_w, &m->conv2_b, &m->enc_pos, &m->enc_ln_g, &m->enc_ln_b,
                     &m->tok_emb_f, &m->tok_emb_t, &m->dec_pos, &m->dec_ln_g, &m->dec_ln_b, &m->cross_kv_w, &m->cross_kv_b};
    for (DevBuf* b : top) b->release();
    for (EncLayer& l : m->enc) {
        DevBuf* bs[] = {&l.qkv_w, &l.qkv_b, &l.o_w, &l.o_b, &l.ln1_g, &l.ln1_b, &l.fc1_w, &l.fc1_b, &l.fc2_w, &l.fc2_b, &l.ln2_g, &l.ln2_b};
        for (DevBuf* b : bs) b->release();
    }
    for (DecLayer& l : m->dec) {
        DevBuf* bs[] = {&l.sqkv_w, &l.sqkv_b, &l.so_w, &l.so_b, &l.ln1_g, &l.ln1_b, &l.cq_w, &l.cq_b, &l.co_w, &l.co_b,
                        &l.lnx_g, &l.lnx_b, &l.fc1_w, &l.fc1_b, &l.fc2_w, &l.fc2_b, &l.ln2_g, &l.ln2_b};
        for (DevBuf* b : bs) b->release();
    }
    if (m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
}

static int model_build(wm_model* m, const float* w) {
    const wm_dims& c = m->cfg.dims;
    const size_t d = c.d_model, f = c.ffn;
    const int T = m->cfg.compute_dtype;  // encoder-side operands (incl. the cross-K/V projection of the encoder output)
    const int TD = dec_dtype(m->cfg);    // decoder-side operands
    Reader r{w};
    m->Cp = (c.n_mels + 31) / 32 * 32;
    m->Vpad = (c.vocab + 15) / 16 * 16;
    {
        // conv1 as implicit GEMM: K = 3*Cp, rounded up to a multiple of 64 with zero weights (the extra window elements
        // read the start of the next token row and are multiplied by 0) so the 16-bit path takes the LDS-staged kernel
        auto c1r = conv_relayout(r.take(d * c.n_mels * 3), c.d_model, c.n_mels, m->Cp);
        m->K1 = (3 * m->Cp + 63) / 64 * 64;
        std::vector<float> c1((size_t)c.d_model * m->K1, 0.f);
        for (int a = 0; a < c.d_model; ++a) memcpy(&c1[(size_t)a * m->K1], &c1r[(size_t)a * 3 * m->Cp], (size_t)3 * m->Cp * 4);
        WMCHK(upload(m->conv1_w, c1.data(), c1.size(), T));
        WMCHK(upload(m->conv1_b, r.take(d), d, WM_F32));
        auto c2 = conv_relayout(r.take(d * d * 3), c.d_model, c.d_model, c.d_model);
        WMCHK(upload(m->conv2_w, c2.data(), c2.size(), T));
        WMCHK(upload(m->conv2_b, r.take(d), d, WM_F32));
        WMCHK(upload(m->enc_pos, r.take((size_t)c.n_audio_ctx * d), (size_t)c.n_audio_ctx * d, WM_F32));
    }
    std::vector<float> qkv(3 * d * d), qkvb(3 * d);
    auto pack_qkv = [&](const AttnW& a) {
        memcpy(qkv.data(), a.q_w, d * d * 4);
        memcpy(qkv.data() + d * d, a.k_w, d * d * 4);
        memcpy(qkv.data() + 2 * d * d, a.v_w, d * d * 4);
        memcpy(qkvb.data(), a.q_b, d * 4);
        memset(qkvb.data() + d, 0, d * 4);  // k_proj has no bias (layers.mojo:96-103)
        memcpy(qkvb.data() + 2 * d, a.v_b, d * 4);
    };
    m->enc.resize(c.n_layers);
    for (int i = 0; i < c.n_layers; ++i) {
        EncLayer& l = m->enc[i];
        AttnW a = read_attn(r, d);
        pack_qkv(a);
        WMCHK(upload(l.qkv_w, qkv.data(), qkv.size(), T));
        WMCHK(upload(l.qkv_b, qkvb.data(), qkvb.size(), WM_F32));
        WMCHK(upload(l.o_w, a.o_w, d * d, T));
        WMCHK(upload(l.o_b, a.o_b, d, WM_F32));
        WMCHK(upload(l.ln1_g, r.take(d), d, WM_F32));
        WMCHK(upload(l.ln1_b, r.take(d), d, WM_F32));
        WMCHK(upload(l.fc1_w, r.take(f * d), f * d, T));
        WMCHK(upload(l.fc1_b, r.take(f), f, WM_F32));
        WMCHK(upload(l.fc2_w, r.take(d * f), d * f, T));
        WMCHK(upload(l.fc2_b, r.take(d), d, WM_F32));
        WMCHK(upload(l.ln2_g, r.take(d), d, WM_F32));
        WMCHK(upload(l.ln2_b, r.take(d), d, WM_F32));
    }
    WMCHK(upload(m->enc_ln_g, r.take(d), d, WM_F32));
    WMCHK(upload(m->enc_ln_b, r.take(d), d, WM_F32));
    {
        const float* te = r.take((size_t)c.vocab * d);
        WMCHK(upload(m->tok_emb_f, te, (size_t)c.vocab * d, WM_F32));
        if (TD != WM_F32) WMCHK(upload(m->tok_emb_t, te, (size_t)c.vocab * d, TD));
        WMCHK(upload(m->dec_pos, r.take((size_t)c.n_text_ctx * d), (size_t)c.n_text_ctx * d, WM_F32));
    }
    std::vector<float> ckv((size_t)c.n_layers * 2 * d * d), ckvb((size_t)c.n_layers * 2 * d, 0.f);
    m->dec.resize(c.n_layers);
    for (int i = 0; i < c.n_layers; ++i) {
        DecLayer& l = m->dec[i];
        AttnW a = read_attn(r, d);
        pack_qkv(a);
        WMCHK(upload(l.sqkv_w, qkv.data(), qkv.size(), TD));
        WMCHK(upload(l.sqkv_b, qkvb.data(), qkvb.size(), WM_F32));
        WMCHK(upload(l.so_w, a.o_w, d * d, TD));
        WMCHK(upload(l.so_b, a.o_b, d, WM_F32));
        WMCHK(upload(l.ln1_g, r.take(d), d, WM_F32));
        WMCHK(upload(l.ln1_b, r.take(d), d, WM_F32));
        AttnW x = read_attn(r, d);
        WMCHK(upload(l.cq_w, x.q_w, d * d, TD));
        WMCHK(upload(l.cq_b, x.q_b, d, WM_F32));
        memcpy(ckv.data() + (size_t)(2 * i) * d * d, x.k_w, d * d * 4);
        memcpy(ckv.data() + (size_t)(2 * i + 1) * d * d, x.v_w, d * d * 4);
        memcpy(ckvb.data() + (size_t)(2 * i + 1) * d, x.v_b, d * 4);
        WMCHK(upload(l.co_w, x.o_w, d * d, TD));
        WMCHK(upload(l.co_b, x.o_b, d, WM_F32));
        WMCHK(upload(l.lnx_g, r.take(d), d, WM_F32));
        WMCHK(upload(l.lnx_b, r.take(d), d, WM_F32));
        WMCHK(upload(l.fc1_w, r.take(f * d), f * d, TD));
        WMCHK(upload(l.fc1_b, r.take(f), f, WM_F32));
        WMCHK(upload(l.fc2_w, r.take(d * f), d * f, TD));
        WMCHK(upload(l.fc2_b, r.take(d), d, WM_F32));
        WMCHK(upload(l.ln2_g, r.take(d), d, WM_F32));
        WMCHK(upload(l.ln2_b, r.take(d), d, WM_F32));
    }
    WMCHK(upload(m->dec_ln_g, r.take(d), d, WM_F32));
    WMCHK(upload(m->dec_ln_b, r.take(d), d, WM_F32));
    WMCHK(upload(m->cross_kv_w, ckv.data(), ckv.size(), T));
    WMCHK(upload(m->cross_kv_b, ckvb.data(), ckvb.size(), WM_F32));
    if (const char* tr = wm_env("WM_TRACE_EVENTS"); tr && strchr(tr, '/')) WMCHK(m->ts_buf.alloc(((2u << 20) + 1) * 8, true));
    if (r.off != wm_synth_count(&c)) return fail(WM_E_SIZE, "internal: consumed %zu floats, expected %zu", r.off, wm_synth_count(&c));
    return 0;
}

extern "C" int wm_model_load_memory(const float* weights, size_t n_floats, const wm_config* cfg, int device, wm_model** out) {
    if (!weights || !cfg || !out) return fail(WM_E_ARG, "null argument");
    WMCHK(check_cfg(cfg));
    const size_t want = wm_synth_count(&cfg->dims);
    if (n_floats != want)
        return fail(WM_E_SIZE, "weight image holds %zu floats (%zu bytes), this config needs %zu (%zu bytes)", n_floats,
                    n_floats * 4, want, want * 4);
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(WM_E_ARG, "device %d out of range (%d visible)", device, ndev);
    HIPCHK(hipSetDevice(device));
    wm_model* m = new wm_model();
    m->cfg = *cfg;
    m->device = device;
    if (const char* e = wm_env("WM_ENC_CHUNK")) m->enc_chunk = std::max(1, atoi(e));
    hipError_t e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete m;
        return fail(WM_E_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    int rc = model_build(m, weights);
    if (rc) {
        std::string keep = g_err;
        wm_model_free(m);
        g_err = keep;
        return rc;
    }
    *out = m;
    return 0;
}

// ---- weight file formats ------------------------------------------------------------------------------------------
// v1: the reference's headerless fp32 dump (export_weights.py:19-90; loader.mojo reads it with no size check).
// v2: 64-byte header {magic "WMIWGT2", version, matrix dtype, dims, flags, payload bytes} + the same
// tensors in the same order; conv / linear matrices (and, unless flag bit 0 is set, the token embedding) stored in the
// matrix dtype, every vector and positional table in fp32.  Loading v2 expands to the fp32 image the builder takes, so a
// v2 file in the model's compute dtype gives bit-identical weights to the v1 file (16-bit rounding is idempotent).
struct WmV2Header {
    char magic[8];
    uint32_t version, matrix_dtype;
    int32_t dims[8];
    uint32_t flags, reserved;
    uint64_t payload_bytes;
};
static_assert(sizeof(WmV2Header) == 64, "v2 header is 64 bytes");
static const char WM_V2_MAGIC[8] = {'W', 'M', 'I', 'W', 'G', 'T', '2', '\0'};
enum { WM_V2_EMB_F32 = 1 };

struct TensorSpan {
    int kind;
    size_t count;
};
static void collect_tensor(void* user, int, int kind, size_t count) { ((std::vector<TensorSpan>*)user)->push_back({kind, count}); }
static bool v2_is_matrix(int kind, uint32_t flags) { return kind == WM_K_WEIGHT || kind == WM_K_QK || (kind == WM_K_EMB && !(flags & WM_V2_EMB_F32)); }
static size_t v2_payload_bytes(const wm_dims* d, int dtype, uint32_t flags) {
    std::vector<TensorSpan> t;
    wm_synth_walk(d, collect_tensor, &t);
    size_t n = 0;
    for (auto& s : t) n += s.count * (v2_is_matrix(s.kind, flags) ? dt_size(dtype) : 4);
    return n;
}
static inline float bf16_to_f32_host(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline float f16_to_f32_host(uint16_t h) {
    _Float16 x;
    memcpy(&x, &h, 2);
    return (float)x;
}

extern "C" int wm_weights_convert_v2(const char* v1_path, const char* v2_path, const wm_dims* dims, int dtype, int emb_f32) {
    if (!v1_path || !v2_path || !dims || dtype < 0 || dtype > 2) return fail(WM_E_ARG, "bad argument");
    const size_t n = wm_synth_count(dims);
    FILE* f = fopen(v1_path, "rb");
    if (!f) return fail(WM_E_IO, "cannot open %s", v1_path);
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (sz < 0 || (size_t)sz != n * 4) {
        fclose(f);
        return fail(WM_E_SIZE, "%s is %ld bytes, these dims need %zu", v1_path, sz, n * 4);
    }
    std::vector<float> w(n);
    const size_t got = fread(w.data(), 4, n, f);
    fclose(f);
    if (got != n) return fail(WM_E_IO, "short read on %s", v1_path);
    WmV2Header h{};
    memcpy(h.magic, WM_V2_MAGIC, 8);
    h.version = 2;
    h.matrix_dtype = (uint32_t)dtype;
    memcpy(h.dims, dims, sizeof h.dims);
    h.flags = emb_f32 ? WM_V2_EMB_F32 : 0;
    h.payload_bytes = v2_payload_bytes(dims, dtype, h.flags);
    FILE* o = fopen(v2_path, "wb");
    if (!o) return fail(WM_E_IO, "cannot create %s", v2_path);
    bool ok = fwrite(&h, sizeof h, 1, o) == 1;
    std::vector<TensorSpan> t;
    wm_synth_walk(dims, collect_tensor, &t);
    size_t off = 0;
    std::vector<uint16_t> tmp;
    for (auto& s : t) {
        if (v2_is_matrix(s.kind, h.flags) && dtype != WM_F32) {
            tmp.resize(s.count);
            for (size_t i = 0; i < s.count; ++i) tmp[i] = dtype == WM_BF16 ? f32_to_bf16_host(w[off + i]) : f32_to_f16_host(w[off + i]);
            ok = ok && fwrite(tmp.data(), 2, s.count, o) == s.count;
        } else {
            ok = ok && fwrite(w.data() + off, 4, s.count, o) == s.count;
        }
        off += s.count;
    }
    ok = (fclose(o) == 0) && ok;
    return ok ? 0 : fail(WM_E_IO, "write error on %s", v2_path);
}

// Reads a v1 or v2 file into an fp32 image of wm_weight_count(dims) floats, validating size / header against dims.
extern "C" int wm_weights_read(const char* path, const wm_dims* dims, float* out) {
    if (!path || !dims || !out) return fail(WM_E_ARG, "null argument");
    const size_t n = wm_synth_count(dims);
    FILE* f = fopen(path, "rb");
    if (!f) return fail(WM_E_IO, "cannot open %s", path);
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    WmV2Header h{};
    const bool v2 = sz >= (long)sizeof h && fread(&h, sizeof h, 1, f) == 1 && memcmp(h.magic, WM_V2_MAGIC, 8) == 0;
    if (!v2) {
        fseek(f, 0, SEEK_SET);
        if (sz < 0 || (size_t)sz != n * 4) {
            fclose(f);
            return fail(WM_E_SIZE, "%s is %ld bytes, this config needs %zu", path, sz, n * 4);
        }
        const size_t got = fread(out, 4, n, f);
        fclose(f);
        return got == n ? 0 : fail(WM_E_IO, "short read on %s", path);
    }
    int rc = 0;
    if (h.version != 2 || h.matrix_dtype > 2)
        rc = fail(WM_E_SIZE, "%s: unsupported v2 header (version %u, dtype %u)", path, h.version, h.matrix_dtype);
    else if (memcmp(h.dims, dims, sizeof h.dims) != 0)
        rc = fail(WM_E_SIZE, "%s was written for other dims (d_model %d, layers %d, vocab %d)", path, h.dims[0], h.dims[2], h.dims[7]);
    else if (h.payload_bytes != v2_payload_bytes(dims, (int)h.matrix_dtype, h.flags) || (size_t)sz != sizeof h + h.payload_bytes)
        rc = fail(WM_E_SIZE, "%s: payload size does not match its header", path);
    if (!rc) {
        std::vector<TensorSpan> t;
        wm_synth_walk(dims, collect_tensor, &t);
        size_t off = 0;
        std::vector<uint16_t> tmp;
        for (auto& s : t) {
            if (v2_is_matrix(s.kind, h.flags) && h.matrix_dtype != WM_F32) {
                tmp.resize(s.count);
                if (fread(tmp.data(), 2, s.count, f) != s.count) {
                    rc = fail(WM_E_IO, "short read on %s", path);
                    break;
                }
                for (size_t i = 0; i < s.count; ++i) out[off + i] = h.matrix_dtype == WM_BF16 ? bf16_to_f32_host(tmp[i]) : f16_to_f32_host(tmp[i]);
            } else if (fread(out + off, 4, s.count, f) != s.count) {
                rc = fail(WM_E_IO, "short read on %s", path);
                break;
            }
            off += s.count;
        }
    }
    fclose(f);
    return rc;
}

extern "C" int wm_model_load(const char* path, const wm_config* cfg, int device, wm_model** out) {
    if (!path || !cfg || !out) return fail(WM_E_ARG, "null argument");
    WMCHK(check_cfg(cfg));
    std::vector<float> buf(wm_synth_count(&cfg->dims));
    WMCHK(wm_weights_read(path, &cfg->dims, buf.data()));
    return wm_model_load_memory(buf.data(), buf.size(), cfg, device, out);
}

// ---- state ---------------------------------------------------------------------------------------------------
extern "C" void wm_state_free(wm_state* s) {
    if (!s) return;
    {
        std::lock_guard<std::mutex> lk(g_states_mu);
        if (!g_live_states.erase(s)) return;  // stale handle: its model was freed (and took the state with it)
    }
    {  // the loop pump must not touch this state any more (it works under this mutex)
        std::lock_guard<std::mutex> lk(s->m->pump_mu);
        auto& wq = s->m->pump_work;
        wq.erase(std::remove(wq.begin(), wq.end(), s), wq.end());
    }
    s->m->states.erase(s);
    if (s->m->cached == s) s->m->cached = nullptr;
    for (auto& sl : s->m->slots)
        if (sl == s) sl = nullptr;
    for (auto& pr : s->m->pairs)
        if (pr == s) pr = nullptr;
    for (auto& r : s->m->slot_ref)
        if (r.st == s) r = wm_model::SlotRef{};
    (void)hipSetDevice(s->m->device);
    // nothing of this state may still be running when its graphs, streams and arenas go away (a submitted pass that was
    // never waited for, or the model stream's last wm_decode_step)
    for (auto& ln : s->lanes)
        if (ln.st) (void)hipStreamSynchronize(ln.st);
    if (s->m->stream) (void)hipStreamSynchronize(s->m->stream);
    for (auto& ln : s->lanes) {
        for (auto& g : ln.graph)
            if (g) (void)hipGraphExecDestroy(g);
        if (ln.st) (void)hipStreamDestroy(ln.st);
        if (ln.done) (void)hipEventDestroy(ln.done);
    }
    if (s->enc_done) (void)hipEventDestroy(s->enc_done);
    for (auto& ev : s->chunk_ev)
        if (ev) (void)hipEventDestroy(ev);
    if (s->h_prog) (void)hipHostFree((void*)s->h_prog);
    DevBuf* bs[] = {&s->mel_dev, &s->mel_t, &s->h1, &s->x, &s->xn, &s->qkv, &s->ao, &s->hid, &s->enc_t, &s->enc_f,
                    &s->cross_kv, &s->self_kv, &s->dx, &s->dq, &s->dattn, &s->dhid, &s->part_o, &s->part_ml, &s->logits, &s->amax_val, &s->amax_idx, &s->ts_state, &s->ts_val, &s->ts_idx, &s->ts_m, &s->ts_s, &s->mask_steady, &s->mask_begin,
                    &s->tok, &s->pos, &s->tok_rows, &s->pos_rows, &s->ctl, &s->out_tokens, &s->n_tokens, &s->finished};
    for (DevBuf* b : bs) b->release();
    delete s;
}

static const int OUT_STRIDE_MAX = 1024;

static int state_new(wm_model* m, int B, wm_state** out, bool pair);
extern "C" int wm_state_new(wm_model* m, int B, wm_state** out) { return state_new(m, B, out, false); }
// pair: the 2·B-row state of a coalesced pair of submits (B <= max_batch each)
static int state_new(wm_model* m, int B, wm_state** out, bool pair) {
    if (!m || !out || B <= 0) return fail(WM_E_ARG, "bad argument");
    if (B > m->cfg.max_batch * (pair ? 2 : 1)) return fail(WM_E_ARG, "batch %d exceeds max_batch %d", B, m->cfg.max_batch);
    HIPCHK(hipSetDevice(m->device));
    const wm_dims& c = m->cfg.dims;
    const size_t d = c.d_model, L = 2 * (size_t)c.n_audio_ctx, T = c.n_audio_ctx;
    const size_t ts = dt_size(m->cfg.compute_dtype), ks = dt_size(m->cfg.kv_dtype);
    wm_state* s = new wm_state();
    s->m = m;
    s->B = B;
    {
        std::lock_guard<std::mutex> lk(g_states_mu);
        g_live_states.insert(s);
    }
    m->states.insert(s);
    s->Bc = std::min(pair ? B / 2 : B, m->enc_chunk);  // (pair: an encoder chunk never straddles the two batches)
    // key chunks per utterance for the cross-attention kernel: a function of the MODEL's max_batch, never of this call's
    // B, so that an utterance's result does not depend on how it was batched (bitwise batch invariance within a model).
    // Aim for 1024 workgroups at full batch (4 per CU, one full round): 16 chunks at max_batch 64 (measured, µs per launch /
    // per whole step: 12 chunks 26.4 / 280, 16: 25.2 / 273, 20: 25.5 / 278, 24: 26.4 / 278), up to 48 for small-batch /
    // latency models (B = 1: 12.1 -> 4 us per launch).  WM_NSPLIT overrides for tuning.
    {
        const int min_split = (int)((T + 511) / 512);
        // fp32 K/V rows are twice as wide: half the chunks move the same bytes per workgroup, the stream runs as fast (46.6 vs 46.5 us
        // per launch at B = 64) and the merge reads half the partials (decode step 406 -> 402 us; 10 / 12 chunks: 418 / 412 us)
        const int target = ks == 4 ? 512 : 1024;
        int ns = (target + m->cfg.max_batch - 1) / m->cfg.max_batch;
        ns = std::max(ks == 4 ? 8 : 12, std::min(48, ns));
        ns = std::max(min_split, std::min(ns, (int)((T + 31) / 32)));
        if (const char* e = wm_env("WM_NSPLIT")) ns = std::max(min_split, std::min(64, atoi(e)));
        s->nsplit = ns;
    }
    s->out_stride = OUT_STRIDE_MAX;
    const size_t Bc = s->Bc;
    const size_t Mp = (Bc * T + 255) / 256 * 256 + 256;  // padded rows: tail tiles (up to 256 rows) read, never store, past M
    int rc = 0;
    auto A = [&](DevBuf& b, size_t bytes, bool zero = false) {
        if (!rc) rc = b.alloc(bytes, zero);
    };
    A(s->mel_dev, (size_t)B * c.n_mels * L * 4);
    A(s->mel_t, (Bc * (L + 2) + 256) * m->Cp * ts, true);
    A(s->h1, (Bc * (L + 2) + 256) * d * ts, true);  // rows 0 and L+1 of each utterance stay zero = conv padding
    A(s->x, Mp * d * 4, true);
    A(s->xn, Mp * d * ts, true);
    A(s->qkv, Mp * 3 * d * ts, true);
    A(s->ao, Mp * d * ts, true);
    A(s->hid, Mp * c.ffn * ts, true);
    A(s->enc_t, Mp * d * ts, true);
    A(s->enc_f, (size_t)B * T * d * 4);
    A(s->cross_kv, (size_t)c.n_layers * 2 * B * T * d * ks);
    A(s->self_kv, (size_t)c.n_layers * 2 * B * c.n_text_ctx * d * ks, true);
    // decode activations: rows for one token per utterance, or — prompt prefill — for up to PREFILL_MAX positions at once
    const size_t R = (size_t)B * wm_state::PREFILL_MAX;
    A(s->dx, R * d * 4);
    A(s->dq, R * d * 4);
    A(s->dattn, R * d * 4);
    A(s->dhid, R * c.ffn * 4);
    A(s->part_o, R * s->nsplit * d * 4);
    A(s->part_ml, R * s->nsplit * c.n_heads * 2 * 4);
    A(s->tok_rows, R * 4, true);
    A(s->pos_rows, R * 4, true);
    A(s->logits, (size_t)B * m->Vpad * 4);
    s->npart = dec_logits_parts(c.vocab);
    A(s->amax_val, (size_t)B * s->npart * 4);
    A(s->amax_idx, (size_t)B * s->npart * 4);
    A(s->ts_val, (size_t)B * s->npart * 4, true);
    A(s->ts_idx, (size_t)B * s->npart * 4, true);
    A(s->ts_m, (size_t)B * s->npart * 4, true);
    A(s->ts_s, (size_t)B * s->npart * 4, true);
    A(s->ts_state, (size_t)B * sizeof(TsState), true);
    A(s->mask_steady, (size_t)m->Vpad * 4, true);
    A(s->mask_begin, (size_t)m->Vpad * 4, true);
    A(s->tok, (size_t)B * 4, true);
    A(s->pos, (size_t)B * 4, true);
    A(s->ctl, sizeof(StepCtl) * 8, true);  // [0] whole-batch control (wm_decode_step), [1..] one per decode lane
    A(s->out_tokens, (size_t)B * s->out_stride * 4, true);
    A(s->n_tokens, (size_t)B * 4, true);
    A(s->finished, (size_t)B * 4, true);
    if (rc) {
        std::string keep = g_err;
        wm_state_free(s);
        g_err = keep;
        return rc;
    }
    {  // decode lanes
        // measured on MI355X (round 1): 2 lanes 49.1 ms vs 1 lane 47.4 ms per 64-clip pass — kernel boundaries of one
        // queue also stall the other queue's kernels, so extra lanes stay opt-in (WM_DEC_LANES)
        int nl = 1;
        if (const char* e = wm_env("WM_DEC_LANES")) nl = std::max(1, std::min(4, atoi(e)));
        nl = std::min(nl, (B + 15) / 16);
        s->lanes.resize(nl);
        const int per = ((B + nl - 1) / nl + 15) / 16 * 16;  // whole MFMA row blocks per lane
        int b0 = 0;
        for (int i = 0; i < nl; ++i) {
            wm_state::Lane& ln = s->lanes[i];
            ln.b0 = b0;
            ln.nb = std::max(0, std::min(per, B - b0));
            b0 += ln.nb;
            ln.ctl = s->ctl.as<StepCtl>() + 1 + i;
            // decode launches are latency-critical (34 dependent sub-5-us kernels per token); the encoder stream carries
            // throughput work.  WM_DEC_PRIORITY=1 puts the lane streams on the highest HIP stream priority.
            int lo_p = 0, hi_p = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo_p, &hi_p);
            static const bool prio = wm_env("WM_DEC_PRIORITY") != nullptr;
            hipError_t e = prio ? hipStreamCreateWithPriority(&ln.st, hipStreamNonBlocking, hi_p)
                                : hipStreamCreateWithFlags(&ln.st, hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ln.done, hipEventDisableTiming);
            if (e != hipSuccess) {
                wm_state_free(s);
                return fail(WM_E_HIP, "lane stream: %s", hipGetErrorString(e));
            }
        }
        while (!s->lanes.empty() && s->lanes.back().nb == 0) {
            (void)hipStreamDestroy(s->lanes.back().st);
            (void)hipEventDestroy(s->lanes.back().done);
            s->lanes.pop_back();
        }
        hipError_t e = hipEventCreateWithFlags(&s->enc_done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipHostMalloc((void**)&s->h_prog, 64, hipHostMallocMapped);
        if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&s->d_prog, (void*)s->h_prog, 0);
        for (auto& ev : s->chunk_ev)
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e != hipSuccess) {
            wm_state_free(s);
            return fail(WM_E_HIP, "event: %s", hipGetErrorString(e));
        }
    }
    *out = s;
    return 0;
}

extern "C" int wm_state_reset(wm_state* s) {
    if (!state_is_live(s)) return fail(WM_E_ARG, "null or stale state handle");
    HIPCHK(hipSetDevice(s->m->device));
    hipStream_t st = s->m->stream;
    HIPCHK(hipMemsetAsync(s->ctl.p, 0, s->ctl.bytes, st));
    HIPCHK(hipMemsetAsync(s->n_tokens.p, 0, s->n_tokens.bytes, st));
    HIPCHK(hipMemsetAsync(s->finished.p, 0, s->finished.bytes, st));
    s->has_enc = s->has_cross = false;
    s->host_len = 0;
    return 0;
}
extern "C" int wm_state_len(const wm_state* s) { return state_is_live(s) ? s->host_len : -1; }

// ---- encoder: whisper.mojo:71-99 ------------------------------------------------------------------------------------
static void* off_bytes(const DevBuf& b, size_t bytes) { return (char*)b.p + bytes; }

static int cross_kv_chunk(wm_model* m, wm_state* s, int c0, int bc, hipStream_t st) {
    // cross K/V for utterances [c0, c0+bc) from enc_t (rows 0..bc*T): layers.mojo:150-154, all layers in one GEMM
    const wm_dims& c = m->cfg.dims;
    const size_t d = c.d_model, T = c.n_audio_ctx;
    GemmParams p{};
    p.A = s->enc_t.p;
    p.W = m->cross_kv_w.p;
    p.C = off_bytes(s->cross_kv, (size_t)c0 * T * d * dt_size(m->cfg.kv_dtype));
    p.M = (int)(bc * T);
    p.N = c.n_layers * 2 * c.d_model;
    p.K = c.d_model;
    p.lda = d;
    p.ldw = d;
    p.ldc = d;
    p.bias = m->cross_kv_b.as<float>();
    p.group_n = c.d_model;
    p.group_stride = (long)((size_t)s->B * T * d);
    return gemm_dispatch(m->cfg.compute_dtype, m->cfg.kv_dtype, p, 1, st);
}

// mel2 / split_at (coalesced pairs): utterances [split_at, B) come from mel2 (their own batch's buffer); split_at is a multiple of Bc
// want_f32: also produce the fp32 encoder output (s->enc_f, what wm_encode returns).  The decoder never reads it — the cross-K/V
// projection consumes the operand-dtype rows s->enc_t — so the transcribe paths skip it, and where the last fc2 can carry a
// LayerNorm in its epilogue (16-bit operands, d = 384) `ln_post` rides there: no LayerNorm launch, no 147 MB fp32 write per pass.
// BOTH kinds of caller take the operand rows from the same epilogue, so wm_encode + wm_decode_step and wm_transcribe see
// bit-identical cross-K/V (round 2 had fused it for the transcribe paths only and reverted: the two entry points parted at a near-tie).
static int run_encoder(wm_model* m, wm_state* s, const float* mel_dev, int B, hipStream_t st, const float* mel2 = nullptr, int split_at = 0,
                       bool want_f32 = true) {
    const wm_dims& c = m->cfg.dims;
    const int T = m->cfg.compute_dtype;
    const size_t d = c.d_model, L = 2 * (size_t)c.n_audio_ctx, NT = c.n_audio_ctx, ts = dt_size(T);
    const float scale = 1.0f / sqrtf(64.0f);
    for (int c0 = 0; c0 < B; c0 += s->Bc) {
        const int bc = std::min(s->Bc, B - c0);
        const int M = (int)(bc * NT);
        const int opb = T == WM_F32 ? 4 : 2;
        bool xn_is_ln1 = false;  // xn holds LN1(x) of the coming block, written by the epilogue of the GEMM that produced x
        bool post_fused = false;  // enc_t was written by the last fc2's epilogue
        const float* mel_c = (mel2 && c0 >= split_at) ? mel2 + (size_t)(c0 - split_at) * c.n_mels * L : mel_dev + (size_t)c0 * c.n_mels * L;
        DISPATCH_DT(T, TT, launch_mel_transpose_pad<TT>(mel_c, s->mel_t.p, bc, c.n_mels, (int)L, m->Cp, st));
        {  // conv1 + GELU -> h1 rows 1..L (token-major)   whisper.mojo:73-75
            GemmParams p{};
            p.A = s->mel_t.p;
            p.W = m->conv1_w.p;
            p.C = off_bytes(s->h1, d * ts);
            p.M = (int)L;
            p.N = c.d_model;
            p.K = m->K1;
            p.lda = m->Cp;
            p.ldw = m->K1;
            p.ldc = d;
            p.strideA = (long)((L + 2) * m->Cp);
            p.strideC = (long)((L + 2) * d);
            p.bias = m->conv1_b.as<float>();
            p.act = 1;
            p.gelu_mode = m->cfg.gelu_mode;
            WMCHK(gemm_dispatch(T, T, p, bc, st));
        }
        {  // conv2 (stride 2) + GELU + pos_emb -> x   whisper.mojo:78-89
            GemmParams p{};
            p.A = s->h1.p;
            p.W = m->conv2_w.p;
            p.C = s->x.p;
            p.M = (int)NT;
            p.N = c.d_model;
            p.K = 3 * c.d_model;
            p.lda = 2 * d;
            p.ldw = 3 * d;
            p.ldc = d;
            p.strideA = (long)((L + 2) * d);
            p.strideC = (long)(NT * d);
            p.bias = m->conv2_b.as<float>();
            p.act = 1;
            p.gelu_mode = m->cfg.gelu_mode;
            p.pos = m->enc_pos.as<float>();
            if (gemm_nt_fuses_layernorm_out(opb, p)) {  // the first block's LN1 rides conv2's epilogue
                p.lno_g = m->enc[0].ln1_g.as<float>();
                p.lno_b = m->enc[0].ln1_b.as<float>();
                p.lno_out = s->xn.p;
                xn_is_ln1 = true;
            }
            WMCHK(gemm_dispatch(T, WM_F32, p, bc, st));
        }
        for (int l = 0; l < c.n_layers; ++l) {  // layers.mojo:435-519 with is_decoder=False
            EncLayer& w = m->enc[l];
            GemmParams p{};
            p.A = s->xn.p;
            p.W = w.qkv_w.p;
            p.C = s->qkv.p;
            p.M = M;
            p.N = 3 * c.d_model;
            p.K = c.d_model;
            p.lda = d;
            p.ldw = d;
            p.ldc = 3 * d;
            p.bias = w.qkv_b.as<float>();
            if (xn_is_ln1)  // the producer of x wrote LN1(x) next to it
                WMCHK(gemm_dispatch(T, T, p, 1, st));
            else
                WMCHK(ln_then_gemm(T, T, p, s->x.as<float>(), w.ln1_g.as<float>(), w.ln1_b.as<float>(), st));
            DISPATCH_DT(T, TT, launch_flash_attn_enc<TT>(s->qkv.p, s->ao.p, bc, c.n_heads, c.n_audio_ctx, scale, st));
            GemmParams o{};
            o.A = s->ao.p;
            o.W = w.o_w.p;
            o.C = s->x.p;
            o.M = M;
            o.N = c.d_model;
            o.K = c.d_model;
            o.lda = d;
            o.ldw = d;
            o.ldc = d;
            o.bias = w.o_b.as<float>();
            o.residual = s->x.as<float>();
            o.ldr = d;
            const bool xn_is_ln2 = gemm_nt_fuses_layernorm_out(opb, o);
            if (xn_is_ln2) {
                o.lno_g = w.ln2_g.as<float>();
                o.lno_b = w.ln2_b.as<float>();
                o.lno_out = s->xn.p;
            }
            WMCHK(gemm_dispatch(T, WM_F32, o, 1, st));
            GemmParams f1{};
            f1.A = s->xn.p;
            f1.W = w.fc1_w.p;
            f1.C = s->hid.p;
            f1.M = M;
            f1.N = c.ffn;
            f1.K = c.d_model;
            f1.lda = d;
            f1.ldw = d;
            f1.ldc = c.ffn;
            f1.bias = w.fc1_b.as<float>();
            f1.act = 1;
            f1.gelu_mode = m->cfg.gelu_mode;
            if (xn_is_ln2)
                WMCHK(gemm_dispatch(T, T, f1, 1, st));
            else
                WMCHK(ln_then_gemm(T, T, f1, s->x.as<float>(), w.ln2_g.as<float>(), w.ln2_b.as<float>(), st));
            GemmParams f2{};
            f2.A = s->hid.p;
            f2.W = w.fc2_w.p;
            f2.C = s->x.p;
            f2.M = M;
            f2.N = c.d_model;
            f2.K = c.ffn;
            f2.lda = c.ffn;
            f2.ldw = c.ffn;
            f2.ldc = d;
            f2.bias = w.fc2_b.as<float>();
            f2.residual = s->x.as<float>();
            f2.ldr = d;
            const bool fuse_out = gemm_nt_fuses_layernorm_out(opb, f2);
            xn_is_ln1 = l + 1 < c.n_layers && fuse_out;
            if (xn_is_ln1) {  // the next block's LN1
                f2.lno_g = m->enc[l + 1].ln1_g.as<float>();
                f2.lno_b = m->enc[l + 1].ln1_b.as<float>();
                f2.lno_out = s->xn.p;
            } else if (fuse_out) {  // last block: ln_post (whisper.mojo:97-98) as operand rows for the cross-K/V projection
                f2.lno_g = m->enc_ln_g.as<float>();
                f2.lno_b = m->enc_ln_b.as<float>();
                f2.lno_out = s->enc_t.p;
                post_fused = true;
            }
            WMCHK(gemm_dispatch(T, WM_F32, f2, 1, st));
        }
        float* encf = s->enc_f.as<float>() + (size_t)c0 * NT * d;
        if (!post_fused)  // operand rows (and the fp32 copy) from the LayerNorm kernel
            DISPATCH_DT(T, TT, launch_layernorm_rows<TT>(s->x.as<float>(), m->enc_ln_g.as<float>(), m->enc_ln_b.as<float>(), s->enc_t.p, want_f32 ? encf : nullptr, M, c.d_model, 1e-5f, st));
        else if (want_f32)  // the fp32 copy only
            DISPATCH_DT(T, TT, launch_layernorm_rows<TT>(s->x.as<float>(), m->enc_ln_g.as<float>(), m->enc_ln_b.as<float>(), nullptr, encf, M, c.d_model, 1e-5f, st));
        WMCHK(cross_kv_chunk(m, s, c0, bc, st));
    }
    HIPCHK(hipGetLastError());
    return 0;
}

// B < 0: any batch size (the caller takes it from the state AFTER this check — a stale handle is never dereferenced)
static int check_state(wm_model* m, wm_state* s, int B) {
    if (!m || !s) return fail(WM_E_ARG, "null handle");
    if (!state_is_live(s)) return fail(WM_E_ARG, "stale state handle (its model was freed or reloaded)");
    if (s->m != m) return fail(WM_E_ARG, "state belongs to another model");
    if (B >= 0 && B != s->B) return fail(WM_E_ARG, "B=%d but the state was created for %d", B, s->B);
    return 0;
}

extern "C" int wm_encode(wm_model* m, wm_state* s, const float* mel, int mel_on_device, int B, float* enc_out) {
    WMCHK(check_state(m, s, B));
    if (!mel) return fail(WM_E_ARG, "null mel");
    HIPCHK(hipSetDevice(m->device));
    WMCHK(wm_state_reset(s));
    const wm_dims& c = m->cfg.dims;
    const float* mel_dev = mel;
    if (!mel_on_device) {
        HIPCHK(hipMemcpyAsync(s->mel_dev.p, mel, (size_t)B * c.n_mels * 2 * c.n_audio_ctx * 4, hipMemcpyHostToDevice, m->stream));
        mel_dev = s->mel_dev.as<float>();
    }
    WMCHK(run_encoder(m, s, mel_dev, B, m->stream, nullptr, 0, enc_out != nullptr));
    s->has_enc = s->has_cross = true;
    s->last_mel = mel_dev;
    if (enc_out) HIPCHK(hipMemcpyAsync(enc_out, s->enc_f.p, (size_t)B * c.n_audio_ctx * c.d_model * 4, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    return 0;
}

extern "C" int wm_state_set_encoder_output(wm_model* m, wm_state* s, const float* enc_out, int B) {
    WMCHK(check_state(m, s, B));
    if (!enc_out) return fail(WM_E_ARG, "null enc_out");
    HIPCHK(hipSetDevice(m->device));
    WMCHK(wm_state_reset(s));
    const wm_dims& c = m->cfg.dims;
    const size_t NT = c.n_audio_ctx, d = c.d_model;
    HIPCHK(hipMemcpyAsync(s->enc_f.p, enc_out, (size_t)B * NT * d * 4, hipMemcpyHostToDevice, m->stream));
    for (int c0 = 0; c0 < B; c0 += s->Bc) {
        const int bc = std::min(s->Bc, B - c0);
        DISPATCH_DT(m->cfg.compute_dtype, TT, launch_convert<TT>(s->enc_f.as<float>() + (size_t)c0 * NT * d, s->enc_t.p, (size_t)bc * NT * d, m->stream));
        WMCHK(cross_kv_chunk(m, s, c0, bc, m->stream));
    }
    HIPCHK(hipStreamSynchronize(m->stream));
    s->has_enc = s->has_cross = true;
    return 0;
}

// ---- one decode step for all B utterances: whisper.mojo:130-167 with L_tgt = 1 ------------------------------------
static int dec_linear_dispatch(int dt, const DecLinearParams& p, hipStream_t st) {
    int rc = 0;
    DISPATCH_DT(dt, TT, rc = launch_dec_linear<TT>(p, st));
    return launch_rc(rc);
}
static int attn_decode_dispatch(int dt, const AttnDecParams& p, hipStream_t st) {
    int rc = 0;
    DISPATCH_DT(dt, TT, rc = launch_attn_decode<TT>(p, st));
    return launch_rc(rc);
}

// A view of `nb` utterances starting at b0, decoding on stream st under control block ctl.
struct DecView {
    int b0, nb;
    hipStream_t st;
    StepCtl* ctl;
};
static DecView whole_batch(wm_model* m, wm_state* s) { return DecView{0, s->B, m->stream, s->ctl.as<StepCtl>()}; }

static int launch_cross_attn(wm_model* m, wm_state* s, int l, const DecView& v, int P = 1) {
    const wm_dims& c = m->cfg.dims;
    const size_t d = c.d_model, ks = dt_size(m->cfg.kv_dtype);
    const size_t cross_l = (size_t)s->B * c.n_audio_ctx * d;
    const size_t boff = (size_t)v.b0 * c.n_audio_ctx * d;
    AttnDecParams a{};
    a.q = s->dq.as<float>() + (size_t)v.b0 * d;
    a.K = off_bytes(s->cross_kv, ((size_t)(2 * l) * cross_l + boff) * ks);
    a.V = off_bytes(s->cross_kv, ((size_t)(2 * l + 1) * cross_l + boff) * ks);
    a.batch_stride = (long)((size_t)c.n_audio_ctx * d);
    a.n_keys = c.n_audio_ctx;
    a.ctl = v.ctl;
    a.nsplit = s->nsplit;
    a.scale = 1.0f / sqrtf(64.0f);
    a.part_o = s->part_o.as<float>() + (size_t)v.b0 * s->nsplit * d;
    a.part_ml = s->part_ml.as<float>() + (size_t)v.b0 * s->nsplit * c.n_heads * 2;
    a.H = c.n_heads;
    a.d = c.d_model;
    a.B = v.nb * P;
    a.q_B = P > 1 ? v.nb : 0;
    static const bool no_mq = wm_env("WM_NO_MQ_PREFILL") != nullptr;
    a.nq = (P == 4 && !no_mq) ? 4 : 0;  // the reference's 4-token prompt: one K/V sweep for the four positions
    a.ts = (long long*)m->ts_buf.p;
    a.ts_id = s->trace_id;
    // fewer K/V-streaming workgroups per CU when passes share the chip (AttnDecParams.lds_pad).  16-bit K/V: 34 KB of pad (+ 20 KB
    // static) = two per CU.  fp32 K/V (12 KB static): ONE per CU measured best with four 128-row passes in flight — pipelined ms per
    // 64-clip pass: pad 0 (four per CU) 28.2, 20 KB 28.4, 34 KB (three) 27.8, 42 / 50 / 60 KB (two) 27.3 / 27.5 / 27.4, 90 KB (one) 27.05
    a.lds_pad = s->shares_chip ? (m->cfg.kv_dtype == WM_F32 ? 90000 : 34 * 1024) : 0;
    return attn_decode_dispatch(m->cfg.kv_dtype, a, v.st);
}

// want_logits: run the final LN + vocabulary projection.  full_logits: also materialise [B, vocab] fp32 (stage tests,
// wm_decode_step); the greedy loop only needs the fused-argmax partials.
// P > 1: prompt prefill — P positions of every utterance in ONE pass (whisper.mojo:195: the q_len = n_prompt block).  Rows are
// position-major (row = t * B + b), tokens / positions come from tok_rows / pos_rows, K/V rows go to cache rows len + t, the
// self-attention of position t sees keys 0..len+t (the causal mask of layers.mojo:309-318), logits only for the last
// position.  Every row's arithmetic is what the single-position pass does for it, so the ids are the same bit for bit.
static int decode_core(wm_model* m, wm_state* s, const DecView& v, bool want_logits, bool full_logits = false,
                       const float* mask = nullptr, int P = 1, bool embed = true, const TsRules* rules = nullptr) {
    const wm_dims& c = m->cfg.dims;
    const int T = dec_dtype(m->cfg), KV = m->cfg.kv_dtype;  // T: the decoder's operand dtype
    const int B = v.nb * P;          // activation rows of this pass
    const int qB = P > 1 ? v.nb : 0;  // position-major row mapping on
    const size_t d = c.d_model, ks = dt_size(KV);
    hipStream_t st = v.st;
    const StepCtl* ctl = v.ctl;
    const float scale = 1.0f / sqrtf(64.0f);
    const size_t self_l = (size_t)s->B * c.n_text_ctx * d;  // elements per (layer, K|V)
    const size_t self_b = (size_t)v.b0 * c.n_text_ctx * d;
    float* dx = s->dx.as<float>() + (size_t)v.b0 * d;
    float* dq = s->dq.as<float>() + (size_t)v.b0 * d;
    // attention output / MLP hidden rows: operand dtype T (fp32 buffers, used at T's width)
    float* dattn = (float*)off_bytes(s->dattn, (size_t)v.b0 * d * dt_size(T));
    float* dhid = (float*)off_bytes(s->dhid, (size_t)v.b0 * c.ffn * dt_size(T));
    if (embed)  // (the greedy loop's steps get their input row from the previous step's argmax launch instead)
        launch_dec_embed(m->tok_emb_f.as<float>(), m->dec_pos.as<float>(), P > 1 ? s->tok_rows.as<int>() : s->tok.as<int>() + v.b0,
                     P > 1 ? s->pos_rows.as<int>() : s->pos.as<int>() + v.b0, dx, B, c.d_model, st);
    for (int l = 0; l < c.n_layers; ++l) {
        DecLayer& w = m->dec[l];
        void* sk = off_bytes(s->self_kv, ((size_t)(2 * l) * self_l + self_b) * ks);
        void* sv = off_bytes(s->self_kv, ((size_t)(2 * l + 1) * self_l + self_b) * ks);
        {  // LN1 -> q | k,v appended to the cache at row current_len   (layers.mojo:118-147)
            DecLinearParams p{};
            p.x = dx;
            p.ldx = c.d_model;
            p.ln_g = w.ln1_g.as<float>();
            p.ln_b = w.ln1_b.as<float>();
            p.W = w.sqkv_w.p;
            p.N = 3 * c.d_model;
            p.K = c.d_model;
            p.B = B;
            p.bias = w.sqkv_b.as<float>();
            p.out = dq;
            p.ldo = c.d_model;
            p.kcache = sk;
            p.vcache = sv;
            p.kv_batch_stride = (long)((size_t)c.n_text_ctx * d);
            p.d_model = c.d_model;
            p.kv_dtype = KV;
            p.kv_B = qB;
            p.ctl = ctl;
            WMCHK(dec_linear_dispatch(T, p, st));
        }
        {  // self-attention over current_len+1 cached rows   (layers.mojo:186-272)
            AttnDecParams a{};
            a.q = dq;
            a.K = sk;
            a.V = sv;
            a.batch_stride = (long)((size_t)c.n_text_ctx * d);
            a.n_keys = -1;
            a.q_B = qB;
            a.ctl = ctl;
            a.nsplit = 1;
            a.scale = scale;
            a.direct_out = dattn;
            a.out_dtype = T;
            a.H = c.n_heads;
            a.d = c.d_model;
            a.B = B;
            WMCHK(attn_decode_dispatch(KV, a, st));
        }
        // the attention outputs and the MLP hidden rows are handed over in operand dtype T (what the next MFMA consumes)
        auto proj_residual = [&](const float* in, int K, const DevBuf& W, const DevBuf& bias) -> int {  // x += in·Wᵀ + b
            DecLinearParams p{};
            p.x = in;
            p.ldx = K;
            p.x_is_t = 1;
            p.W = W.p;
            p.N = c.d_model;
            p.K = K;
            p.B = B;
            p.bias = bias.as<float>();
            p.residual = dx;
            p.ldr = c.d_model;
            p.out = dx;
            p.ldo = c.d_model;
            return dec_linear_dispatch(T, p, st);
        };
        WMCHK(proj_residual(dattn, c.d_model, w.so_w, w.so_b));
        {  // LNx -> cross q
            DecLinearParams p{};
            p.x = dx;
            p.ldx = c.d_model;
            p.ln_g = w.lnx_g.as<float>();
            p.ln_b = w.lnx_b.as<float>();
            p.W = w.cq_w.p;
            p.N = c.d_model;
            p.K = c.d_model;
            p.B = B;
            p.bias = w.cq_b.as<float>();
            p.out = dq;
            p.ldo = c.d_model;
            WMCHK(dec_linear_dispatch(T, p, st));
        }
        WMCHK(launch_cross_attn(m, s, l, v, P));
        // (merging the chunk partials inside the projection's prologue was measured 14 us per layer SLOWER than this
        // 3 us launch: 96 workgroups each re-reading 295 KB of partials)
        launch_attn_combine(s->part_o.as<float>() + (size_t)v.b0 * s->nsplit * d, s->part_ml.as<float>() + (size_t)v.b0 * s->nsplit * c.n_heads * 2,
                            dattn, T, B, s->nsplit, c.n_heads, c.d_model, st, (long long*)m->ts_buf.p, s->trace_id);
        WMCHK(proj_residual(dattn, c.d_model, w.co_w, w.co_b));
        {  // LN2 -> fc1 + GELU
            DecLinearParams p{};
            p.x = dx;
            p.ldx = c.d_model;
            p.ln_g = w.ln2_g.as<float>();
            p.ln_b = w.ln2_b.as<float>();
            p.W = w.fc1_w.p;
            p.N = c.ffn;
            p.K = c.d_model;
            p.B = B;
            p.bias = w.fc1_b.as<float>();
            p.act = 1;
            p.gelu_mode = m->cfg.gelu_mode;
            p.out = dhid;
            p.out_is_t = 1;
            p.ldo = c.ffn;
            WMCHK(dec_linear_dispatch(T, p, st));
        }
        WMCHK(proj_residual(dhid, c.ffn, w.fc2_w, w.fc2_b));
    }
    if (want_logits) {  // final LN + tied-embedding logits (whisper.mojo:156-166), no bias
        DecLinearParams p{};
        p.x = dx + (size_t)(P - 1) * v.nb * d;  // rows of the last position
        p.ldx = c.d_model;
        p.ln_g = m->dec_ln_g.as<float>();
        p.ln_b = m->dec_ln_b.as<float>();
        p.W = T == WM_F32 ? m->tok_emb_f.p : m->tok_emb_t.p;
        p.N = c.vocab;
        p.K = c.d_model;
        p.B = v.nb;
        p.out = full_logits ? s->logits.as<float>() + (size_t)v.b0 * m->Vpad : nullptr;
        p.ldo = m->Vpad;
        p.amax_val = s->amax_val.as<float>() + (size_t)v.b0 * s->npart;
        p.amax_idx = s->amax_idx.as<int>() + (size_t)v.b0 * s->npart;
        p.amax_stride = s->npart;
        p.amax_mask = mask;
        if (rules && rules->tb > 0) {  // timestamp rules: split the candidates by the utterances' admissible ranges
            p.ts_state = s->ts_state.as<TsState>() + v.b0;
            p.ts_begin = rules->tb;
            p.ts_val = s->ts_val.as<float>() + (size_t)v.b0 * s->npart;
            p.ts_idx = s->ts_idx.as<int>() + (size_t)v.b0 * s->npart;
            p.ts_m = s->ts_m.as<float>() + (size_t)v.b0 * s->npart;
            p.ts_s = s->ts_s.as<float>() + (size_t)v.b0 * s->npart;
        }
        p.ts = (long long*)m->ts_buf.p;
        p.ts_id = s->trace_id;
        int lrc = 0;
        DISPATCH_DT(T, TT, lrc = launch_dec_logits<TT>(p, st));
        LCHK(lrc);
    }
    return 0;
}

static ArgmaxParams argmax_params(wm_model* m, wm_state* s, const DecView& v, bool record, int eot, int ignore_eot,
                                  bool advance = false, bool embed_next = false, const TsRules* rules = nullptr) {
    ArgmaxParams a{};
    if (rules && rules->tb > 0) {
        a.ts_state = s->ts_state.as<TsState>() + v.b0;
        a.rules = *rules;
        a.ts_val = s->ts_val.as<float>() + (size_t)v.b0 * s->npart;
        a.ts_idx = s->ts_idx.as<int>() + (size_t)v.b0 * s->npart;
        a.ts_m = s->ts_m.as<float>() + (size_t)v.b0 * s->npart;
        a.ts_s = s->ts_s.as<float>() + (size_t)v.b0 * s->npart;
        a.ts_part0 = rules->tb / dec_logits_ids_per_part(m->cfg.dims.vocab);
    }
    a.logits = s->logits.as<float>() + (size_t)v.b0 * m->Vpad;
    a.pval = s->amax_val.as<float>() + (size_t)v.b0 * s->npart;
    a.pidx = s->amax_idx.as<int>() + (size_t)v.b0 * s->npart;
    a.npart = s->npart;
    a.ldl = m->Vpad;
    a.V = m->cfg.dims.vocab;
    a.B = v.nb;
    a.next = s->tok.as<int>() + v.b0;
    a.out_tokens = record ? s->out_tokens.as<int>() + (size_t)v.b0 * s->out_stride : nullptr;
    a.out_stride = s->out_stride;
    a.n_tokens = s->n_tokens.as<int>() + v.b0;
    a.finished = s->finished.as<int>() + v.b0;
    a.ctl = v.ctl;
    a.eot = eot;
    a.ignore_eot = ignore_eot;
    a.advance = advance ? 1 : 0;
    a.pos = s->pos.as<int>() + v.b0;
    a.ts = (long long*)m->ts_buf.p;
    a.ts_id = s->trace_id;
    if (advance) a.host_progress = s->d_prog;  // the greedy loop's steps report (finished, cache length) to the host
    if (embed_next) {
        a.emb_tok = m->tok_emb_f.as<float>();
        a.emb_pos = m->dec_pos.as<float>();
        a.emb_out = s->dx.as<float>() + (size_t)v.b0 * m->cfg.dims.d_model;
        a.d = m->cfg.dims.d_model;
        a.max_pos = m->cfg.dims.n_text_ctx - 1;
    }
    return a;
}

extern "C" int wm_decode_step(wm_model* m, wm_state* s, const int32_t* tokens, int q_len, const int32_t* start_pos,
                              float* logits, int32_t* next) {
    if (!m || !s || !tokens || !start_pos || q_len <= 0) return fail(WM_E_ARG, "bad argument");
    WMCHK(check_state(m, s, -1));
    if (!s->has_enc) return fail(WM_E_STATE, "no encoder output in this state (call wm_encode first)");
    const wm_dims& c = m->cfg.dims;
    const int B = s->B;
    if (s->host_len + q_len > c.n_text_ctx) return fail(WM_E_STATE, "KV cache full (%d + %d > %d)", s->host_len, q_len, c.n_text_ctx);
    for (int b = 0; b < B; ++b) {
        if (start_pos[b] < 0 || start_pos[b] + q_len > c.n_text_ctx) return fail(WM_E_ARG, "start_pos[%d]=%d out of range", b, start_pos[b]);
        for (int i = 0; i < q_len; ++i)
            if (tokens[b * q_len + i] < 0 || tokens[b * q_len + i] >= c.vocab) return fail(WM_E_ARG, "token id out of range");
    }
    HIPCHK(hipSetDevice(m->device));
    hipStream_t st = m->stream;
    const DecView v = whole_batch(m, s);
    std::vector<int32_t> col(B), pos(B);
    if (q_len > 1 && q_len <= wm_state::PREFILL_MAX) {  // the q_len block of whisper.mojo:195 as ONE position-major pass
        std::vector<int32_t> trow((size_t)q_len * B), prow((size_t)q_len * B);
        for (int i = 0; i < q_len; ++i)
            for (int b = 0; b < B; ++b) {
                trow[(size_t)i * B + b] = tokens[b * q_len + i];
                prow[(size_t)i * B + b] = start_pos[b] + i;
            }
        HIPCHK(hipMemcpyAsync(s->tok_rows.p, trow.data(), trow.size() * 4, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(s->pos_rows.p, prow.data(), prow.size() * 4, hipMemcpyHostToDevice, st));
        launch_set_step(v.ctl, s->host_len, 1, nullptr, 0, nullptr, 0, B, st);
        WMCHK(decode_core(m, s, v, true, true, nullptr, q_len));
        HIPCHK(hipGetLastError());         // a launch that failed (bad configuration, LDS attribute) is reported here
        HIPCHK(hipStreamSynchronize(st));  // trow / prow go out of scope
        s->host_len += q_len;
        q_len = 0;
    }
    for (int i = 0; i < q_len; ++i) {
        for (int b = 0; b < B; ++b) {
            col[b] = tokens[b * q_len + i];
            pos[b] = start_pos[b] + i;
        }
        HIPCHK(hipMemcpyAsync(s->tok.p, col.data(), B * 4, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(s->pos.p, pos.data(), B * 4, hipMemcpyHostToDevice, st));
        launch_set_step(v.ctl, s->host_len, 1, nullptr, 0, nullptr, 0, B, st);
        WMCHK(decode_core(m, s, v, i == q_len - 1, true));
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(st));  // col/pos are reused next iteration
        s->host_len += 1;
    }
    launch_set_step(v.ctl, s->host_len, 1, nullptr, 0, nullptr, 0, B, st);
    if (next) {
        launch_argmax_step(argmax_params(m, s, v, false, -1, 1), st);
        HIPCHK(hipMemcpyAsync(next, s->tok.p, B * 4, hipMemcpyDeviceToHost, st));
    }
    if (logits)
        HIPCHK(hipMemcpy2DAsync(logits, (size_t)c.vocab * 4, s->logits.p, (size_t)m->Vpad * 4, (size_t)c.vocab * 4, B, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    return 0;
}

// ---- greedy-loop enqueue machinery (used by transcribe_decode and the loop pump) -----------------------------------------------
static const int LOOP_CHUNK = 8;  // graph replays per sub-chunk; two sub-chunks (16 steps) are queued ahead of the GPU

// n more steps of the pending pass's loop on the state's lane streams (captured graph, or eager launches in the developer build)
static int enqueue_loop_steps(wm_model* m, wm_state* s, int n) {
    const TsRules* rp = s->loop_opts.rules.tb > 0 ? &s->loop_opts.rules : nullptr;
    for (int k = 0; k < n; ++k, ++s->loop_enq) {
        const int it = s->loop_enq;
        for (auto& ln : s->lanes) {
            if (trace_events_on() && it % 10 == 9) trace_mark(ln.st, "state %p lane %d step %d", (void*)s, ln.b0, it);
            if (ln.graph[0] && s->graphs_valid) {
                const hipError_t e = hipGraphLaunch(ln.graph[it % wm_state::Lane::NEXEC], ln.st);
                if (e != hipSuccess) return fail(WM_E_HIP, "hipGraphLaunch: %s", hipGetErrorString(e));
            } else {
                const DecView v{ln.b0, ln.nb, ln.st, ln.ctl};
                WMCHK(decode_core(m, s, v, true, false, s->mask_steady.as<float>(), 1, false, rp));
                launch_argmax_step(argmax_params(m, s, v, true, s->loop_opts.eot, s->loop_opts.ignore_eot, true, true, rp), v.st);
            }
        }
    }
    return 0;
}
// the pending pass is fully enqueued: its completion events go behind the last step
static int finish_enqueue(wm_model* m, wm_state* s) {
    (void)m;
    int rc = 0;
    for (auto& ln : s->lanes) {
        trace_mark(ln.st, "state %p lane %d decode end", (void*)s, ln.b0);
        const hipError_t e = hipEventRecord(ln.done, ln.st);
        if (e != hipSuccess && !rc) rc = fail(WM_E_HIP, "hipEventRecord: %s", hipGetErrorString(e));
    }
    if (!rc && hipGetLastError() != hipSuccess) rc = fail(WM_E_HIP, "a launch of the greedy loop failed");
    s->last_steps = s->loop_enq;
    if (rc && !s->enq_rc) {
        s->enq_rc = rc;
        s->enq_err = g_err;
    }
    s->enq_done.store(true);
    return rc;
}
static int enqueue_chunk(wm_model* m, wm_state* s) {
    const int n = std::min(LOOP_CHUNK, s->loop_total - s->loop_enq);
    WMCHK(enqueue_loop_steps(m, s, n));
    HIPCHK(hipEventRecord(s->chunk_ev[s->chunk_k & 1], s->lanes[0].st));
    ++s->chunk_k;
    return 0;
}
// One decision of the loop pump for state s: once sub-chunk k-2 has completed, either stop (every utterance finished, or the loop
// bound reached) or enqueue sub-chunk k.  Returns 1 when it did something, 0 when sub-chunk k-2 is still running (non-blocking
// form), < 0 on error (the pass is then closed with enq_rc set).
static int pump_step(wm_model* m, wm_state* s, bool block) {
    hipEvent_t ev = s->chunk_ev[s->chunk_k & 1];  // recorded behind sub-chunk chunk_k - 2
    hipError_t e = block ? hipEventSynchronize(ev) : hipEventQuery(ev);
    if (e == hipErrorNotReady) return 0;
    int rc = e == hipSuccess ? 0 : fail(WM_E_HIP, "loop pump: %s", hipGetErrorString(e));
    if (!rc) {
        const bool all_done = s->h_prog[0] >= s->B;  // a lower bound of the finished count: never stops a running utterance
        if (all_done || s->loop_enq >= s->loop_total) {
            rc = finish_enqueue(m, s);
            return rc ? -1 : 1;
        }
        rc = enqueue_chunk(m, s);
        if (!rc) return 1;
    }
    s->enq_rc = rc;
    s->enq_err = g_err;
    (void)finish_enqueue(m, s);
    return -1;
}
static void pump_main(wm_model* m) {
    (void)hipSetDevice(m->device);
    std::unique_lock<std::mutex> lk(m->pump_mu);
    for (;;) {
        m->pump_cv.wait(lk, [&] { return m->pump_quit || !m->pump_work.empty(); });
        if (m->pump_quit) return;
        bool progressed = false;
        for (size_t i = 0; i < m->pump_work.size();) {
            wm_state* s = m->pump_work[i];
            if (pump_step(m, s, false) != 0) progressed = true;
            if (s->enq_done.load()) {
                m->pump_work.erase(m->pump_work.begin() + (long)i);
                m->pump_cv.notify_all();  // wm_transcribe_wait may be waiting for this pass to be fully enqueued
            } else {
                ++i;
            }
        }
        if (!progressed) {  // every pending sub-chunk still running: they take milliseconds
            lk.unlock();
            std::this_thread::sleep_for(std::chrono::microseconds(100));
            lk.lock();
        }
    }
}

// ---- Whisper.transcribe: whisper.mojo:184-223 ------------------------------------------------------------------------
// Enqueues the prompt prefill and the greedy loop for state s on its decode lane streams; returns without waiting.  The lanes
// first wait for the encoder (recorded on the stream it ran on).  Both entry points stop once every utterance has emitted eot
// (whisper.mojo:206-207): allow_poll = the synchronous one feeds its loop itself, sub-chunk by sub-chunk; the pipelined one
// hands the rest of the loop to the model's pump thread (see the enqueue machinery above).
static int transcribe_decode(wm_model* m, wm_state* s, const wm_decode_opts* o, bool allow_poll) {
    HIPCHK(hipEventRecord(s->enc_done, s->enc_stream ? s->enc_stream : m->stream));  // encoder + cross K/V of this state
    static const bool trace_phase = wm_env("WM_TRACE_HOST") != nullptr;
    const auto tp0 = std::chrono::steady_clock::now();
    static const bool no_graph = wm_env("WM_NO_GRAPH") != nullptr;
    TsRules rules{};  // timestamp rules of this pass (tb <= 0: off)
    rules.tb = o->timestamp_begin > 0 ? o->timestamp_begin : 0;
    rules.eos = o->eot;
    rules.max_init = o->max_initial_timestamp_index;
    rules.vocab = m->cfg.dims.vocab;
    const TsRules* rp = rules.tb > 0 ? &rules : nullptr;
    const int no_ts = rp ? o->no_timestamps_token : -1;
    s->shares_chip = !allow_poll;  // the pipelined entry (wm_transcribe_submit): other passes are, or will be, in flight
    const bool recapture = !s->graphs_valid || s->graph_eot != o->eot || s->graph_ignore != o->ignore_eot ||
                           s->graph_shares != s->shares_chip || memcmp(&s->graph_rules, &rules, sizeof rules) != 0;
    const int first_pos = o->pos_mode == WM_POS_REF ? o->n_prompt - 1 : o->n_prompt;
    // logit masks (§8f rank 4): rebuilt only when the id lists change; always passed (all-zero = the reference's raw argmax)
    {
        std::vector<int32_t> sup(o->suppress_tokens, o->suppress_tokens + (o->suppress_tokens ? o->n_suppress : 0));
        std::vector<int32_t> bsup(o->begin_suppress_tokens, o->begin_suppress_tokens + (o->begin_suppress_tokens ? o->n_begin_suppress : 0));
        if (!s->masks_valid || sup != s->sup_cached || bsup != s->bsup_cached || no_ts != s->no_ts_cached) {
            std::vector<float> ms(m->Vpad, 0.f), mb(m->Vpad, 0.f);
            if (no_ts >= 0 && no_ts < m->cfg.dims.vocab) ms[no_ts] = mb[no_ts] = -INFINITY;  // <|notimestamps|> is never emitted under the timestamp rules
            s->no_ts_cached = no_ts;
            for (int32_t id : sup)
                if (id >= 0 && id < m->cfg.dims.vocab) ms[id] = mb[id] = -INFINITY;
            for (int32_t id : bsup)
                if (id >= 0 && id < m->cfg.dims.vocab) mb[id] = -INFINITY;
            HIPCHK(hipMemcpy(s->mask_steady.p, ms.data(), ms.size() * 4, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(s->mask_begin.p, mb.data(), mb.size() * 4, hipMemcpyHostToDevice));
            s->sup_cached = sup;
            s->bsup_cached = bsup;
            s->masks_valid = true;
        }
    }
    InitTokensParams ip{};
    ip.n_prompt = o->n_prompt;
    for (int i = 0; i < o->n_prompt; ++i) ip.prompt[i] = o->prompt[i];
    for (auto& ln : s->lanes) {
        const DecView v{ln.b0, ln.nb, ln.st, ln.ctl};
        HIPCHK(hipStreamWaitEvent(v.st, s->enc_done, 0));
        trace_mark(v.st, "state %p lane %d decode start", (void*)s, v.b0);
        // tokens = prompt (whisper.mojo:187-191, 200-202); finished = 0; control block = 0
        ip.out_tokens = s->out_tokens.as<int>() + (size_t)v.b0 * s->out_stride;
        ip.out_stride = s->out_stride;
        ip.n_tokens = s->n_tokens.as<int>() + v.b0;
        ip.finished = s->finished.as<int>() + v.b0;
        ip.ctl = v.ctl;
        ip.B = v.nb;
        ip.tok_rows = s->lanes.size() == 1 ? s->tok_rows.as<int>() : nullptr;
        ip.pos_rows = s->pos_rows.as<int>();
        ip.ts_state = rp ? s->ts_state.as<TsState>() + v.b0 : nullptr;
        ip.rules = rules;
        launch_init_tokens(ip, v.st);
        // prefill (whisper.mojo:195, start_pos=0): the q_len = n_prompt causal block equals n_prompt single-token steps
        static const bool seq_prefill = wm_env("WM_SEQ_PREFILL") != nullptr;  // A/B: one pass per prompt position
        if (!seq_prefill && s->lanes.size() == 1 && o->n_prompt > 1 && o->n_prompt <= wm_state::PREFILL_MAX) {
            WMCHK(decode_core(m, s, v, true, false, s->mask_begin.as<float>(), o->n_prompt, true, rp));  // init_tokens filled tok_rows / pos_rows
        } else {
            for (int i = 0; i < o->n_prompt; ++i) {
                launch_set_step(v.ctl, i, 1, s->pos.as<int>() + v.b0, i, s->tok.as<int>() + v.b0, o->prompt[i], v.nb, v.st);
                WMCHK(decode_core(m, s, v, i == o->n_prompt - 1, false, s->mask_begin.as<float>(), 1, true, rp));
            }
        }
        launch_argmax_step(argmax_params(m, s, v, true, o->eot, o->ignore_eot, false, false, rp), v.st);  // :198-203
        trace_mark(v.st, "state %p lane %d prefill end", (void*)s, v.b0);
        // incremental steps: start_pos = current_len - 1 (reference, :217) or current_len (HF)
        launch_set_step(v.ctl, o->n_prompt, 1, s->pos.as<int>() + v.b0, first_pos, nullptr, 0, v.nb, v.st);
        // input row of the first loop step; every later step's row is written by the preceding step's argmax launch
        launch_dec_embed(m->tok_emb_f.as<float>(), m->dec_pos.as<float>(), s->tok.as<int>() + v.b0, s->pos.as<int>() + v.b0,
                         s->dx.as<float>() + (size_t)v.b0 * m->cfg.dims.d_model, v.nb, m->cfg.dims.d_model, v.st);
        // steady state: one captured graph per lane = [37 decode-step launches + argmax/bookkeeping]; every per-step
        // quantity (token, position, cache length) lives in HBM, so the same graph is replayed for every token
        if (!no_graph && (recapture || !ln.graph[0])) {
            for (auto& ge : ln.graph) {
                if (ge) (void)hipGraphExecDestroy(ge);
                ge = nullptr;
            }
            hipGraph_t g = nullptr;
            HIPCHK(hipStreamBeginCapture(v.st, hipStreamCaptureModeThreadLocal));
            const int crc = decode_core(m, s, v, true, false, s->mask_steady.as<float>(), 1, false, rp);
            launch_argmax_step(argmax_params(m, s, v, true, o->eot, o->ignore_eot, true, true, rp), v.st);
            const hipError_t cap = hipStreamEndCapture(v.st, &g);  // always closed, also when a launcher refused
            if (crc) {
                if (g) (void)hipGraphDestroy(g);
                return crc;
            }
            HIPCHK(cap);
            hipError_t ge = hipSuccess;
            for (int k = 0; k < wm_state::Lane::NEXEC && ge == hipSuccess; ++k) ge = hipGraphInstantiate(&ln.graph[k], g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            if (ge != hipSuccess) return fail(WM_E_HIP, "hipGraphInstantiate: %s", hipGetErrorString(ge));
        }
    }
    s->graphs_valid = !no_graph;
    s->graph_eot = o->eot;
    s->graph_ignore = o->ignore_eot;
    s->graph_rules = rules;
    s->graph_shares = s->shares_chip;
    if (trace_phase) {
        for (auto& ln : s->lanes) (void)hipStreamSynchronize(ln.st);
        fprintf(stderr, "[wm] encoder wait + prefill (+graph capture if any): %.3f ms\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - tp0).count() * 1e3);
    }
    // The loop itself.  Fixed-length passes (ignore_eot: the bench's "fixed" mode) and short loops are enqueued whole.  With the
    // reference's stop rule the loop goes out in sub-chunks of LOOP_CHUNK graph replays, two sub-chunks ahead of the GPU; before
    // each further sub-chunk the host reads the (finished, length) pair the argmax launches write to pinned memory — no stream
    // synchronisation, the GPU never runs dry — and stops enqueueing once every utterance has emitted eot: "if next_token == eot:
    // break" (whisper.mojo:206-207) for the whole batch, at most two sub-chunks late.
    s->loop_total = o->max_loop;
    s->loop_enq = 0;
    s->chunk_k = 0;
    s->loop_opts.eot = o->eot;
    s->loop_opts.ignore_eot = o->ignore_eot;
    s->loop_opts.rules = rules;
    s->enq_rc = 0;
    s->enq_err.clear();
    s->h_prog[0] = 0;
    s->h_prog[1] = 0;
    const bool natural = !o->ignore_eot && s->lanes.size() == 1 && o->max_loop > 2 * LOOP_CHUNK && !no_graph;
    s->enq_done.store(false);
    if (!natural) {
        WMCHK(enqueue_loop_steps(m, s, o->max_loop));
        return finish_enqueue(m, s);
    }
    WMCHK(enqueue_chunk(m, s));
    WMCHK(enqueue_chunk(m, s));
    if (allow_poll) {  // the synchronous entry pumps its own loop
        while (!s->enq_done.load()) WMCHK(pump_step(m, s, true) < 0 ? s->enq_rc : 0);
        return 0;
    }
    {  // pipelined entry: the model's pump thread carries on; wm_transcribe_wait waits for it
        std::lock_guard<std::mutex> lk(m->pump_mu);
        if (!m->pump.joinable()) m->pump = std::thread(pump_main, m);
        m->pump_work.push_back(s);
    }
    m->pump_cv.notify_all();
    return 0;
}

static int check_opts(wm_model* m, const wm_decode_opts* o, int B) {
    if (!o || B <= 0) return fail(WM_E_ARG, "bad argument");
    if (!o->prompt || o->n_prompt <= 0 || o->n_prompt > 16 || o->max_loop < 0) return fail(WM_E_ARG, "bad decode options (1 <= n_prompt <= 16)");
    if (o->n_suppress < 0 || o->n_begin_suppress < 0 || (o->n_suppress > 0 && !o->suppress_tokens) || (o->n_begin_suppress > 0 && !o->begin_suppress_tokens))
        return fail(WM_E_ARG, "bad suppress-token lists");
    const wm_dims& c = m->cfg.dims;
    if (o->timestamp_begin > 0) {
        if (o->timestamp_begin >= c.vocab) return fail(WM_E_ARG, "timestamp_begin %d is not a vocabulary id", o->timestamp_begin);
        if (o->no_timestamps_token >= c.vocab) return fail(WM_E_ARG, "no_timestamps_token out of range");
        if (o->eot < 0 || o->eot > o->timestamp_begin) return fail(WM_E_ARG, "timestamp rules need 0 <= eot <= timestamp_begin (eot is the processor's eos id)");
    }
    const int total = o->n_prompt + 1 + o->max_loop;
    if (total > c.n_text_ctx + 1 || total > OUT_STRIDE_MAX)
        return fail(WM_E_ARG, "n_prompt + 1 + max_loop = %d exceeds the decoder context %d", total, c.n_text_ctx);
    for (int i = 0; i < o->n_prompt; ++i)
        if (o->prompt[i] < 0 || o->prompt[i] >= c.vocab) return fail(WM_E_ARG, "prompt id out of range");
    return 0;
}

// Enqueues one whole pass (encoder, prefill, greedy loop) for B utterances on state *slot (created / re-created on demand).
// mel2 != null: a coalesced pair — *slot is a 2·(B/2)-row pair state, utterances [B/2, B) come from mel2.
static int submit_on(wm_model* m, wm_state** slot, const float* mel, int mel_on_device, int B, const wm_decode_opts* o, bool allow_poll,
                     const float* mel2 = nullptr, int mel2_on_device = 0) {
    const wm_dims& c = m->cfg.dims;
    HIPCHK(hipSetDevice(m->device));
    const bool pair = mel2 != nullptr;
    // a pass that was submitted and not yet waited for owns the slot's state: refuse BEFORE touching it (re-creating the
    // state for another batch size would destroy graphs, streams and arenas under its running kernels)
    if (*slot && (*slot)->pending) return fail(WM_E_STATE, "this slot still holds a pass that was not waited for");
    if (!*slot || (*slot)->B != B) {
        if (*slot) wm_state_free(*slot);
        *slot = nullptr;
        WMCHK(state_new(m, B, slot, pair));
    }
    wm_state* s = *slot;
    s->trace_id = slot == &m->cached ? 1 : (slot >= m->slots && slot < m->slots + (wm_model::NSLOT - 1)) ? 2 + (int)(slot - m->slots) : 10 + (int)(slot - m->pairs);
    // The whole pass — encoder, prefill, greedy loop — goes on the slot's own stream: four slots are then four hardware
    // queues, which is what the chip runs concurrently (a fifth queue, e.g. a shared encoder stream, lands on a pipe that
    // already serves one of them and the two take turns: 22.3 vs 20.8 ms per pass at four passes in flight).
    // WM_ENC_ON_MODEL_STREAM=1 restores the shared encoder stream for A/B runs.
    static const bool enc_on_lane = wm_env("WM_ENC_ON_MODEL_STREAM") == nullptr;
    hipStream_t est = enc_on_lane ? s->lanes[0].st : m->stream;
    s->has_enc = s->has_cross = false;
    s->host_len = 0;
    if (est != m->stream) {  // whatever produced the mel on the model stream (the log-mel front end) comes first
        HIPCHK(hipEventRecord(s->enc_done, m->stream));
        HIPCHK(hipStreamWaitEvent(est, s->enc_done, 0));
    }
    const size_t mel_floats = (size_t)c.n_mels * 2 * c.n_audio_ctx;  // per utterance
    const int B1 = pair ? B / 2 : B;
    const float* mel_dev = mel;
    if (!mel_on_device) {
        HIPCHK(hipMemcpyAsync(s->mel_dev.p, mel, (size_t)B1 * mel_floats * 4, hipMemcpyHostToDevice, est));
        mel_dev = s->mel_dev.as<float>();
    }
    const float* mel2_dev = mel2;
    if (pair && !mel2_on_device) {
        float* dst = s->mel_dev.as<float>() + (size_t)B1 * mel_floats;
        HIPCHK(hipMemcpyAsync(dst, mel2, (size_t)B1 * mel_floats * 4, hipMemcpyHostToDevice, est));
        mel2_dev = dst;
    }
    const auto tt0 = std::chrono::steady_clock::now();
    trace_mark(est, "state %p encoder start", (void*)s);
    WMCHK(run_encoder(m, s, mel_dev, B, est, mel2_dev, B1, false));
    trace_mark(est, "state %p encoder end", (void*)s);
    s->enc_stream = est;
    s->has_enc = s->has_cross = true;
    s->last_mel = pair ? nullptr : mel_dev;
    if (wm_env("WM_TRACE_HOST")) {
        const auto tt1 = std::chrono::steady_clock::now();
        (void)hipStreamSynchronize(m->stream);
        fprintf(stderr, "[wm] encoder: enqueue %.3f ms, done after %.3f ms\n", std::chrono::duration<double>(tt1 - tt0).count() * 1e3,
                std::chrono::duration<double>(std::chrono::steady_clock::now() - tt0).count() * 1e3);
    }
    WMCHK(transcribe_decode(m, s, o, allow_poll));
    s->pending = true;
    s->synced = false;
    s->halves_left = pair ? 2 : 1;
    s->pend_total = o->n_prompt + 1 + o->max_loop;
    s->host_len = 0;
    return 0;
}

// Blocks until the state's pending pass is complete, then copies `rows` utterances starting at row0 out.  The pass stays pending
// until every slot that shares the state (one, or the two of a coalesced pair) has collected its rows.
static int wait_on(wm_model* m, wm_state* s, int32_t* tokens_out, int32_t* n_tokens, int row0 = 0, int rows = -1, int32_t* dev_packed = nullptr,
                   int rows_cap = 0, int pack_stride = 0) {
    if (!s || !s->pending) return fail(WM_E_STATE, "nothing was submitted on this slot");
    HIPCHK(hipSetDevice(m->device));
    if (rows < 0) rows = s->B;
    if (!s->synced) {
        if (!s->enq_done.load()) {  // the loop pump is still feeding this pass
            std::unique_lock<std::mutex> lk(m->pump_mu);
            m->pump_cv.wait(lk, [&] { return s->enq_done.load(); });
        }
        if (s->enq_rc) {
            s->pending = false;
            for (auto& ln : s->lanes) (void)hipStreamSynchronize(ln.st);
            return fail(s->enq_rc, "%s", s->enq_err.c_str());
        }
        for (auto& ln : s->lanes) HIPCHK(hipEventSynchronize(ln.done));
        s->synced = true;
    }
    const int total = s->pend_total;
    if (dev_packed) {  // the gather buffer, built on the device in the caller's DEVICE memory (no host round trip before the collective)
        launch_pack_tokens(s->out_tokens.as<int>() + (size_t)row0 * s->out_stride, s->n_tokens.as<int>() + row0, s->out_stride, rows, rows_cap,
                           pack_stride, dev_packed, s->lanes[0].st);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(s->lanes[0].st));  // the caller's collective runs on another stream
    } else {
        HIPCHK(hipMemcpy2D(tokens_out, (size_t)total * 4, s->out_tokens.as<int>() + (size_t)row0 * s->out_stride, (size_t)s->out_stride * 4, (size_t)total * 4, rows,
                           hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(n_tokens, s->n_tokens.as<int>() + row0, (size_t)rows * 4, hipMemcpyDeviceToHost));
    }
    if (--s->halves_left <= 0) {
        s->pending = false;
        s->has_enc = false;  // the KV cache now holds a finished decode: a new wm_encode is needed before wm_decode_step
    }
    return 0;
}

static wm_state** slot_state(wm_model* m, int slot) { return slot == 0 ? &m->cached : &m->slots[slot - 1]; }

// ---- coalescing of consecutive submits (wm_config.coalesce == 2) ------------------------------------------------------------------
static bool same_opts(const wm_model::Held& h, const wm_decode_opts* o) {
    const wm_decode_opts& a = h.o;
    if (a.n_prompt != o->n_prompt || a.eot != o->eot || a.max_loop != o->max_loop || a.pos_mode != o->pos_mode || a.ignore_eot != o->ignore_eot ||
        a.n_suppress != (o->suppress_tokens ? o->n_suppress : 0) || a.n_begin_suppress != (o->begin_suppress_tokens ? o->n_begin_suppress : 0) ||
        a.timestamp_begin != o->timestamp_begin || a.no_timestamps_token != o->no_timestamps_token ||
        a.max_initial_timestamp_index != o->max_initial_timestamp_index)
        return false;
    return std::equal(h.prompt.begin(), h.prompt.end(), o->prompt) && std::equal(h.sup.begin(), h.sup.end(), o->suppress_tokens) &&
           std::equal(h.bsup.begin(), h.bsup.end(), o->begin_suppress_tokens);
}
static void hold(wm_model* m, int slot, const float* mel, int on_dev, int B, const wm_decode_opts* o) {
    wm_model::Held& h = m->held;
    h.active = true;
    h.slot = slot;
    h.B = B;
    h.on_dev = on_dev;
    h.mel = mel;
    h.prompt.assign(o->prompt, o->prompt + o->n_prompt);
    h.sup.assign(o->suppress_tokens, o->suppress_tokens + (o->suppress_tokens ? o->n_suppress : 0));
    h.bsup.assign(o->begin_suppress_tokens, o->begin_suppress_tokens + (o->begin_suppress_tokens ? o->n_begin_suppress : 0));
    h.o = *o;
    h.o.prompt = h.prompt.data();
    h.o.suppress_tokens = h.sup.empty() ? nullptr : h.sup.data();
    h.o.n_suppress = (int)h.sup.size();
    h.o.begin_suppress_tokens = h.bsup.empty() ? nullptr : h.bsup.data();
    h.o.n_begin_suppress = (int)h.bsup.size();
    m->slot_ref[slot] = wm_model::SlotRef{true, nullptr, 0, B, o->n_prompt + 1 + o->max_loop};
}
// the held submit runs alone, on its own slot's state (no partner came, or the partner did not match)
static int flush_held(wm_model* m) {
    wm_model::Held& h = m->held;
    if (!h.active) return 0;
    h.active = false;
    wm_model::SlotRef& r = m->slot_ref[h.slot];
    const int rc = submit_on(m, slot_state(m, h.slot), h.mel, h.on_dev, h.B, &h.o, false);
    if (rc) {
        r = wm_model::SlotRef{};
        return rc;
    }
    r.st = *slot_state(m, h.slot);
    r.row0 = 0;
    return 0;
}

extern "C" int wm_transcribe(wm_model* m, const float* mel, int mel_on_device, int B, const wm_decode_opts* o,
                             int32_t* tokens_out, int32_t* n_tokens) {
    if (!m || !mel || !tokens_out || !n_tokens) return fail(WM_E_ARG, "bad argument");
    WMCHK(check_opts(m, o, B));
    WMCHK(flush_held(m));  // a held submit goes first: this call may use its mel buffers' stream order, and slot 0
    if (m->slot_ref[0].pending) return fail(WM_E_STATE, "this slot still holds a pass that was not waited for");
    WMCHK(submit_on(m, &m->cached, mel, mel_on_device, B, o, true));
    WMCHK(wait_on(m, m->cached, tokens_out, n_tokens));
    m->last_steps[0] = m->cached->last_steps;
    return 0;
}

// Pipelined form of Whisper.transcribe for back-to-back batches: submit enqueues the encoder and the greedy loop on the slot's
// stream and returns; wait blocks until that slot's tokens are ready.
extern "C" int wm_transcribe_submit(wm_model* m, int slot, const float* mel, int mel_on_device, int B, const wm_decode_opts* o) {
    if (!m || !mel || slot < 0 || slot >= wm_model::NSLOT) return fail(WM_E_ARG, "bad argument (slot must be 0..7)");
    WMCHK(check_opts(m, o, B));
    wm_model::SlotRef& r = m->slot_ref[slot];
    if (r.pending) return fail(WM_E_STATE, "this slot still holds a pass that was not waited for");
    const int total = o->n_prompt + 1 + o->max_loop;
    const bool can_pair = m->cfg.coalesce == 2 && B <= m->cfg.max_batch && (B <= m->enc_chunk || B % m->enc_chunk == 0);
    if (can_pair && m->held.active && m->held.B == B && same_opts(m->held, o)) {
        // the partner of the held submit: both batches go out as ONE pass on a 2·B-row state
        wm_state** ps = nullptr;
        for (auto& pr : m->pairs)
            if (pr && !pr->pending && pr->B == 2 * B) ps = &pr;
        for (auto& pr : m->pairs)
            if (!ps && !pr) ps = &pr;
        for (auto& pr : m->pairs)
            if (!ps && !pr->pending) ps = &pr;  // another batch size: re-created
        if (ps) {
            wm_model::Held& h = m->held;
            h.active = false;
            wm_model::SlotRef& r0 = m->slot_ref[h.slot];
            const int rc = submit_on(m, ps, h.mel, h.on_dev, 2 * B, &h.o, false, mel, mel_on_device);
            if (rc) {
                r0 = wm_model::SlotRef{};
                return rc;
            }
            r0.st = *ps;
            r0.row0 = 0;
            r = wm_model::SlotRef{true, *ps, B, B, total};
            return 0;
        }
    }
    WMCHK(flush_held(m));
    if (can_pair) {  // wait for a partner (or for this slot's wm_transcribe_wait)
        hold(m, slot, mel, mel_on_device, B, o);
        return 0;
    }
    WMCHK(submit_on(m, slot_state(m, slot), mel, mel_on_device, B, o, false));
    r = wm_model::SlotRef{true, *slot_state(m, slot), 0, B, total};
    return 0;
}
extern "C" int wm_transcribe_wait(wm_model* m, int slot, int32_t* tokens_out, int32_t* n_tokens) {
    if (!m || !tokens_out || !n_tokens || slot < 0 || slot >= wm_model::NSLOT) return fail(WM_E_ARG, "bad argument");
    wm_model::SlotRef& r = m->slot_ref[slot];
    if (!r.pending) return fail(WM_E_STATE, "nothing was submitted on this slot");
    if (m->held.active && m->held.slot == slot) {  // no partner came: the held batch runs alone now
        const int rc = flush_held(m);
        if (rc) return rc;
    }
    wm_state* s = r.st;
    const int rc = wait_on(m, s, tokens_out, n_tokens, r.row0, r.rows);
    if (!rc) m->last_steps[slot] = s->last_steps;
    r = wm_model::SlotRef{};
    return rc;
}
// wm_transcribe_wait with the result left ON THE DEVICE as the gather buffer of the multi-GPU path: dev_packed
// [rows_cap, 1 + stride] int32 in the caller's device memory (e.g. a torch tensor), row r = [length, ids zero-padded]; rows past the
// batch are zeroed.  stride >= n_prompt + 1 + max_loop of the pass.
extern "C" int wm_transcribe_wait_device(wm_model* m, int slot, int32_t* dev_packed, int rows_cap, int stride) {
    if (!m || !dev_packed || slot < 0 || slot >= wm_model::NSLOT || stride <= 0) return fail(WM_E_ARG, "bad argument");
    wm_model::SlotRef& r = m->slot_ref[slot];
    if (!r.pending) return fail(WM_E_STATE, "nothing was submitted on this slot");
    if (rows_cap < r.rows) return fail(WM_E_ARG, "rows_cap %d is smaller than the batch (%d)", rows_cap, r.rows);
    if (stride < r.total) return fail(WM_E_ARG, "stride %d is smaller than the pass's ids per utterance (%d)", stride, r.total);
    if (m->held.active && m->held.slot == slot) {
        const int rc = flush_held(m);
        if (rc) return rc;
    }
    wm_state* s = r.st;
    const int rc = wait_on(m, s, nullptr, nullptr, r.row0, r.rows, dev_packed, rows_cap, stride);
    if (!rc) m->last_steps[slot] = s->last_steps;
    r = wm_model::SlotRef{};
    return rc;
}
extern "C" int wm_transcribe_steps(wm_model* m, int slot) {
    if (!m || slot < 0 || slot >= wm_model::NSLOT) return -1;
    return m->last_steps[slot];
}

// ---- log-mel front end: 16 kHz PCM -> [n_mels, n_frames]  (export_weights.py:100-116 delegates this to
// HF WhisperProcessor = transformers feature_extraction_whisper._np_extract_fbank_features)
static const int FE_NFFT = 400, FE_HOP = 160, FE_NFREQ = 201;

static double hz_to_mel_slaney(double f) { return f >= 1000.0 ? 15.0 + std::log(f / 1000.0) * (27.0 / std::log(6.4)) : 3.0 * f / 200.0; }
static double mel_to_hz_slaney(double m) { return m >= 15.0 ? 1000.0 * std::exp(std::log(6.4) / 27.0 * (m - 15.0)) : 200.0 * m / 3.0; }

static int frontend_init(wm_model* m) {
    if (m->fe.ready) return 0;
    const wm_dims& c = m->cfg.dims;
    const int n_mels = c.n_mels, n_frames = 2 * c.n_audio_ctx, N = FE_HOP * n_frames, maxB = m->cfg.max_batch;
    const double PI = 3.14159265358979323846;
    std::vector<float> window(FE_NFFT), dft((size_t)512 * 416, 0.f), fb((size_t)FE_NFREQ * n_mels, 0.f);
    for (int n = 0; n < FE_NFFT; ++n) window[n] = (float)(0.5 - 0.5 * std::cos(2.0 * PI * n / FE_NFFT));  // periodic Hann
    for (int k = 0; k < FE_NFREQ; ++k)
        for (int n = 0; n < FE_NFFT; ++n) {
            const double ang = 2.0 * PI * (double)((k * n) % FE_NFFT) / FE_NFFT;
            dft[(size_t)k * 416 + n] = (float)std::cos(ang);
            dft[(size_t)(256 + k) * 416 + n] = (float)(-std::sin(ang));
        }
    // slaney-scale, slaney-normalised triangular filters (transformers.audio_utils.mel_filter_bank)
    std::vector<double> ff(n_mels + 2);
    const double m0 = hz_to_mel_slaney(0.0), m1 = hz_to_mel_slaney(8000.0);
    for (int i = 0; i < n_mels + 2; ++i) ff[i] = mel_to_hz_slaney(m0 + (m1 - m0) * i / (n_mels + 1));
    std::vector<int> band(2 * n_mels);
    for (int mm = 0; mm < n_mels; ++mm) {
        int lo = FE_NFREQ, hi = 0;
        const double enorm = 2.0 / (ff[mm + 2] - ff[mm]);
        for (int k = 0; k < FE_NFREQ; ++k) {
            const double f = 8000.0 * k / (FE_NFREQ - 1);
            const double down = (f - ff[mm]) / (ff[mm + 1] - ff[mm]), up = (ff[mm + 2] - f) / (ff[mm + 2] - ff[mm + 1]);
            const double v = std::max(0.0, std::min(down, up)) * enorm;
            fb[(size_t)k * n_mels + mm] = (float)v;
            if (v > 0.0) {
                lo = std::min(lo, k);
                hi = std::max(hi, k + 1);
            }
        }
        band[2 * mm] = lo < hi ? lo : 0;
        band[2 * mm + 1] = lo < hi ? hi : 0;
    }
    WMCHK(upload(m->fe.window, window.data(), window.size(), WM_F32));
    WMCHK(upload(m->fe.dft, dft.data(), dft.size(), WM_F32));
    WMCHK(upload(m->fe.fb, fb.data(), fb.size(), WM_F32));
    WMCHK(m->fe.band.alloc(band.size() * 4));
    HIPCHK(hipMemcpy(m->fe.band.p, band.data(), band.size() * 4, hipMemcpyHostToDevice));
    const size_t ch = std::min(maxB, m->fe.chunk);
    const size_t rows = (ch * n_frames + 255) / 256 * 256 + 256;
    WMCHK(m->fe.pcm.alloc((size_t)maxB * N * 4, true));
    WMCHK(m->fe.lens.alloc((size_t)maxB * 4, true));
    WMCHK(m->fe.frames.alloc(rows * 416 * 4, true));
    WMCHK(m->fe.spec.alloc(rows * 512 * 4, true));
    WMCHK(m->fe.logtmp.alloc((size_t)maxB * n_mels * n_frames * 4));
    WMCHK(m->fe.mel.alloc((size_t)maxB * n_mels * n_frames * 4));
    m->fe.ready = true;
    return 0;
}

// pcm: host [B][stride] fp32 at 16 kHz; n_samples[b] valid samples (<= stride).  Result stays in m->fe.mel (device).
static int frontend_run(wm_model* m, const float* pcm, const int32_t* n_samples, int B, int stride) {
    if (!pcm || !n_samples || B <= 0 || stride <= 0) return fail(WM_E_ARG, "bad argument");
    if (B > m->cfg.max_batch) return fail(WM_E_ARG, "batch %d exceeds max_batch %d", B, m->cfg.max_batch);
    HIPCHK(hipSetDevice(m->device));
    WMCHK(frontend_init(m));
    const wm_dims& c = m->cfg.dims;
    const int n_frames = 2 * c.n_audio_ctx, N = FE_HOP * n_frames;
    hipStream_t st = m->stream;
    // pad / trim to the 30 s window: ONE strided upload of the common prefix, then a kernel zeroes each utterance's tail
    for (int b = 0; b < B; ++b)
        if (n_samples[b] < 0 || n_samples[b] > stride) return fail(WM_E_ARG, "n_samples[%d]=%d out of range", b, n_samples[b]);
    const size_t w = std::min(stride, N);
    if (w < (size_t)N) HIPCHK(hipMemsetAsync(m->fe.pcm.p, 0, (size_t)B * N * 4, st));
    HIPCHK(hipMemcpy2DAsync(m->fe.pcm.p, (size_t)N * 4, pcm, (size_t)stride * 4, w * 4, B, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(m->fe.lens.p, n_samples, (size_t)B * 4, hipMemcpyHostToDevice, st));
    launch_zero_tails(m->fe.pcm.as<float>(), m->fe.lens.as<int>(), B, N, st);
    for (int c0 = 0; c0 < B; c0 += m->fe.chunk) {
        const int bc = std::min(m->fe.chunk, B - c0);
        launch_frames(m->fe.pcm.as<float>() + (size_t)c0 * N, m->fe.frames.as<float>(), m->fe.window.as<float>(), bc, N, n_frames, FE_HOP, st);
        GemmParams p{};
        p.A = m->fe.frames.p;
        p.W = m->fe.dft.p;
        p.C = m->fe.spec.p;
        p.M = bc * n_frames;
        p.N = 512;
        p.K = 416;
        p.lda = 416;
        p.ldw = 416;
        p.ldc = 512;
        LCHK((launch_gemm_nt<float, float>(p, 1, st)));
        launch_mel_log(m->fe.spec.as<float>(), m->fe.fb.as<float>(), m->fe.band.as<int>(), m->fe.logtmp.as<float>() + (size_t)c0 * c.n_mels * n_frames,
                       bc, n_frames, c.n_mels, st);
    }
    launch_mel_norm(m->fe.logtmp.as<float>(), m->fe.mel.as<float>(), B, c.n_mels * n_frames, st);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int wm_log_mel(wm_model* m, const float* pcm, const int32_t* n_samples, int B, int stride, float* mel_out) {
    if (!m) return fail(WM_E_ARG, "null model");
    WMCHK(frontend_run(m, pcm, n_samples, B, stride));
    const wm_dims& c = m->cfg.dims;
    if (mel_out) HIPCHK(hipMemcpyAsync(mel_out, m->fe.mel.p, (size_t)B * c.n_mels * 2 * c.n_audio_ctx * 4, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    return 0;
}

extern "C" int wm_transcribe_pcm(wm_model* m, const float* pcm, const int32_t* n_samples, int B, int stride, const wm_decode_opts* o,
                                 int32_t* tokens_out, int32_t* n_tokens) {
    if (!m || !tokens_out || !n_tokens) return fail(WM_E_ARG, "bad argument");
    WMCHK(check_opts(m, o, B));
    WMCHK(frontend_run(m, pcm, n_samples, B, stride));  // same stream as the encoder: ordered, no host sync
    WMCHK(flush_held(m));
    if (m->slot_ref[0].pending) return fail(WM_E_STATE, "this slot still holds a pass that was not waited for");
    WMCHK(submit_on(m, &m->cached, m->fe.mel.as<float>(), 1, B, o, true));
    WMCHK(wait_on(m, m->cached, tokens_out, n_tokens));
    m->last_steps[0] = m->cached->last_steps;
    return 0;
}

// ---- measurement helpers ------------------------------------------------------------------------------------------------
// A freshly encoded state has an empty self-attention cache; the decode step is priced AND timed mid-sequence, at this many
// cached rows (SURVEY §8d quotes its byte figures at t = 50; the rows of a fresh cache are zero, which the timing does not care about)
static const int BENCH_STEP_LEN = 50;
extern "C" int wm_bench_bytes(wm_model* m, wm_state* s, int which, double* bytes) {
    if (!m || !s || !bytes) return fail(WM_E_ARG, "null argument");
    if (!state_is_live(s) || s->m != m) return fail(WM_E_ARG, "stale or foreign state handle");
    const wm_dims& c = m->cfg.dims;
    const double d = c.d_model, H = c.n_heads, B = s->B;
    const double ks = dt_size(m->cfg.kv_dtype), ws = dt_size(dec_dtype(m->cfg));
    if (which == WM_KERNEL_CROSS_ATTN) {
        // one layer: K and V rows of every utterance once + q in + partials out
        *bytes = B * 2.0 * c.n_audio_ctx * d * ks + B * d * 4 + B * s->nsplit * (d + 2 * H) * 4;
    } else if (which == WM_KERNEL_DECODE_STEP || which == WM_KERNEL_DECODE_STEP_SHARED) {
        // every weight once per step, KV once per utterance, KV write; logits are NOT materialised
        // (fused argmax: only B x ceil(V/128) (value, index) partials are written and re-read)
        const double f = c.ffn, L = c.n_layers, V = c.vocab;
        const double p_blk = 8 * d * d + 2 * f * d + (4 + 4 + 1 + 1 + 6) * d + f;  // 2 attn (4 mats each) + mlp + biases + 3 LN
        const double t = s->host_len > 0 ? s->host_len : BENCH_STEP_LEN;
        *bytes = ws * (L * (8 * d * d + 2 * f * d) + V * d) + 4 * (L * (p_blk - 8 * d * d - 2 * f * d) + 2 * d) +
                 B * L * 2 * d * ks * (c.n_audio_ctx + t) + B * L * 2 * d * ks + B * s->npart * 8.0 * 2 + B * 4;
    } else if (which == WM_KERNEL_ENCODER) {
        *bytes = 0;  // MFMA-bound: see wm_bench_flops in DESIGN.md
    } else {
        return fail(WM_E_ARG, "unknown kernel id %d", which);
    }
    return 0;
}

extern "C" int wm_bench_kernel(wm_model* m, wm_state* s, int which, int reps, float* avg_us) {
    if (!m || !s || !avg_us || reps <= 0) return fail(WM_E_ARG, "bad argument");
    WMCHK(check_state(m, s, -1));
    if (!s->has_cross) return fail(WM_E_STATE, "state has no cross K/V (call wm_encode first)");
    HIPCHK(hipSetDevice(m->device));
    hipStream_t st = m->stream;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    const int L = m->cfg.dims.n_layers;
    if (which == WM_KERNEL_CROSS_ATTN) {
        const DecView v = whole_batch(m, s);
        for (int i = 0; i < L; ++i) WMCHK(launch_cross_attn(m, s, i, v));  // warm-up
        HIPCHK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) WMCHK(launch_cross_attn(m, s, i % L, v));  // cycles the layers: 4 x 295 MB > 256 MB L3
        HIPCHK(hipEventRecord(e1, st));
    } else if (which == WM_KERNEL_DECODE_STEP || which == WM_KERNEL_DECODE_STEP_SHARED) {
        s->shares_chip = which == WM_KERNEL_DECODE_STEP_SHARED;  // the K/V stream as pipelined passes launch it
        // on the state's own decode stream, as the transcribe loop runs it: several states can be timed concurrently
        // from several host threads (bench.py: four chains in flight)
        const int len0 = s->host_len > 0 ? s->host_len : BENCH_STEP_LEN;  // the cache length wm_bench_bytes prices
        HIPCHK(hipStreamSynchronize(m->stream));  // wm_encode ran there
        const bool own = s->lanes.size() == 1;      // (WM_DEC_LANES > 1: whole batch on the model stream as before)
        if (own) st = s->lanes[0].st;
        const DecView v{0, s->B, st, own ? s->lanes[0].ctl : s->ctl.as<StepCtl>()};
        launch_set_step(v.ctl, len0, 1, nullptr, 0, nullptr, 0, s->B, st);
        WMCHK(decode_core(m, s, v, true));
        // timed as the transcribe loop runs it: a captured graph of the step, replayed (cache length held constant)
        hipGraph_t g = nullptr;
        hipGraphExec_t ge = nullptr;
        HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        const int crc = decode_core(m, s, v, true);
        const hipError_t cap = hipStreamEndCapture(st, &g);
        if (crc) {
            if (g) (void)hipGraphDestroy(g);
            return crc;
        }
        HIPCHK(cap);
        HIPCHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        (void)hipGraphDestroy(g);
        HIPCHK(hipGraphLaunch(ge, st));
        HIPCHK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) HIPCHK(hipGraphLaunch(ge, st));
        HIPCHK(hipEventRecord(e1, st));
        HIPCHK(hipEventSynchronize(e1));
        (void)hipGraphExecDestroy(ge);
        s->shares_chip = false;
#ifdef WM_DEV
    } else if (which == 41 || which == 42) {  // developer: phase stamps of one LN1 + QKV (41) / O-proj (42) launch
        const wm_dims& c = m->cfg.dims;
        const int T = dec_dtype(m->cfg);
        DecLayer& w0 = m->dec[0];
        DevBuf dbg;
        WMCHK(dbg.alloc((size_t)4096 * 16 * 8 * 8, true));
        DecLinearParams p{};
        p.B = s->B;
        if (which == 41) {
            p.x = s->dx.as<float>(); p.ldx = c.d_model; p.ln_g = w0.ln1_g.as<float>(); p.ln_b = w0.ln1_b.as<float>();
            p.W = w0.sqkv_w.p; p.N = 3 * c.d_model; p.K = c.d_model; p.bias = w0.sqkv_b.as<float>(); p.out = s->dq.as<float>();
            p.ldo = c.d_model; p.kcache = s->self_kv.p; p.vcache = off_bytes(s->self_kv, (size_t)s->B * c.n_text_ctx * c.d_model * dt_size(m->cfg.kv_dtype));
            p.kv_batch_stride = (long)((size_t)c.n_text_ctx * c.d_model); p.d_model = c.d_model; p.kv_dtype = m->cfg.kv_dtype; p.ctl = s->ctl.as<StepCtl>();
        } else {
            p.x = s->dattn.as<float>(); p.ldx = c.d_model; p.x_is_t = 1; p.W = w0.so_w.p; p.N = c.d_model; p.K = c.d_model; p.bias = w0.so_b.as<float>();
            p.residual = s->dx.as<float>(); p.ldr = c.d_model; p.out = s->dx.as<float>(); p.ldo = c.d_model;
        }
        launch_set_step(s->ctl.as<StepCtl>(), 10, 1, nullptr, 0, nullptr, 0, s->B, st);
        for (int i = 0; i < 3; ++i) { WMCHK(dec_linear_dispatch(T, p, st)); if (!wm_env("WM_STAMP_NO_STREAM")) WMCHK(launch_cross_attn(m, s, i, whole_batch(m, s))); }
        p.dbg = dbg.as<long long>();
        HIPCHK(hipEventRecord(e0, st));
        WMCHK(dec_linear_dispatch(T, p, st));
        HIPCHK(hipEventRecord(e1, st));
        HIPCHK(hipEventSynchronize(e1));
        std::vector<long long> h((size_t)4096 * 16 * 8);
        HIPCHK(hipMemcpy(h.data(), dbg.p, h.size() * 8, hipMemcpyDeviceToHost));
        long long t0 = -1;
        for (size_t g = 0; g < h.size() / 8; ++g) if (h[g * 8] > 0 && (t0 < 0 || h[g * 8] < t0)) t0 = h[g * 8];
        double mx[6] = {0}, av[6] = {0};
        int n = 0;
        for (size_t g = 0; g < h.size() / 8; ++g) {
            if (h[g * 8] <= 0) continue;
            ++n;
            for (int k = 0; k < 6; ++k) {
                const double us = (double)(h[g * 8 + k] - t0) / 100.0;
                av[k] += us;
                mx[k] = std::max(mx[k], us);
            }
        }
        fprintf(stderr, "[wm] dec_linear %s phases (us after the first wave's entry; mean / max over %d waves): entry %.2f/%.2f  loads issued %.2f/%.2f  operands ready %.2f/%.2f  MFMA + partial stored %.2f/%.2f  after barrier %.2f/%.2f  end %.2f/%.2f\n",
                which == 41 ? "LN1+QKV" : "O-proj", n, av[0] / n, mx[0], av[1] / n, mx[1], av[2] / n, mx[2], av[3] / n, mx[3], av[4] / n, mx[4], av[5] / n, mx[5]);
        dbg.release();
    } else if (which == 43) {  // developer: phase stamps of the encoder's QKV row-panel GEMM (layer 0) on the rows of the last encode
        const wm_dims& c = m->cfg.dims;
        const int T = m->cfg.compute_dtype;
        const int M = std::min(s->B, s->Bc) * c.n_audio_ctx;
        EncLayer& w0 = m->enc[0];
        GemmParams p{};
        p.A = s->xn.p; p.W = w0.qkv_w.p; p.C = s->qkv.p; p.M = M; p.N = 3 * c.d_model; p.K = c.d_model;
        p.lda = c.d_model; p.ldw = c.d_model; p.ldc = 3 * c.d_model; p.bias = w0.qkv_b.as<float>();
        if (!gemm_nt_fuses_layernorm(T == WM_F32 ? 4 : 2, p, 1)) return fail(WM_E_ARG, "this configuration does not run the row-panel kernel");
        for (int i = 0; i < 3; ++i) WMCHK(gemm_dispatch(T, T, p, 1, st));
        DevBuf dbg;
        WMCHK(dbg.alloc((size_t)512 * 96 * 8, true));
        p.dbg = dbg.as<long long>();
        HIPCHK(hipEventRecord(e0, st));
        WMCHK(gemm_dispatch(T, T, p, 1, st));
        HIPCHK(hipEventRecord(e1, st));
        HIPCHK(hipEventSynchronize(e1));
        std::vector<long long> h((size_t)512 * 96);
        HIPCHK(hipMemcpy(h.data(), dbg.p, h.size() * 8, hipMemcpyDeviceToHost));
        double wait = 0, issue = 0, comp = 0, epi = 0, gap = 0, unit = 0;
        int ns = 0, nu = 0, ng = 0;
        for (int g = 0; g < 512; ++g)
            for (int uu = 1; uu < 3; ++uu) {  // second and third unit of each workgroup (the first carries the panel load)
                const long long* d = &h[(size_t)g * 96 + uu * 32];
                if (d[0] <= 0 || d[25] <= 0) continue;
                for (int t = 0; t < 6; ++t) {
                    wait += (double)(d[t * 4 + 1] - d[t * 4 + 0]);
                    issue += (double)(d[t * 4 + 2] - d[t * 4 + 1]);
                    comp += (double)(d[t * 4 + 3] - d[t * 4 + 2]);
                    if (t > 0) { gap += (double)(d[t * 4] - d[t * 4 - 1]); ++ng; }
                    ++ns;
                }
                epi += (double)(d[25] - d[24]);
                unit += (double)(d[25] - d[0]);
                ++nu;
            }
        if (ns)
            fprintf(stderr, "[wm] row-panel QKV, wave 0, per 64-k stage (ns): wait + barrier %.0f  DMA issue %.0f  fragment reads + 32 MFMAs issued %.0f  to next top %.0f | "
                            "per unit: epilogue %.0f, whole unit %.0f (%d stages, %d units)\n",
                    wait / ns * 10, issue / ns * 10, comp / ns * 10, ng ? gap / ng * 10 : 0.0, epi / nu * 10, unit / nu * 10, ns, nu);
        dbg.release();
    } else if (which == 40) {  // developer: phase stamps of one logits launch (after a few warm ones), printed to stderr
        const wm_dims& c = m->cfg.dims;
        const int T = dec_dtype(m->cfg);
        DevBuf dbg;
        const int nwg = s->npart;
        WMCHK(dbg.alloc((size_t)nwg * 8 * 8 * 8, true));
        DecLinearParams p{};
        p.x = s->dx.as<float>(); p.ldx = c.d_model; p.ln_g = m->dec_ln_g.as<float>(); p.ln_b = m->dec_ln_b.as<float>();
        p.W = T == WM_F32 ? m->tok_emb_f.p : m->tok_emb_t.p; p.N = c.vocab; p.K = c.d_model; p.B = s->B; p.ldo = m->Vpad;
        p.amax_val = s->amax_val.as<float>(); p.amax_idx = s->amax_idx.as<int>(); p.amax_stride = s->npart;
        for (int i = 0; i < 3; ++i) { DISPATCH_DT(T, TT, (void)launch_dec_logits<TT>(p, st)); WMCHK(launch_cross_attn(m, s, i, whole_batch(m, s))); }
        p.dbg = dbg.as<long long>();
        HIPCHK(hipEventRecord(e0, st));
        DISPATCH_DT(T, TT, (void)launch_dec_logits<TT>(p, st));
        HIPCHK(hipEventRecord(e1, st));
        HIPCHK(hipEventSynchronize(e1));
        std::vector<long long> h((size_t)nwg * 64);
        HIPCHK(hipMemcpy(h.data(), dbg.p, h.size() * 8, hipMemcpyDeviceToHost));
        long long t0 = -1;
        for (int g = 0; g < nwg * 8; ++g) if (h[(size_t)g * 8] > 0 && (t0 < 0 || h[(size_t)g * 8] < t0)) t0 = h[(size_t)g * 8];
        double mx[8] = {0, 0, 0, 0, 0, 0, 0, 0}, av[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int n = 0;
        for (int g = 0; g < nwg * 8; ++g) {
            if (h[(size_t)g * 8] <= 0) continue;
            ++n;
            for (int k = 0; k < 7; ++k) {
                const double us = (double)(h[(size_t)g * 8 + k] - t0) / 100.0;
                av[k] += us;
                mx[k] = std::max(mx[k], us);
            }
        }
        fprintf(stderr, "[wm] logits phases (us after the first wave's entry; mean / max over %d waves): entry %.2f/%.2f  loads issued + staged %.2f/%.2f  after barrier %.2f/%.2f  MFMA done %.2f/%.2f  end %.2f/%.2f\n",
                n, av[0] / n, mx[0], av[1] / n, mx[1], av[2] / n, mx[2], av[3] / n, mx[3], av[4] / n, mx[4]);
        fprintf(stderr, "[wm]   gamma/beta + activations arrived (own wave) %.2f/%.2f  all waves of the workgroup %.2f/%.2f\n", av[5] / n, mx[5], av[6] / n, mx[6]);
        dbg.release();
#endif
    } else if (which == WM_KERNEL_ENCODER) {
        if (!s->last_mel) return fail(WM_E_STATE, "no mel was encoded into this state");
        WMCHK(run_encoder(m, s, s->last_mel, s->B, st, nullptr, 0, false));  // as the transcribe paths run it
        HIPCHK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) WMCHK(run_encoder(m, s, s->last_mel, s->B, st, nullptr, 0, false));
        HIPCHK(hipEventRecord(e1, st));
    } else {
        return fail(WM_E_ARG, "unknown kernel id %d", which);
    }
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *avg_us = ms * 1000.0f / (float)reps;
    return 0;
}

// ---- op-level entry points (whisper_tensor.mojo) --------------------------------------------------------------------------
struct TmpDev {
    std::vector<DevBuf> bufs;
    ~TmpDev() {
        for (auto& b : bufs) b.release();
    }
    DevBuf& add() {
        bufs.emplace_back();
        return bufs.back();
    }
};

extern "C" int wm_op_matmul_nt(float* C, const float* A, const float* Bm, const float* bias, int M, int N, int K, int dtype) {
    if (!C || !A || !Bm || M <= 0 || N <= 0 || K <= 0) return fail(WM_E_ARG, "bad argument");
    if (K % 32) return fail(WM_E_ARG, "K must be a multiple of 32");
    if (dtype < 0 || dtype > 2) return fail(WM_E_ARG, "bad dtype");
    TmpDev t;
    t.bufs.reserve(8);
    hipStream_t st = nullptr;
    if (N % 128 == 0) {  // dense path: the encoder GEMM kernel
        const size_t Mp = ((size_t)M + 127) / 128 * 128;
        std::vector<float> Ap(Mp * K, 0.f);
        memcpy(Ap.data(), A, (size_t)M * K * 4);
        DevBuf &a = t.add(), &w = t.add(), &c = t.add(), &b = t.add();
        WMCHK(upload(a, Ap.data(), Ap.size(), dtype));
        WMCHK(upload(w, Bm, (size_t)N * K, dtype));
        WMCHK(c.alloc((size_t)M * N * 4));
        if (bias) WMCHK(upload(b, bias, N, WM_F32));
        GemmParams p{};
        p.A = a.p;
        p.W = w.p;
        p.C = c.p;
        p.M = M;
        p.N = N;
        p.K = K;
        p.lda = K;
        p.ldw = K;
        p.ldc = N;
        p.bias = bias ? b.as<float>() : nullptr;
        WMCHK(gemm_dispatch(dtype, WM_F32, p, 1, st));
        HIPCHK(hipMemcpy(C, c.p, (size_t)M * N * 4, hipMemcpyDeviceToHost));
    } else {  // skinny path: the decode-step linear kernel
        const int Np = (N + 15) / 16 * 16;
        // the kernel splits K over NW <= 16 waves, <= 4 k-steps of 32 each: zero-pad K to the next such size
        auto splittable = [](int k) {
            const int ks = k / 32;
            for (int c = 16; c >= 1; --c)
                if (ks % c == 0 && ks / c <= 4) return true;
            return false;
        };
        int Kp = (K + 127) / 128 * 128;
        while (Kp <= 2048 && !splittable(Kp)) Kp += 128;  // (bounded: no K above 2048 splits, and an unbounded search overflowed)
        if (Kp > 2048) return fail(WM_E_ARG, "K too large for the skinny path (<= 2048)");
        std::vector<float> Apad((size_t)M * Kp, 0.f), Bpad((size_t)N * Kp, 0.f);
        for (int i = 0; i < M; ++i) memcpy(&Apad[(size_t)i * Kp], A + (size_t)i * K, (size_t)K * 4);
        for (int i = 0; i < N; ++i) memcpy(&Bpad[(size_t)i * Kp], Bm + (size_t)i * K, (size_t)K * 4);
        K = Kp;
        DevBuf &a = t.add(), &w = t.add(), &c = t.add(), &b = t.add();
        WMCHK(upload(a, Apad.data(), Apad.size(), WM_F32));
        WMCHK(upload(w, Bpad.data(), Bpad.size(), dtype));
        WMCHK(c.alloc((size_t)M * Np * 4));
        if (bias) WMCHK(upload(b, bias, N, WM_F32));
        DecLinearParams p{};
        p.x = a.as<float>();
        p.ldx = K;
        p.W = w.p;
        p.N = N;
        p.K = K;
        p.B = M;
        p.bias = bias ? b.as<float>() : nullptr;
        p.out = c.as<float>();
        p.ldo = Np;
        WMCHK(dec_linear_dispatch(dtype, p, st));
        HIPCHK(hipMemcpy2D(C, (size_t)N * 4, c.p, (size_t)Np * 4, (size_t)N * 4, M, hipMemcpyDeviceToHost));
    }
    HIPCHK(hipGetLastError());
    return 0;
}

// C = layer_norm(A, ln_g, ln_b) · Bᵀ (+ bias): the LN -> projection pair of ResidualAttentionBlock.forward (layers.mojo:449-455,
// 489-497) on the encoder's kernels.  require_fused != 0 demands the ONE-kernel form (LayerNorm applied while the row-panel GEMM
// loads its fp32 A rows): a shape that kernel does not take is refused by the launcher — WM_E_ARG, nothing launched, C untouched.
extern "C" int wm_op_ln_matmul_nt(float* C, const float* A, const float* ln_g, const float* ln_b, const float* Bm, const float* bias,
                                  int M, int N, int K, int dtype, int require_fused) {
    if (!C || !A || !ln_g || !ln_b || !Bm || M <= 0 || N <= 0 || K <= 0) return fail(WM_E_ARG, "bad argument");
    if (dtype < 0 || dtype > 2) return fail(WM_E_ARG, "bad dtype");
    if (K % 128 || K > 1024) return fail(WM_E_ARG, "K must be a multiple of 128 and <= 1024 (layer_norm rows)");
    TmpDev t;
    t.bufs.reserve(8);
    hipStream_t st = nullptr;
    const size_t Mp = ((size_t)M + 127) / 128 * 128;
    std::vector<float> Ap(Mp * K, 0.f);
    memcpy(Ap.data(), A, (size_t)M * K * 4);
    DevBuf &x = t.add(), &xn = t.add(), &w = t.add(), &c = t.add(), &b = t.add(), &g = t.add(), &be = t.add();
    WMCHK(upload(x, Ap.data(), Ap.size(), WM_F32));
    WMCHK(xn.alloc(Mp * K * dt_size(dtype), true));
    WMCHK(upload(w, Bm, (size_t)N * K, dtype));
    WMCHK(c.alloc((size_t)M * N * 4));
    WMCHK(upload(g, ln_g, K, WM_F32));
    WMCHK(upload(be, ln_b, K, WM_F32));
    if (bias) WMCHK(upload(b, bias, N, WM_F32));
    GemmParams p{};
    p.A = xn.p;
    p.W = w.p;
    p.C = c.p;
    p.M = M;
    p.N = N;
    p.K = K;
    p.lda = K;
    p.ldw = K;
    p.ldc = N;
    p.bias = bias ? b.as<float>() : nullptr;
    if (require_fused) {  // handed to the launcher as asked: it either fuses or refuses
        p.A = x.p;
        p.ln_g = g.as<float>();
        p.ln_b = be.as<float>();
        WMCHK(gemm_dispatch(dtype, WM_F32, p, 1, st));
    } else {
        WMCHK(ln_then_gemm(dtype, WM_F32, p, x.as<float>(), g.as<float>(), be.as<float>(), st));
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(C, c.p, (size_t)M * N * 4, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int wm_op_layer_norm(float* out, const float* inp, const float* gamma, const float* beta, int rows, int cols, float eps) {
    if (!out || !inp || !gamma || !beta || rows <= 0 || cols <= 0) return fail(WM_E_ARG, "bad argument");
    if (cols % 128 || cols > 1024) return fail(WM_E_ARG, "cols must be a multiple of 128 and <= 1024");
    TmpDev t;
    t.bufs.reserve(8);
    DevBuf &x = t.add(), &g = t.add(), &b = t.add(), &o = t.add();
    WMCHK(upload(x, inp, (size_t)rows * cols, WM_F32));
    WMCHK(upload(g, gamma, cols, WM_F32));
    WMCHK(upload(b, beta, cols, WM_F32));
    WMCHK(o.alloc((size_t)rows * cols * 4));
    launch_layernorm_rows<float>(x.as<float>(), g.as<float>(), b.as<float>(), nullptr, o.as<float>(), rows, cols, eps, nullptr);
    HIPCHK(hipMemcpy(out, o.p, (size_t)rows * cols * 4, hipMemcpyDeviceToHost));
    return 0;
}

static void widen_to_f32(const void* src, int dtype, size_t n, float* dst);
extern "C" int wm_op_mlp_block(float* x, const float* ln_g, const float* ln_b, const float* fc1_w, const float* fc1_b, const float* fc2_w,
                               const float* fc2_b, const float* next_g, const float* next_b, float* xn_out, int M, int d, int ffn,
                               int dtype, int gelu_mode) {
    if (!x || !ln_g || !ln_b || !fc1_w || !fc1_b || !fc2_w || !fc2_b || M <= 0) return fail(WM_E_ARG, "bad argument");
    if (d <= 0 || d % 128 || d > 1024 || ffn <= 0 || ffn % 128) return fail(WM_E_ARG, "d and ffn must be multiples of 128 (d <= 1024)");
    if (dtype < 0 || dtype > 2 || (gelu_mode != 0 && gelu_mode != 1)) return fail(WM_E_ARG, "bad dtype / gelu_mode");
    const bool want_next = next_g && next_b && xn_out;
    if (!want_next && (next_g || next_b || xn_out)) return fail(WM_E_ARG, "next_g, next_b and xn_out go together");
    TmpDev t;
    t.bufs.reserve(16);
    hipStream_t st = nullptr;
    const size_t Mp = ((size_t)M + 127) / 128 * 128, ts = dt_size(dtype);  // the tile kernels read whole 128-row panels
    DevBuf &dx = t.add(), &xn = t.add(), &hid = t.add(), &g1 = t.add(), &b1 = t.add(), &w1 = t.add(), &bb1 = t.add(), &w2 = t.add(),
           &bb2 = t.add(), &g2 = t.add(), &b2 = t.add();
    std::vector<float> xp(Mp * d, 0.f);
    memcpy(xp.data(), x, (size_t)M * d * 4);
    WMCHK(upload(dx, xp.data(), xp.size(), WM_F32));
    WMCHK(xn.alloc(Mp * d * ts, true));
    WMCHK(hid.alloc(Mp * ffn * ts, true));
    WMCHK(upload(g1, ln_g, d, WM_F32));
    WMCHK(upload(b1, ln_b, d, WM_F32));
    WMCHK(upload(w1, fc1_w, (size_t)ffn * d, dtype));
    WMCHK(upload(bb1, fc1_b, ffn, WM_F32));
    WMCHK(upload(w2, fc2_w, (size_t)d * ffn, dtype));
    WMCHK(upload(bb2, fc2_b, d, WM_F32));
    if (want_next) {
        WMCHK(upload(g2, next_g, d, WM_F32));
        WMCHK(upload(b2, next_b, d, WM_F32));
    }
    GemmParams f1{};
    f1.A = xn.p;
    f1.W = w1.p;
    f1.C = hid.p;
    f1.M = M;
    f1.N = ffn;
    f1.K = d;
    f1.lda = d;
    f1.ldw = d;
    f1.ldc = ffn;
    f1.bias = bb1.as<float>();
    f1.act = 1;
    f1.gelu_mode = gelu_mode;
    WMCHK(ln_then_gemm(dtype, dtype, f1, dx.as<float>(), g1.as<float>(), b1.as<float>(), st));
    GemmParams f2{};
    f2.A = hid.p;
    f2.W = w2.p;
    f2.C = dx.p;
    f2.M = M;
    f2.N = d;
    f2.K = ffn;
    f2.lda = ffn;
    f2.ldw = ffn;
    f2.ldc = d;
    f2.bias = bb2.as<float>();
    f2.residual = dx.as<float>();
    f2.ldr = d;
    bool fused_next = false;
    if (want_next && gemm_nt_fuses_layernorm_out(dtype == WM_F32 ? 4 : 2, f2)) {
        f2.lno_g = g2.as<float>();
        f2.lno_b = b2.as<float>();
        f2.lno_out = xn.p;
        fused_next = true;
    }
    WMCHK(gemm_dispatch(dtype, WM_F32, f2, 1, st));
    if (want_next && !fused_next)
        DISPATCH_DT(dtype, TT, launch_layernorm_rows<TT>(dx.as<float>(), g2.as<float>(), b2.as<float>(), xn.p, nullptr, M, d, 1e-5f, st));
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(x, dx.p, (size_t)M * d * 4, hipMemcpyDeviceToHost));
    if (want_next) {  // the operand rows as the next GEMM reads them, widened on the host
        const size_t n = (size_t)M * d;
        std::vector<unsigned char> h(n * ts);
        HIPCHK(hipMemcpy(h.data(), xn.p, h.size(), hipMemcpyDeviceToHost));
        widen_to_f32(h.data(), dtype, n, xn_out);
    }
    return 0;
}

// widen n operand-dtype values on the host (known-answer entry points only)
static void widen_to_f32(const void* src, int dtype, size_t n, float* dst) {
    if (dtype == WM_F32) {
        memcpy(dst, src, n * 4);
        return;
    }
    const uint16_t* h = static_cast<const uint16_t*>(src);
    for (size_t i = 0; i < n; ++i) {
        if (dtype == WM_BF16) {
            const uint32_t u = (uint32_t)h[i] << 16;
            memcpy(&dst[i], &u, 4);
        } else {
            _Float16 hv;
            memcpy(&hv, &h[i], 2);
            dst[i] = (float)hv;
        }
    }
}

extern "C" int wm_op_attention(float* out, const float* q, const float* k, const float* v, int n_ctx, int n_heads, int dtype) {
    if (!out || !q || !k || !v || n_ctx <= 0 || n_heads <= 0 || n_heads > 64) return fail(WM_E_ARG, "bad argument");
    if (dtype < 0 || dtype > 2) return fail(WM_E_ARG, "bad dtype");
    const size_t d = (size_t)n_heads * 64, n = (size_t)n_ctx * d;
    std::vector<float> packed(3 * n);  // the fused projection layout the kernel reads: row = [q | k | v]
    for (int t = 0; t < n_ctx; ++t) {
        memcpy(&packed[(size_t)t * 3 * d], q + (size_t)t * d, d * 4);
        memcpy(&packed[(size_t)t * 3 * d + d], k + (size_t)t * d, d * 4);
        memcpy(&packed[(size_t)t * 3 * d + 2 * d], v + (size_t)t * d, d * 4);
    }
    TmpDev t;
    t.bufs.reserve(4);
    DevBuf &qkv = t.add(), &o = t.add();
    WMCHK(upload(qkv, packed.data(), packed.size(), dtype));
    WMCHK(o.alloc(n * dt_size(dtype), true));
    DISPATCH_DT(dtype, TT, launch_flash_attn_enc<TT>(qkv.p, o.p, 1, n_heads, n_ctx, 0.125f, nullptr));
    HIPCHK(hipGetLastError());
    std::vector<unsigned char> h(n * dt_size(dtype));
    HIPCHK(hipMemcpy(h.data(), o.p, h.size(), hipMemcpyDeviceToHost));
    widen_to_f32(h.data(), dtype, n, out);
    return 0;
}

extern "C" int wm_op_attention_cached(float* out, const float* q, const float* k, const float* v, int B, int t, int n_heads, int kv_dtype,
                                      int n_chunks) {
    if (!out || !q || !k || !v || B <= 0 || t <= 0 || n_heads <= 0 || n_heads > 16) return fail(WM_E_ARG, "bad argument");
    if (kv_dtype < 0 || kv_dtype > 2) return fail(WM_E_ARG, "bad dtype");
    if (n_chunks < 1 || (n_chunks > 1 && (n_chunks < (t + 511) / 512 || n_chunks > (t + 31) / 32)))
        return fail(WM_E_ARG, "n_chunks must be 1, or between ceil(t/512) and ceil(t/32)");
    if (n_chunks == 1 && t > 512) return fail(WM_E_ARG, "the single-workgroup form serves up to 512 keys (the text context)");
    const size_t d = (size_t)n_heads * 64;
    TmpDev tmp;
    tmp.bufs.reserve(8);
    DevBuf &dq = tmp.add(), &dk = tmp.add(), &dv = tmp.add(), &po = tmp.add(), &pml = tmp.add(), &o = tmp.add(), &ctl = tmp.add();
    WMCHK(upload(dq, q, (size_t)B * d, WM_F32));
    WMCHK(upload(dk, k, (size_t)B * t * d, kv_dtype));
    WMCHK(upload(dv, v, (size_t)B * t * d, kv_dtype));
    WMCHK(o.alloc((size_t)B * d * 4, true));
    AttnDecParams a{};
    a.q = dq.as<float>();
    a.K = dk.p;
    a.V = dv.p;
    a.batch_stride = (long)((size_t)t * d);
    a.scale = 0.125f;
    a.H = n_heads;
    a.d = (int)d;
    a.B = B;
    if (n_chunks > 1) {
        WMCHK(po.alloc((size_t)B * n_chunks * d * 4, true));
        WMCHK(pml.alloc((size_t)B * n_chunks * n_heads * 2 * 4, true));
        a.n_keys = t;
        a.nsplit = n_chunks;
        a.part_o = po.as<float>();
        a.part_ml = pml.as<float>();
        WMCHK(attn_decode_dispatch(kv_dtype, a, nullptr));
        launch_attn_combine(po.as<float>(), pml.as<float>(), o.p, WM_F32, B, n_chunks, n_heads, (int)d, nullptr);
    } else {
        StepCtl h{};
        h.len = t - 1;  // the kernel sweeps len + 1 rows: the cache as it stands after this step's row was appended
        WMCHK(ctl.alloc(sizeof(StepCtl)));
        HIPCHK(hipMemcpy(ctl.p, &h, sizeof h, hipMemcpyHostToDevice));
        a.n_keys = -1;
        a.ctl = ctl.as<StepCtl>();
        a.nsplit = 1;
        a.direct_out = o.as<float>();
        a.out_dtype = WM_F32;
        WMCHK(attn_decode_dispatch(kv_dtype, a, nullptr));
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(out, o.p, (size_t)B * d * 4, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int wm_op_gelu(float* tt, size_t n, int mode) {
    if (!tt || (mode != 0 && mode != 1)) return fail(WM_E_ARG, "bad argument");
    if (n == 0) return 0;
    TmpDev t;
    t.bufs.reserve(2);
    DevBuf& x = t.add();
    WMCHK(upload(x, tt, n, WM_F32));
    launch_gelu(x.as<float>(), n, mode, nullptr);
    HIPCHK(hipMemcpy(tt, x.p, n * 4, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int wm_op_softmax_rows(float* tt, int rows, int cols) {
    if (!tt || rows <= 0 || cols <= 0) return fail(WM_E_ARG, "bad argument");
    TmpDev t;
    t.bufs.reserve(2);
    DevBuf& x = t.add();
    WMCHK(upload(x, tt, (size_t)rows * cols, WM_F32));
    launch_softmax_rows(x.as<float>(), rows, cols, nullptr);
    HIPCHK(hipMemcpy(tt, x.p, (size_t)rows * cols * 4, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int wm_op_argmax(const float* tt, int n, int32_t* idx) {
    if (!tt || !idx || n <= 0) return fail(WM_E_ARG, "bad argument");
    TmpDev t;
    t.bufs.reserve(2);
    DevBuf &x = t.add(), &o = t.add();
    WMCHK(upload(x, tt, n, WM_F32));
    WMCHK(o.alloc(4));
    launch_argmax_plain(x.as<float>(), n, o.as<int>(), nullptr);
    HIPCHK(hipMemcpy(idx, o.p, 4, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int wm_op_conv1d_k3(float* out, const float* inp, const float* weight, const float* bias, int C_in, int L_in,
                               int C_out, int stride, int out_T, int dtype) {
    if (!out || !inp || !weight || !bias || C_in <= 0 || L_in <= 0 || C_out <= 0) return fail(WM_E_ARG, "bad argument");
    if (stride != 1 && stride != 2) return fail(WM_E_ARG, "stride must be 1 or 2");
    if (C_out % 128) return fail(WM_E_ARG, "C_out must be a multiple of 128");
    if (dtype < 0 || dtype > 2) return fail(WM_E_ARG, "bad dtype");
    const int Cp = (C_in + 31) / 32 * 32;
    const int L_out = (L_in + 2 - 3) / stride + 1;
    TmpDev t;
    t.bufs.reserve(8);
    DevBuf &x = t.add(), &xt = t.add(), &w = t.add(), &b = t.add(), &o = t.add(), &o2 = t.add();
    WMCHK(upload(x, inp, (size_t)C_in * L_in, WM_F32));
    WMCHK(xt.alloc(((size_t)L_in + 2 + 512) * Cp * dt_size(dtype), true));
    auto wr = conv_relayout(weight, C_out, C_in, Cp);
    WMCHK(upload(w, wr.data(), wr.size(), dtype));
    WMCHK(upload(b, bias, C_out, WM_F32));
    WMCHK(o.alloc((size_t)L_out * C_out * 4));
    DISPATCH_DT(dtype, TT, launch_mel_transpose_pad<TT>(x.as<float>(), xt.p, 1, C_in, L_in, Cp, nullptr));
    GemmParams p{};
    p.A = xt.p;
    p.W = w.p;
    p.C = o.p;
    p.M = L_out;
    p.N = C_out;
    p.K = 3 * Cp;
    p.lda = (long)stride * Cp;
    p.ldw = 3 * Cp;
    p.ldc = C_out;
    p.bias = b.as<float>();
    WMCHK(gemm_dispatch(dtype, WM_F32, p, 1, nullptr));
    if (out_T) {
        HIPCHK(hipMemcpy(out, o.p, (size_t)L_out * C_out * 4, hipMemcpyDeviceToHost));
    } else {
        WMCHK(o2.alloc((size_t)L_out * C_out * 4));
        launch_transpose_f32(o.as<float>(), o2.as<float>(), L_out, C_out, nullptr);
        HIPCHK(hipMemcpy(out, o2.p, (size_t)L_out * C_out * 4, hipMemcpyDeviceToHost));
    }
    HIPCHK(hipGetLastError());
    return 0;
}
