/* wm_synth.h — deterministic synthetic weights / mels in the reference's flat file format.
 *
 * Header-only, plain C99 (also valid C++).  Used by the oracle, the tests, bench.py (through the C-ABI
 * helper wm_synth_*) — NOT by the compute path.  The real whisper_tiny_weights.bin is git-ignored upstream
 * (/root/reference/.gitignore:1) and needs network to create (export_weights.py:13-14), so every
 * offline test/bench runs on weights generated here, bit-identically in C and in numpy
 * (whisper.mojo_amd/synth.py implements the same integer recipe; tests/test_synth.py checks equality).
 *
 * File layout produced = the order export_weights.py:19-90 writes and loader.next_tensor consumes
 * (whisper.mojo:60-69,122-128; layers.mojo:96-103,418-433):
 *   enc conv1.w[d,n_mels,3] conv1.b conv2.w[d,d,3] conv2.b enc_pos[n_audio_ctx,d]
 *   x n_layers enc blocks {q.w q.b k.w v.w v.b o.w o.b ln1.w ln1.b fc1.w fc1.b fc2.w fc2.b ln2.w ln2.b}
 *   enc ln.w ln.b; tok_emb[vocab,d]; dec_pos[n_text_ctx,d]
 *   x n_layers dec blocks {self q.w..o.b ln1 | cross q.w..o.b lnx | fc1 fc2 ln2}; dec ln.w ln.b
 *
 * Values: no libm anywhere (so C and numpy agree to the bit): an Irwin-Hall(4) sum of 16-bit fields of a
 * splitmix64 hash of (seed, tensor index, element index) — an integer in [-131070,131070] with
 * std 37837.23 — converted exactly to fp32 and scaled by ONE fp32 multiply (plus one fp32 add for LN gamma).
 */
#ifndef WM_SYNTH_H
#define WM_SYNTH_H
#include <stddef.h>
#include <stdint.h>

typedef struct {
    int d_model, n_heads, n_layers, ffn, n_mels, n_audio_ctx, n_text_ctx, vocab;
} wm_dims;

enum { WM_K_WEIGHT = 0, WM_K_QK = 1, WM_K_BIAS = 2, WM_K_GAMMA = 3, WM_K_BETA = 4, WM_K_POS = 5, WM_K_EMB = 6 };

static inline uint64_t wm_mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
/* integer "normal": sum of four 16-bit uniforms, centred */
static inline int32_t wm_ih4(uint64_t seed, uint64_t tensor, uint64_t idx) {
    uint64_t h = wm_mix64(wm_mix64(seed * 0x100000001B3ull + tensor) + idx);
    int32_t s = (int32_t)(h & 0xFFFF) + (int32_t)((h >> 16) & 0xFFFF) + (int32_t)((h >> 32) & 0xFFFF) +
                (int32_t)((h >> 48) & 0xFFFF);
    return s - 131070;
}
#define WM_IH4_INV_STD 2.6428996e-05f /* 1/37837.23, as an fp32 literal */

static inline float wm_kind_scale(int kind) {
    switch (kind) {
        case WM_K_WEIGHT: return 0.02f * WM_IH4_INV_STD;
        case WM_K_QK: return 0.04f * WM_IH4_INV_STD;
        case WM_K_BIAS: return 0.02f * WM_IH4_INV_STD;
        case WM_K_GAMMA: return 0.05f * WM_IH4_INV_STD;
        case WM_K_BETA: return 0.05f * WM_IH4_INV_STD;
        case WM_K_POS: return 0.05f * WM_IH4_INV_STD;
        default: return 0.05f * WM_IH4_INV_STD; /* WM_K_EMB */
    }
}
static inline float wm_synth_value(uint64_t seed, uint64_t tensor, uint64_t idx, int kind) {
    float v = (float)wm_ih4(seed, tensor, idx) * wm_kind_scale(kind);
    return kind == WM_K_GAMMA ? 1.0f + v : v;
}

/* Tensor table: calls cb(user, tensor_index, kind, count) in file order; returns the total float count. */
typedef void (*wm_tensor_cb)(void* user, int tensor, int kind, size_t count);

static inline size_t wm_synth_attn_(const wm_dims* c, int* t, size_t n, wm_tensor_cb cb, void* u) {
    size_t d = (size_t)c->d_model;
    const int kinds[7] = {WM_K_QK, WM_K_BIAS, WM_K_QK, WM_K_WEIGHT, WM_K_BIAS, WM_K_WEIGHT, WM_K_BIAS};
    const size_t cnt[7] = {d * d, d, d * d, d * d, d, d * d, d};
    for (int i = 0; i < 7; ++i) {
        if (cb) cb(u, *t, kinds[i], cnt[i]);
        ++*t;
        n += cnt[i];
    }
    return n;
}
static inline size_t wm_synth_ln_(const wm_dims* c, int* t, size_t n, wm_tensor_cb cb, void* u) {
    if (cb) cb(u, *t, WM_K_GAMMA, (size_t)c->d_model);
    ++*t;
    if (cb) cb(u, *t, WM_K_BETA, (size_t)c->d_model);
    ++*t;
    return n + 2 * (size_t)c->d_model;
}
static inline size_t wm_synth_mlp_(const wm_dims* c, int* t, size_t n, wm_tensor_cb cb, void* u) {
    size_t d = (size_t)c->d_model, f = (size_t)c->ffn;
    const int kinds[4] = {WM_K_WEIGHT, WM_K_BIAS, WM_K_WEIGHT, WM_K_BIAS};
    const size_t cnt[4] = {f * d, f, d * f, d};
    for (int i = 0; i < 4; ++i) {
        if (cb) cb(u, *t, kinds[i], cnt[i]);
        ++*t;
        n += cnt[i];
    }
    return n;
}
static inline size_t wm_synth_walk(const wm_dims* c, wm_tensor_cb cb, void* u) {
    size_t d = (size_t)c->d_model, n = 0;
    int t = 0;
#define WM_EMIT(kind, count)            \
    do {                                \
        if (cb) cb(u, t, (kind), (count)); \
        ++t;                            \
        n += (count);                   \
    } while (0)
    WM_EMIT(WM_K_WEIGHT, d * (size_t)c->n_mels * 3);
    WM_EMIT(WM_K_BIAS, d);
    WM_EMIT(WM_K_WEIGHT, d * d * 3);
    WM_EMIT(WM_K_BIAS, d);
    WM_EMIT(WM_K_POS, (size_t)c->n_audio_ctx * d);
    for (int l = 0; l < c->n_layers; ++l) {
        n = wm_synth_attn_(c, &t, n, cb, u);
        n = wm_synth_ln_(c, &t, n, cb, u);
        n = wm_synth_mlp_(c, &t, n, cb, u);
        n = wm_synth_ln_(c, &t, n, cb, u);
    }
    n = wm_synth_ln_(c, &t, n, cb, u);
    WM_EMIT(WM_K_EMB, (size_t)c->vocab * d);
    WM_EMIT(WM_K_POS, (size_t)c->n_text_ctx * d);
    for (int l = 0; l < c->n_layers; ++l) {
        n = wm_synth_attn_(c, &t, n, cb, u);
        n = wm_synth_ln_(c, &t, n, cb, u);
        n = wm_synth_attn_(c, &t, n, cb, u);
        n = wm_synth_ln_(c, &t, n, cb, u);
        n = wm_synth_mlp_(c, &t, n, cb, u);
        n = wm_synth_ln_(c, &t, n, cb, u);
    }
    n = wm_synth_ln_(c, &t, n, cb, u);
#undef WM_EMIT
    return n;
}
static inline size_t wm_synth_count(const wm_dims* c) { return wm_synth_walk(c, 0, 0); }

typedef struct {
    uint64_t seed;
    float* out;
    size_t off;
} wm_synth_fill_ctx_;
static inline void wm_synth_fill_cb_(void* user, int tensor, int kind, size_t count) {
    wm_synth_fill_ctx_* x = (wm_synth_fill_ctx_*)user;
    float* o = x->out + x->off;
    for (size_t i = 0; i < count; ++i) o[i] = wm_synth_value(x->seed, (uint64_t)tensor, (uint64_t)i, kind);
    x->off += count;
}
/* Fills out[wm_synth_count(c)] with the whole weight file image. */
static inline size_t wm_synth_fill(const wm_dims* c, uint64_t seed, float* out) {
    wm_synth_fill_ctx_ x;
    x.seed = seed;
    x.out = out;
    x.off = 0;
    return wm_synth_walk(c, wm_synth_fill_cb_, &x);
}
/* Synthetic log-mel [n_mels, n_frames] row-major: clip(0.5*n, -1, 1.5)  (SURVEY §8d config 3). */
static inline void wm_synth_mel(uint64_t seed, int n_mels, int n_frames, float* out) {
    size_t n = (size_t)n_mels * (size_t)n_frames;
    for (size_t i = 0; i < n; ++i) {
        float v = (float)wm_ih4(seed, 0x4D454Cull, (uint64_t)i) * (0.5f * WM_IH4_INV_STD);
        out[i] = v < -1.0f ? -1.0f : (v > 1.5f ? 1.5f : v);
    }
}
#endif
