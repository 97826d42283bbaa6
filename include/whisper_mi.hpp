// whisper_mi.hpp — C++ host mirror of the reference's Mojo interface, header-only, over the C-ABI of whisper_mi.h.
//
// The reference (antonvice/whisper.Mojo) is compiled Mojo; its toolchain is not in this image, so the compiled-language
// host layer is written in C++ with the reference's names, argument meaning and error behaviour:
//   WhisperConfig            whisper.mojo:9-37     (tiny() defaults)
//   Tensor                   whisper_tensor.mojo:14-60   (rows x cols fp32, row-major, owning)
//   WeightLoader(filename)   loader.mojo:5-31      (raises when the file cannot be opened)
//   Whisper / load / transcribe   whisper.mojo:169-223   (prompt 50258 50259 50359 50363, eot 50257, 195-step bound)
//   Tokenizer(path) / decode tokenizer.mojo:4-28   (bug-compatible rendering)
// Everything computes through libwhispermi.so; there is no CPU path here.  Errors become std::runtime_error carrying
// wm_last_error().  examples/main.cpp is main.mojo:11-45 written against this header.
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "whisper_mi.h"

namespace whisper_mi {

inline void check(int rc) {
    if (rc != 0) throw std::runtime_error(std::string("whisper_mi: ") + wm_last_error());
}
// the structs this header fills are WM_ABI_VERSION's: a library of another version would read tails this host never wrote
inline void check_abi() {
    if (wm_abi_version() != WM_ABI_VERSION)
        throw std::runtime_error("whisper_mi: libwhispermi.so speaks ABI version " + std::to_string(wm_abi_version()) + ", this header " +
                                 std::to_string(WM_ABI_VERSION));
}

// whisper.mojo:9-37
struct WhisperConfig {
    int d_model = 384, n_heads = 6, n_layers = 4, ffn = 1536, n_mels = 80, n_audio_ctx = 1500, n_text_ctx = 448,
        vocab_size = 51865;
    static WhisperConfig tiny() { return WhisperConfig{}; }
    static WhisperConfig base() { return WhisperConfig{512, 8, 6, 2048, 80, 1500, 448, 51865}; }
    static WhisperConfig micro() { return WhisperConfig{128, 2, 2, 512, 16, 100, 64, 1000}; }  // test-size model
    wm_dims dims() const { return wm_dims{d_model, n_heads, n_layers, ffn, n_mels, n_audio_ctx, n_text_ctx, vocab_size}; }
    int n_frames() const { return 2 * n_audio_ctx; }
    size_t weight_count() const {
        const wm_dims d = dims();
        return wm_weight_count(&d);
    }
};

// whisper_tensor.mojo:14-60 — just the owning rows x cols buffer the call surface passes around
struct Tensor {
    int rows = 0, cols = 0;
    std::vector<float> data;
    Tensor() = default;
    Tensor(int r, int c) : rows(r), cols(c), data((size_t)r * c, 0.f) {}
    float* ptr() { return data.data(); }
    const float* ptr() const { return data.data(); }
    size_t size() const { return data.size(); }
};

// loader.mojo:5-31: the constructor raises if the file cannot be opened; the library validates the size against the
// config at load (the reference silently reads past the end, loader.mojo:21-27)
class WeightLoader {
public:
    explicit WeightLoader(const std::string& filename) : filename_(filename) {
        std::ifstream f(filename, std::ios::binary);
        if (!f) throw std::runtime_error("WeightLoader: cannot open " + filename);
    }
    const std::string& filename() const { return filename_; }

private:
    std::string filename_;
};

// whisper.mojo:169-223
class Whisper {
public:
    static constexpr int32_t PROMPT[4] = {50258, 50259, 50359, 50363};  // whisper.mojo:187-191
    static constexpr int32_t EOT = 50257;                                // whisper.mojo:206
    static constexpr int MAX_LOOP = 195;                                 // whisper.mojo:205

    explicit Whisper(const WhisperConfig& cfg = WhisperConfig::tiny(), int compute_dtype = WM_F32, int kv_dtype = -1,
                     int max_batch = 1, int device = 0, int coalesce = 0)
        : cfg_(cfg), device_(device) {
        wcfg_.dims = cfg.dims();
        wcfg_.gelu_mode = WM_GELU_TANH;  // whisper_tensor.mojo:288-308
        wcfg_.compute_dtype = compute_dtype;
        wcfg_.kv_dtype = kv_dtype < 0 ? compute_dtype : kv_dtype;
        wcfg_.max_batch = max_batch;
        wcfg_.coalesce = coalesce;  // 2: consecutive transcribe_submit calls share one 2·B-row decode state (wm_config.coalesce)
    }
    Whisper(const Whisper&) = delete;
    Whisper& operator=(const Whisper&) = delete;
    ~Whisper() {
        if (model_) wm_model_free(model_);
    }

    // whisper.load(loader)  (whisper.mojo:180-182, main.mojo:16-17)
    void load(const WeightLoader& loader) {
        check_abi();
        if (model_) wm_model_free(model_), model_ = nullptr;
        check(wm_model_load(loader.filename().c_str(), &wcfg_, device_, &model_));
    }
    // weights already in memory (flat fp32, the file's order)
    void load(const float* weights, size_t n_floats) {
        check_abi();
        if (model_) wm_model_free(model_), model_ = nullptr;
        check(wm_model_load_memory(weights, n_floats, &wcfg_, device_, &model_));
    }

    // whisper.transcribe(mel) -> List[Int]  (whisper.mojo:184-223): prompt + generated ids (+ eot when hit)
    std::vector<int> transcribe(const Tensor& mel, int max_loop = MAX_LOOP) const {
        if (mel.rows != cfg_.n_mels || mel.cols != cfg_.n_frames()) throw std::runtime_error("transcribe: mel must be n_mels x 2*n_audio_ctx");
        return transcribe_batch(mel.ptr(), 1, max_loop)[0];
    }
    // B utterances, host mels [B][n_mels][n_frames]
    std::vector<std::vector<int>> transcribe_batch(const float* mels, int B, int max_loop = MAX_LOOP, bool ignore_eot = false) const {
        need_model();
        wm_decode_opts o = opts(max_loop, ignore_eot);
        const int stride = o.n_prompt + 1 + max_loop;
        std::vector<int32_t> toks((size_t)B * stride), n(B);
        check(wm_transcribe(model_, mels, 0, B, &o, toks.data(), n.data()));
        return unpack(toks, n, B, stride);
    }
    // pipelined form: submit on slot 0..7, wait later (four in flight is the optimum; see whisper_mi.h)
    void transcribe_submit(const float* mels, int B, int slot, int max_loop = MAX_LOOP, bool ignore_eot = false) {
        need_model();
        wm_decode_opts o = opts(max_loop, ignore_eot);
        check(wm_transcribe_submit(model_, slot, mels, 0, B, &o));
        pend_[slot] = {B, o.n_prompt + 1 + max_loop};
    }
    std::vector<std::vector<int>> transcribe_wait(int slot) {
        need_model();
        const auto [B, stride] = pend_[slot];
        std::vector<int32_t> toks((size_t)B * stride), n(B);
        check(wm_transcribe_wait(model_, slot, toks.data(), n.data()));
        return unpack(toks, n, B, stride);
    }
    // other prompts / stop ids (reduced test models have small vocabularies); defaults are the reference's
    void set_prompt(const std::vector<int32_t>& prompt, int32_t eot) {
        prompt_ = prompt;
        eot_ = eot;
    }
    // timestamp rules of HF generate (the reference has none): timestamp_begin <= 0 switches them off
    void set_timestamps(int timestamp_begin, int no_timestamps_token, int max_initial_timestamp_index) {
        ts_begin_ = timestamp_begin;
        no_ts_ = no_timestamps_token;
        max_init_ = max_initial_timestamp_index;
    }
    const WhisperConfig& config() const { return cfg_; }
    wm_model* handle() const { return model_; }

private:
    void need_model() const {
        if (!model_) throw std::runtime_error("Whisper: load() first");
    }
    wm_decode_opts opts(int max_loop, bool ignore_eot) const {
        wm_decode_opts o;
        std::memset(&o, 0, sizeof o);
        o.prompt = prompt_.data();
        o.n_prompt = (int)prompt_.size();
        o.eot = eot_;
        o.max_loop = max_loop;
        o.pos_mode = WM_POS_REF;  // start_pos = current_len - 1 (whisper.mojo:217)
        o.ignore_eot = ignore_eot ? 1 : 0;
        o.timestamp_begin = ts_begin_;
        o.no_timestamps_token = no_ts_;
        o.max_initial_timestamp_index = max_init_;
        return o;
    }
    static std::vector<std::vector<int>> unpack(const std::vector<int32_t>& toks, const std::vector<int32_t>& n, int B, int stride) {
        std::vector<std::vector<int>> out(B);
        for (int b = 0; b < B; ++b) out[b].assign(toks.begin() + (size_t)b * stride, toks.begin() + (size_t)b * stride + n[b]);
        return out;
    }
    WhisperConfig cfg_;
    std::vector<int32_t> prompt_{PROMPT, PROMPT + 4};
    int32_t eot_ = EOT;
    int ts_begin_ = 0, no_ts_ = -1, max_init_ = -1;
    wm_config wcfg_{};
    int device_ = 0;
    wm_model* model_ = nullptr;
    struct Pend {
        int B, stride;
    };
    Pend pend_[8] = {};
};

// tokenizer.mojo:4-28: vocab.txt, line index = id; decode drops <|...|>, maps "Ġ" to a space and the escaped "\n" to a
// newline — the reference's rendering, bug for bug (non-ASCII text stays in its byte-level symbols)
class Tokenizer {
public:
    explicit Tokenizer(const std::string& path) {
        std::ifstream f(path, std::ios::binary);
        if (!f) throw std::runtime_error("Tokenizer: cannot open " + path);  // tokenizer.mojo:9 raises
        std::stringstream ss;
        ss << f.rdbuf();
        const std::string content = ss.str();
        size_t a = 0;
        for (;;) {  // content.split("\n"): a trailing newline yields a last empty entry, as in the reference
            const size_t b = content.find('\n', a);
            if (b == std::string::npos) {
                vocab_.push_back(content.substr(a));
                break;
            }
            vocab_.push_back(content.substr(a, b - a));
            a = b + 1;
        }
    }
    std::string decode(const std::vector<int>& tokens) const {
        static const std::string G = "\xC4\xA0";  // "Ġ" (U+0120) in UTF-8
        std::string result;
        for (int id : tokens) {
            if (id < 0 || id >= (int)vocab_.size()) continue;  // tokenizer.mojo:19
            const std::string& t = vocab_[id];
            if (t.size() >= 4 && t.compare(0, 2, "<|") == 0 && t.compare(t.size() - 2, 2, "|>") == 0) continue;
            std::string c = t;
            replace_all(c, G, " ");
            replace_all(c, "\\n", "\n");
            result += c;
        }
        return result;
    }
    size_t size() const { return vocab_.size(); }

private:
    static void replace_all(std::string& s, const std::string& from, const std::string& to) {
        for (size_t p = 0; (p = s.find(from, p)) != std::string::npos; p += to.size()) s.replace(p, from.size(), to);
    }
    std::vector<std::string> vocab_;
};

}  // namespace whisper_mi
