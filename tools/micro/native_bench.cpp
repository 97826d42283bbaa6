// native_bench.cpp — drives libwhispermi.so from plain C++ (no Python / torch in the process) to separate library
// performance from host-environment effects.  g++ native_bench.cpp -I../../include -L... -lwhispermi
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "whisper_mi.h"
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    int B = argc > 1 ? atoi(argv[1]) : 64;
    int dt = argc > 2 ? atoi(argv[2]) : 1;
    wm_config cfg{{384, 6, 4, 1536, 80, 1500, 448, 51865}, 0, dt, dt, B};
    std::vector<float> w(wm_weight_count(&cfg.dims));
    wm_synth_weights(&cfg.dims, 0, w.data());
    wm_model* m = nullptr;
    if (wm_model_load_memory(w.data(), w.size(), &cfg, 0, &m)) { printf("load: %s\n", wm_last_error()); return 1; }
    std::vector<float> mel((size_t)B * 80 * 3000);
    for (int i = 0; i < B; ++i) wm_synth_mel_host(1000 + i, 80, 3000, mel.data() + (size_t)i * 240000);
    int32_t prompt[4] = {50258, 50259, 50359, 50363};
    wm_decode_opts o{prompt, 4, 50257, 99, 0, 1};
    std::vector<int32_t> toks((size_t)B * 104), n(B);
    for (int r = 0; r < 4; ++r) {
        double t0 = now();
        if (wm_transcribe(m, mel.data(), 0, B, &o, toks.data(), n.data())) { printf("transcribe: %s\n", wm_last_error()); return 1; }
        printf("B=%d dtype=%d pass %d: %.2f ms (host mel upload included)\n", B, dt, r, (now() - t0) * 1e3);
    }
    wm_model_free(m);
    return 0;
}
