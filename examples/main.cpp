// main.cpp — main.mojo:11-45 written against include/whisper_mi.hpp (the C++ host mirror over the C-ABI):
// load the flat fp32 weights, load one 80x3000 log-mel, time transcribe, print the ids, render them with vocab.txt.
//
//   whisper_main [--config tiny|base|micro] [--weights FILE | --synthetic-weights SEED] [--mel FILE | --synthetic-mel SEED]
//                [--vocab vocab.txt] [--dtype f32|bf16|f16] [--max-loop N] [--ignore-eot] [--prompt a,b,c,d] [--eot ID]
//                [--pipelined N]  after the reference's flow: the same clip N times through transcribe_submit / transcribe_wait
//                                 (coalesced in pairs by the library), every result checked against the synchronous call's
//
// Defaults are the reference's file names (whisper_tiny_weights.bin, sample_input.bin, vocab.txt).  The synthetic options
// use the library's own generator (wm_synth_weights / wm_synth_mel_host) so the program runs without the reference's
// (network-hosted) weights; tests compare its ids with the Python host layer on the same seeds.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <iostream>
#include <string>

#include "whisper_mi.hpp"

using namespace whisper_mi;

int main(int argc, char** argv) {
    std::string config = "tiny", weights = "whisper_tiny_weights.bin", mel_path = "sample_input.bin", vocab = "vocab.txt", dtype = "f32";
    long synth_w = -1, synth_m = -1;
    int max_loop = Whisper::MAX_LOOP;
    bool ignore_eot = false;
    int pipelined = 0;
    std::vector<int32_t> prompt;
    int eot = Whisper::EOT;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() -> std::string {
            if (i + 1 >= argc) {
                std::cerr << "missing value after " << a << "\n";
                std::exit(2);
            }
            return argv[++i];
        };
        if (a == "--config") config = next();
        else if (a == "--weights") weights = next();
        else if (a == "--synthetic-weights") synth_w = std::atol(next().c_str());
        else if (a == "--mel") mel_path = next();
        else if (a == "--synthetic-mel") synth_m = std::atol(next().c_str());
        else if (a == "--vocab") vocab = next();
        else if (a == "--dtype") dtype = next();
        else if (a == "--max-loop") max_loop = std::atoi(next().c_str());
        else if (a == "--ignore-eot") ignore_eot = true;
        else if (a == "--pipelined") pipelined = std::atoi(next().c_str());
        else if (a == "--eot") eot = std::atoi(next().c_str());
        else if (a == "--prompt") {
            std::stringstream ss(next());
            for (std::string tok; std::getline(ss, tok, ',');) prompt.push_back(std::atoi(tok.c_str()));
        }
        else {
            std::cerr << "unknown option " << a << "\n";
            return 2;
        }
    }
    try {
        const WhisperConfig cfg = config == "base" ? WhisperConfig::base() : config == "micro" ? WhisperConfig::micro() : WhisperConfig::tiny();
        const int dt = dtype == "bf16" ? WM_BF16 : dtype == "f16" ? WM_F16 : WM_F32;
        std::cout << "Initializing Whisper (" << config << ") on MI355X...\n";
        Whisper whisper(cfg, dt, -1, 1, 0, pipelined > 0 ? 2 : 0);
        if (!prompt.empty() || eot != Whisper::EOT) whisper.set_prompt(prompt.empty() ? std::vector<int32_t>(Whisper::PROMPT, Whisper::PROMPT + 4) : prompt, eot);
        if (synth_w >= 0) {
            std::cout << "Generating synthetic weights (seed " << synth_w << ")...\n";
            std::vector<float> w(cfg.weight_count());
            const wm_dims d = cfg.dims();
            wm_synth_weights(&d, (uint64_t)synth_w, w.data());
            whisper.load(w.data(), w.size());
        } else {
            std::cout << "Loading weights from " << weights << "...\n";
            WeightLoader loader(weights);  // raises if missing (loader.mojo:10-11)
            whisper.load(loader);
        }
        Tensor mel(cfg.n_mels, cfg.n_frames());
        if (synth_m >= 0) {
            wm_synth_mel_host((uint64_t)synth_m, cfg.n_mels, cfg.n_frames(), mel.ptr());
        } else {
            std::cout << "Loading sample input from " << mel_path << "...\n";
            std::ifstream f(mel_path, std::ios::binary);
            if (!f) throw std::runtime_error("cannot open " + mel_path);
            f.read(reinterpret_cast<char*>(mel.ptr()), (std::streamsize)(mel.size() * sizeof(float)));  // main.mojo:23-27
            if ((size_t)f.gcount() != mel.size() * sizeof(float)) throw std::runtime_error(mel_path + ": short read");
        }
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<int> tokens = ignore_eot ? whisper.transcribe_batch(mel.ptr(), 1, max_loop, true)[0] : whisper.transcribe(mel, max_loop);
        const auto t1 = std::chrono::steady_clock::now();
        std::cout << "Transcription time:  " << std::chrono::duration<double>(t1 - t0).count() << " seconds\n\nToken IDs:\n";
        for (int t : tokens) std::cout << t << " ";
        std::cout << "\n";
        std::ifstream vf(vocab);
        if (vf) {
            Tokenizer tokenizer(vocab);
            std::cout << "\n========================================\nFINAL TRANSCRIPTION:\n========================================\n"
                      << tokenizer.decode(tokens) << "\n========================================\n";
        } else {
            std::cout << "\n(no " << vocab << ": ids only)\n";
        }
        if (pipelined > 0) {  // back-to-back calls: submit up to eight, collect them, compare with the list above
            int bad = 0;
            for (int k0 = 0; k0 < pipelined; k0 += 8) {
                const int n = std::min(8, pipelined - k0);
                for (int sl = 0; sl < n; ++sl) whisper.transcribe_submit(mel.ptr(), 1, sl, max_loop, ignore_eot);
                for (int sl = 0; sl < n; ++sl)
                    if (whisper.transcribe_wait(sl)[0] != tokens) ++bad;
            }
            std::cout << "\nPipelined: " << pipelined << " submits, " << bad << " differ from the synchronous result\n";
            if (bad) return 1;
        }
        std::cout << "\nDone." << std::endl;  // flushed here: everything main printed is out before any exit-time teardown
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
