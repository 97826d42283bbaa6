#!/usr/bin/env python3
"""Generates tests/golden/micro_timestamps.npz (timestamp rules): transformers' own
WhisperTimeStampLogitsProcessor applied (a) to random score rows under hand-picked histories (a known-answer table for the
rule logic) and (b) to every step of a greedy run of the HF Whisper architecture on the synthetic micro weights (REF-mode
semantics otherwise, as tools/make_golden.py), with and without the suppress processors in front.  Dev container only."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from whisper_mojo_amd import WhisperConfig, synth  # noqa: E402
import make_golden as mg  # noqa: E402
from make_golden_suppress import greedy  # noqa: E402

EOS, NO_TS, TB, MAX_INIT = 900, 940, 941, 5  # micro vocabulary (1000 ids): text < 900, specials 900..940, 59 timestamps


def main():
    from transformers.generation.logits_process import (SuppressTokensAtBeginLogitsProcessor, SuppressTokensLogitsProcessor,
                                                        WhisperTimeStampLogitsProcessor)
    torch.manual_seed(0)
    cfg = WhisperConfig.micro()
    V = cfg.vocab_size
    gc = SimpleNamespace(no_timestamps_token_id=NO_TS, eos_token_id=EOS, bos_token_id=EOS, max_initial_timestamp_index=MAX_INIT)
    # (a) rule table: histories (generated ids after a 4-token prompt) x random score rows
    rng = np.random.default_rng(7)
    histories = [[], [TB + 2], [TB + 2, 17], [TB + 2, 17, 300], [TB + 2, 17, TB + 9], [TB + 2, 17, TB + 9, TB + 9],
                 [TB + 2, 17, TB + 9, TB + 9, 5], [TB, TB], [TB + 57], [TB + 1, 3, TB + 58], [5, 6], [TB + 3, 8, 9, EOS]]
    rows = (rng.standard_normal((len(histories), 3, V)) * np.array([1.0, 1.0, 4.0])[None, :, None]).astype(np.float32)
    rows[:, 1, TB:] += 3.0   # timestamp-heavy rows: rule 5 fires
    rows[:, 2, :TB] += 2.0   # text-heavy rows
    proc = WhisperTimeStampLogitsProcessor(gc, begin_index=4)
    out = np.empty_like(rows)
    for h, hist in enumerate(histories):
        ids = torch.tensor([[1, 2, 3, 4] + hist] * 3)
        out[h] = proc(ids, torch.from_numpy(rows[h])).numpy()
    hist_pad = np.full((len(histories), 8), -1, np.int32)
    for h, hist in enumerate(histories):
        hist_pad[h, :len(hist)] = hist
    # (b) greedy streams of the micro model
    w = synth.split_weights(cfg, synth.synth_weights(cfg, 0))
    m = mg.hf_model(cfg, w, ref_mode=True)
    prompt, steps = [1, 2, 3, 4], 40
    streams, mels = [], []
    for seed in (1000, 1001):
        mel = synth.synth_mel(cfg, seed)
        with mg.TanhStemGelu(True):
            enc_out = m.model.encoder(torch.from_numpy(mel)[None]).last_hidden_state
        plain = greedy(m, enc_out, prompt, steps, [])
        ts = greedy(m, enc_out, prompt, steps, [WhisperTimeStampLogitsProcessor(gc, begin_index=4)])
        sup = sorted(set(int(t) for t in ts[4:] if t < EOS))[:6]
        both = greedy(m, enc_out, prompt, steps, [SuppressTokensLogitsProcessor(sup, device="cpu"),
                                                  SuppressTokensAtBeginLogitsProcessor([int(ts[4])], begin_index=4, device="cpu"),
                                                  WhisperTimeStampLogitsProcessor(gc, begin_index=4)])
        assert ts[4] >= TB and ts[4] <= TB + MAX_INIT and not np.array_equal(plain, ts) and both[4] != ts[4]
        n_ts = int((ts[4:] >= TB).sum())
        print(f"seed {seed}: plain {plain[4:10]} ... with rules {ts[4:16]} ({n_ts} timestamps in {steps + 1} ids)")
        streams.append((plain, ts, np.asarray(sup, np.int32), np.asarray([int(ts[4])], np.int32), both))
        mels.append(seed)
    path = os.path.join(ROOT, "tests", "golden", "micro_timestamps.npz")
    np.savez_compressed(path, eos=EOS, no_timestamps=NO_TS, timestamp_begin=TB, max_initial=MAX_INIT, histories=hist_pad,
                        rows=rows, processed=out, prompt=np.asarray(prompt, np.int32), mel_seeds=np.asarray(mels, np.int32),
                        plain=np.stack([s[0] for s in streams]), with_rules=np.stack([s[1] for s in streams]),
                        suppress=np.stack([s[2] for s in streams]), begin_suppress=np.stack([s[3] for s in streams]),
                        with_all=np.stack([s[4] for s in streams]))
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
