#!/usr/bin/env python3
"""Generates tests/golden/logmel.npz: outputs of the locally importable transformers WhisperFeatureExtractor (the library
the reference's export_weights.py:100-116 delegates to) on seed-reproducible synthetic audio.  Dev container only."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import logmel_oracle as lo  # noqa: E402


def main():
    from transformers import WhisperFeatureExtractor
    fe = WhisperFeatureExtractor()
    assert (fe.n_fft, fe.hop_length, fe.n_samples, fe.nb_max_frames) == (400, 160, 480000, 3000)
    out = {"mel_filters_colsum": fe.mel_filters.astype(np.float64).sum(0), "mel_filters_sample": fe.mel_filters[::10, ::8].copy()}
    cols = np.r_[0:40, 1480:1520, 2960:3000]
    out["cols"] = cols.astype(np.int32)
    for i, (seed, n) in enumerate([(1, 480000), (2, 163840), (3, 600000), (4, 1000)]):  # full, short, over-long, tiny
        audio = lo.synth_audio(seed, n)
        feats = fe(audio, sampling_rate=16000, return_tensors="np").input_features[0]
        assert feats.shape == (80, 3000) and feats.dtype == np.float32
        out[f"seed{i}"] = np.int64(seed)
        out[f"n{i}"] = np.int64(n)
        out[f"mel{i}_cols"] = feats[:, cols].copy()
        out[f"mel{i}_rowsum"] = feats.astype(np.float64).sum(1)
        out[f"mel{i}_colsum"] = feats.astype(np.float64).sum(0)
        ours = lo.log_mel(audio)
        print(f"utt {i}: n={n} oracle-vs-HF max abs err {np.abs(ours - feats).max():.3e}; range [{feats.min():.3f}, {feats.max():.3f}]")
    path = os.path.join(ROOT, "tests", "golden", "logmel.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
