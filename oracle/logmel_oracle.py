"""CPU restatement of the log-mel front end (TEST INFRASTRUCTURE ONLY — see oracle/whisper_oracle.c for the rules).

SURVEY §8(f) rank 1.  The reference has no front end of its own: export_weights.py:100-116 calls
`WhisperProcessor(audio, sampling_rate=16000)` from the third-party `transformers` (4.57.3 pinned, uv.lock:2525-2526),
which is absent from /root/reference.  This restates its published algorithm
(transformers/models/whisper/feature_extraction_whisper.py `_np_extract_fbank_features` +
transformers/audio_utils.py `spectrogram`, `mel_filter_bank`, `window_function`):

  pad / trim to 30 s (480 000 samples) with zeros -> reflect-pad 200 each side -> frames of 400, hop 160 ->
  periodic Hann -> |rfft|^2 (float64) -> 80 slaney-normalised slaney-scale mel filters -> max(., 1e-10) -> log10 ->
  drop the last frame (3000 left) -> max(x, x.max() - 8) -> (x + 4) / 4 -> float32 [80, 3000].

PARITY PIN: tests/golden/logmel.npz was generated in the dev container by the locally importable transformers
WhisperFeatureExtractor (tools/make_golden_mel.py) on seed-reproducible synthetic audio (synth_audio below)."""
from __future__ import annotations

import numpy as np

SAMPLE_RATE, N_FFT, HOP = 16000, 400, 160


def hertz_to_mel_slaney(f):
    f = np.asarray(f, np.float64)
    mels = 3.0 * f / 200.0
    logstep = 27.0 / np.log(6.4)
    return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-30) / 1000.0) * logstep, mels)


def mel_to_hertz_slaney(m):
    m = np.asarray(m, np.float64)
    f = 200.0 * m / 3.0
    logstep = np.log(6.4) / 27.0
    return np.where(m >= 15.0, 1000.0 * np.exp(logstep * (m - 15.0)), f)


def mel_filter_bank(n_freq=201, n_mels=80, fmin=0.0, fmax=8000.0, sr=SAMPLE_RATE) -> np.ndarray:
    """[n_freq, n_mels] float64, norm='slaney', mel_scale='slaney' (audio_utils.mel_filter_bank)."""
    mel_freqs = np.linspace(hertz_to_mel_slaney(fmin), hertz_to_mel_slaney(fmax), n_mels + 2)
    filter_freqs = mel_to_hertz_slaney(mel_freqs)
    fft_freqs = np.linspace(0, sr // 2, n_freq)
    filter_diff = np.diff(filter_freqs)
    slopes = filter_freqs[None, :] - fft_freqs[:, None]
    down = -slopes[:, :-2] / filter_diff[:-1]
    up = slopes[:, 2:] / filter_diff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))
    fb *= (2.0 / (filter_freqs[2:n_mels + 2] - filter_freqs[:n_mels]))[None, :]
    return fb


def log_mel(audio: np.ndarray, n_frames: int = 3000, n_mels: int = 80) -> np.ndarray:
    """audio: 1-D float at 16 kHz, any length -> [n_mels, n_frames] float32."""
    n_samples = HOP * n_frames
    x = np.zeros(n_samples, np.float64)
    a = np.asarray(audio, np.float64)[:n_samples]
    x[:len(a)] = a
    xp = np.pad(x, (N_FFT // 2, N_FFT // 2), mode="reflect")
    window = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(N_FFT) / N_FFT)  # periodic Hann
    idx = np.arange(N_FFT)[None, :] + HOP * np.arange(n_frames + 1)[:, None]
    spec = np.abs(np.fft.rfft(xp[idx] * window[None, :], axis=1)) ** 2  # [frames+1, 201]
    mel = np.maximum(1e-10, mel_filter_bank(n_mels=n_mels).T @ spec.T)  # [n_mels, frames+1]
    log_spec = np.log10(mel)[:, :-1]
    log_spec = np.maximum(log_spec, log_spec.max() - 8.0)
    return ((log_spec + 4.0) / 4.0).astype(np.float32)


# ---- seed-reproducible synthetic audio (integer recipe + single fp32 operations: identical in numpy and C) --------
def _ih4(seed: int, tensor: int, count: int) -> np.ndarray:
    M = (1 << 64) - 1

    def mix(z):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))

    with np.errstate(over="ignore"):
        h = mix(mix(np.uint64((seed * 0x100000001B3 + tensor) & M)) + np.arange(count, dtype=np.uint64))
    m = np.uint64(0xFFFF)
    s = (h & m) + ((h >> np.uint64(16)) & m) + ((h >> np.uint64(32)) & m) + ((h >> np.uint64(48)) & m)
    return s.astype(np.int64) - 131070


def synth_audio(seed: int, n: int) -> np.ndarray:
    """n samples of float32 'speech-like' test audio: noise bursts under a piecewise-constant envelope with silences,
    a 2-tap low-pass tilt, and a 400 Hz square tone — every value is (int -> fp32) * fp32 plus fp32 adds."""
    v = _ih4(seed, 0x415544, n + 1).astype(np.float32) * np.float32(0.1 * 2.6428996e-05)
    t = np.arange(n)
    seg = (t // 4000) % 8  # 0.25 s segments: loud, soft, silence ...
    env = np.array([1.0, 0.25, 0.0, 0.5, 0.05, 1.0, 0.0, 0.125], np.float32)[seg]
    noise = (v[1:] + v[:-1]) * env
    tone = np.where((t // 20) % 2 == 0, np.float32(0.02), np.float32(-0.02)) * (seg % 2 == 0).astype(np.float32)
    return (noise + tone).astype(np.float32)
