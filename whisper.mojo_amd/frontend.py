"""Log-mel front end: the host-side equivalent of the reference's
`processor(audio, sampling_rate=16000, return_tensors="pt").input_features` (export_weights.py:116), computed on the GPU
through the C-ABI (wm_log_mel / wm_transcribe_pcm)."""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

import numpy as np

from . import _lib
from .config import EOT, MAX_LOOP, PROMPT

SAMPLING_RATE = 16000


def _pack(audios: Sequence[np.ndarray]):
    n = np.asarray([len(a) for a in audios], np.int32)
    stride = max(1, int(n.max()))
    buf = np.zeros((len(audios), stride), np.float32)
    for i, a in enumerate(audios):
        buf[i, :len(a)] = np.asarray(a, np.float32)
    return buf, n, stride


def log_mel(model, audios: Sequence[np.ndarray]) -> np.ndarray:
    """audios: list of 1-D float arrays at 16 kHz (any lengths; padded / trimmed to 30 s like WhisperProcessor) ->
    [B, n_mels, 2*n_audio_ctx] float32."""
    buf, n, stride = _pack(audios)
    cfg = model.config
    out = np.empty((len(audios), cfg.n_mels, cfg.n_frames), np.float32)
    fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int32)
    _lib.check(_lib.lib().wm_log_mel(model._h, buf.ctypes.data_as(fp), n.ctypes.data_as(ip), len(audios), stride, out.ctypes.data_as(fp)))
    return out


def transcribe_audio(model, audios: Sequence[np.ndarray], prompt: Sequence[int] = PROMPT, eot: int = EOT,
                     max_loop: int = MAX_LOOP, ignore_eot: bool = False) -> List[List[int]]:
    """PCM in, token ids out (the mel never leaves the GPU)."""
    buf, n, stride = _pack(audios)
    p = np.asarray(prompt, np.int32)
    fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int32)
    opts = _lib.WmDecodeOpts(p.ctypes.data_as(ip), len(p), eot, max_loop, model.pos_mode, int(ignore_eot), None, 0, None, 0)
    total = len(p) + 1 + max_loop
    toks = np.zeros((len(audios), total), np.int32)
    cnt = np.zeros(len(audios), np.int32)
    _lib.check(_lib.lib().wm_transcribe_pcm(model._h, buf.ctypes.data_as(fp), n.ctypes.data_as(ip), len(audios), stride,
                                            C.byref(opts), toks.ctypes.data_as(ip), cnt.ctypes.data_as(ip)))
    return [toks[b, :cnt[b]].tolist() for b in range(len(audios))]


# ---- decoding features the reference lacks, host-level over the same C-ABI -----------------------
HOP = 160  # samples per mel frame (WhisperFeatureExtractor hop_length)


def transcribe_long(model, audio: np.ndarray, prompt: Sequence[int] = PROMPT, eot: int = EOT, max_loop: int = MAX_LOOP,
                    suppress_tokens: Sequence[int] = (), begin_suppress_tokens: Sequence[int] = ()):
    """Long-form audio by consecutive 30 s windows (the reference handles one 30 s clip, main.mojo:22-27): the windows
    are independent utterances, so they go through the batch path `max_batch` at a time.  Returns (per-window id lists
    exactly as `transcribe` would give them, generated ids of all windows concatenated without prompt / eot)."""
    audio = np.asarray(audio, np.float32).ravel()
    win = model.config.n_frames * HOP  # 480 000 samples = 30 s for the released models
    n_win = max(1, -(-len(audio) // win))
    windows = [audio[i * win:(i + 1) * win] for i in range(n_win)]
    per_window: List[List[int]] = []
    for i in range(0, n_win, model.max_batch):
        group = windows[i:i + model.max_batch]
        mels = log_mel(model, group)
        per_window += model.transcribe_batch(mels, prompt=prompt, eot=eot, max_loop=max_loop,
                                             suppress_tokens=suppress_tokens, begin_suppress_tokens=begin_suppress_tokens)
    flat: List[int] = []
    for ids in per_window:
        gen = ids[len(prompt):]
        flat += gen[:-1] if gen and gen[-1] == eot else gen
    return per_window, flat


def detect_language(model, mel, sot: int = 50258, lang_first: int = 50259, lang_last: int = 50357):
    """Whisper's language identification: one decoder step on <|startoftranscript|> and a softmax restricted to the
    language ids (multilingual vocabulary: 50259..50357).  mel: [n_mels, 3000] or [B, n_mels, 3000].
    Returns (ids [B], probabilities [B, n_lang])."""
    from .whisper import KVCache
    m = np.asarray(mel, np.float32)
    batched = m.ndim == 3
    m = m if batched else m[None]
    B = m.shape[0]
    cache = KVCache(model, B)
    model.encoder.forward(m, cache)
    logits = model.decoder.forward(np.full((B, 1), sot, np.int32), None, cache, start_pos=0)
    lang = np.asarray(logits, np.float64).reshape(B, -1)[:, lang_first:lang_last + 1]
    p = np.exp(lang - lang.max(1, keepdims=True))
    p /= p.sum(1, keepdims=True)
    ids = lang_first + p.argmax(1)
    return (ids, p) if batched else (int(ids[0]), p[0])
