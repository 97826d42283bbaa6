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
