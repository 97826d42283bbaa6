"""Host-side mirror of /root/reference/whisper.mojo + layers.mojo's KVCache: same names, argument meaning and
error behaviour, every forward a call through the C-ABI (include/whisper_mi.h) into the HIP path.

    whisper = Whisper()                                  # whisper.mojo:175-178
    whisper.load(WeightLoader("whisper_tiny_weights.bin"))   # main.mojo:16-17
    tokens = whisper.transcribe(mel)                     # main.mojo:30, mel = [80, 3000] fp32

Extensions the reference lacks: a leading batch dimension (transcribe_batch), device-resident mels (torch CUDA
tensors), 16-bit operand / KV-cache dtypes, Whisper-base dims."""
from __future__ import annotations

import ctypes as C
import weakref
from typing import List, Optional, Sequence

import numpy as np

from . import _lib
from .config import (DT_F32, EOT, GELU_TANH, MAX_LOOP, POS_REF, PROMPT, WhisperConfig)
from .loader import WeightLoader


def _fp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int32))


def _mel_arg(mel, cfg: WhisperConfig):
    """-> (pointer, on_device, B, keepalive).  Accepts numpy [B,n_mels,T] / [n_mels,T] or a torch CUDA tensor."""
    shape_tail = (cfg.n_mels, cfg.n_frames)
    if isinstance(mel, np.ndarray):
        a = np.ascontiguousarray(mel, np.float32)
        if a.ndim == 2:
            a = a[None]
        if a.shape[1:] != shape_tail:
            raise ValueError(f"mel must be [B, {cfg.n_mels}, {cfg.n_frames}], got {a.shape}")
        return a.ctypes.data_as(C.c_void_p), 0, a.shape[0], a
    import torch
    if not isinstance(mel, torch.Tensor):
        raise TypeError("mel must be a numpy array or a torch tensor")
    t = mel if mel.dim() == 3 else mel[None]
    if tuple(t.shape[1:]) != shape_tail:
        raise ValueError(f"mel must be [B, {cfg.n_mels}, {cfg.n_frames}], got {tuple(t.shape)}")
    if not t.is_cuda:
        return _mel_arg(t.numpy(), cfg)
    t = t.contiguous().float()
    torch.cuda.current_stream(t.device).synchronize()  # the library runs on its own HIP stream
    return C.c_void_p(t.data_ptr()), 1, t.shape[0], t


class KVCache:
    """layers.mojo:55-69 — KVCache(n_layers, d_model, max_len) for a batch of utterances; owns the device-side
    self/cross K/V arena (wm_state)."""

    def __init__(self, model: "Whisper", batch: int = 1):
        self.model = model
        self.batch = batch
        h = C.c_void_p()
        _lib.check(_lib.lib().wm_state_new(model._h, batch, C.byref(h)))
        self._h = h
        # the model owns the device arenas of its caches (wm_model_free frees them): remember WHICH loaded model this is, so
        # a cache that outlives a reload / close() neither frees nor uses memory that went away with the old model
        self._model_h = model._h.value
        model._caches.add(self)

    @property
    def current_len(self) -> int:
        """LayerCache.current_len (layers.mojo:18) — same for every layer."""
        return _lib.lib().wm_state_len(self._h)

    def reset(self):
        _lib.check(_lib.lib().wm_state_reset(self._h))

    def _invalidate(self):
        """Whisper.close() / load(): the library freed this cache's state together with its model."""
        self._h = None

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        mh = getattr(getattr(self, "model", None), "_h", None)
        if h and mh is not None and mh.value == getattr(self, "_model_h", None):
            _lib.lib().wm_state_free(h)


class WhisperEncoder:
    """whisper.mojo:34-99"""

    def __init__(self, model: "Whisper"):
        self._m = model

    def forward(self, mel, cache: Optional[KVCache] = None) -> np.ndarray:
        """whisper.mojo:71-99: mel [n_mels, 3000] -> [1500, d_model]  (or batched [B,...]).  When `cache` is given
        the encoder output also stays on the device inside it for WhisperDecoder.forward."""
        m = self._m
        ptr, on_dev, B, keep = _mel_arg(mel, m.config)
        cache = cache or KVCache(m, B)
        out = np.empty((B, m.config.n_audio_ctx, m.config.d_model), np.float32)
        _lib.check(_lib.lib().wm_encode(m._h, cache._h, ptr, on_dev, B, _fp(out)))
        single = (isinstance(mel, np.ndarray) and mel.ndim == 2) or (not isinstance(mel, np.ndarray) and mel.dim() == 2)
        return out[0] if single else out


class WhisperDecoder:
    """whisper.mojo:102-167"""

    def __init__(self, model: "Whisper"):
        self._m = model

    def forward(self, tokens, enc_out, cache: KVCache, use_cache: bool = True, start_pos=0) -> np.ndarray:
        """whisper.mojo:130-167.  tokens: List[int] (one utterance) or [B, q_len]; enc_out: the encoder output
        ([1500,d] / [B,1500,d]) — uploaded and projected to cross K/V on the cache's first use
        (layers.mojo:150-154) — or None when `cache` already holds one from WhisperEncoder.forward(mel, cache).
        Returns logits [vocab] / [B, vocab] for the last position."""
        if not use_cache:
            raise NotImplementedError("the decode path always runs KV-cached, as Whisper.transcribe does (whisper.mojo:195,212)")
        m = self._m
        t = np.asarray(tokens, np.int32)
        single = t.ndim == 1
        t = np.ascontiguousarray(t.reshape(1, -1) if single else t)
        B, q_len = t.shape
        if B != cache.batch:
            raise ValueError(f"cache was created for batch {cache.batch}, tokens have batch {B}")
        if enc_out is not None and cache.current_len == 0:
            e = np.ascontiguousarray(enc_out, np.float32).reshape(B, m.config.n_audio_ctx, m.config.d_model)
            _lib.check(_lib.lib().wm_state_set_encoder_output(m._h, cache._h, _fp(e), B))
        sp = np.ascontiguousarray(np.broadcast_to(np.asarray(start_pos, np.int32), (B,)))
        logits = np.empty((B, m.config.vocab_size), np.float32)
        _lib.check(_lib.lib().wm_decode_step(m._h, cache._h, _ip(t), q_len, _ip(sp), _fp(logits), None))
        return logits[0] if single else logits


class Whisper:
    """whisper.mojo:169-223"""

    def __init__(self, config: Optional[WhisperConfig] = None, compute_dtype: int = DT_F32, kv_dtype: Optional[int] = None,
                 gelu_mode: int = GELU_TANH, pos_mode: int = POS_REF, max_batch: int = 64, device: int = 0,
                 decoder_fp32: bool = False, coalesce: int = 0):
        self.config = config or WhisperConfig.tiny()
        self.compute_dtype = compute_dtype
        self.kv_dtype = compute_dtype if kv_dtype is None else kv_dtype
        self.gelu_mode = gelu_mode
        self.pos_mode = pos_mode
        self.max_batch = max_batch
        self.device = device
        self.decoder_fp32 = decoder_fp32  # compute_dtype narrows the encoder only; decoder weights / operands stay fp32
        self.coalesce = coalesce  # 2: consecutive transcribe_submit calls of equal batch size / options share one 2·B-row decode state
        self._h = None
        self._caches = weakref.WeakSet()  # live KVCaches of the loaded model
        self.encoder = WhisperEncoder(self)
        self.decoder = WhisperDecoder(self)

    def _cfg(self) -> _lib.WmConfig:
        return _lib.WmConfig(self.config.dims(), self.gelu_mode, self.compute_dtype, self.kv_dtype, self.max_batch,
                             int(self.decoder_fp32), int(self.coalesce))

    def load(self, loader: WeightLoader):
        """whisper.mojo:180-182.  Raises if the image size does not match the config."""
        self.close()
        h = C.c_void_p()
        cfg = self._cfg()
        w = loader.raw_data
        _lib.check(_lib.lib().wm_model_load_memory(_fp(w), w.size, C.byref(cfg), self.device, C.byref(h)))
        self._h = h

    def load_file(self, path: str):
        self.close()
        h = C.c_void_p()
        cfg = self._cfg()
        _lib.check(_lib.lib().wm_model_load(path.encode(), C.byref(cfg), self.device, C.byref(h)))
        self._h = h

    def close(self):
        h, self._h = self._h, None
        if h:
            for c in list(self._caches):
                c._invalidate()
            self._caches.clear()
            self._pending = {}
            _lib.lib().wm_model_free(h)  # also frees every state (KVCache arena, pipeline slot) created on it

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _opts(self, prompt, eot, max_loop, ignore_eot, suppress_tokens=(), begin_suppress_tokens=(), timestamps=None):
        """timestamps: None (the reference: raw argmax) or (timestamp_begin, no_timestamps_id, max_initial_timestamp_index | None)
        — HF's WhisperTimeStampLogitsProcessor inside the fused argmax."""
        p = np.asarray(prompt, np.int32)
        sup = np.asarray(list(suppress_tokens), np.int32)
        bsup = np.asarray(list(begin_suppress_tokens), np.int32)
        tb, no_ts, max_init = (0, -1, -1) if timestamps is None else (int(timestamps[0]), int(timestamps[1]),
                                                                      -1 if timestamps[2] is None else int(timestamps[2]))
        opts = _lib.WmDecodeOpts(_ip(p), len(p), eot, max_loop, self.pos_mode, int(ignore_eot),
                                 _ip(sup) if len(sup) else None, len(sup), _ip(bsup) if len(bsup) else None, len(bsup),
                                 tb, no_ts, max_init)
        return opts, (p, sup, bsup)

    def transcribe_batch(self, mel, prompt: Sequence[int] = PROMPT, eot: int = EOT, max_loop: int = MAX_LOOP,
                         ignore_eot: bool = False, suppress_tokens: Sequence[int] = (),
                         begin_suppress_tokens: Sequence[int] = (), timestamps=None) -> List[List[int]]:
        """Batched Whisper.transcribe: one List[int] per utterance = prompt + generated ids (+ eot when hit)."""
        if self._h is None:
            raise _lib.WhisperMiError("model not loaded")
        ptr, on_dev, B, keep = _mel_arg(mel, self.config)
        opts, keep2 = self._opts(prompt, eot, max_loop, ignore_eot, suppress_tokens, begin_suppress_tokens, timestamps)
        p = keep2[0]
        total = len(p) + 1 + max_loop
        toks = np.zeros((B, total), np.int32)
        n = np.zeros(B, np.int32)
        _lib.check(_lib.lib().wm_transcribe(self._h, ptr, on_dev, B, C.byref(opts), _ip(toks), _ip(n)))
        self.last_tokens, self.last_counts = toks, n
        return [toks[b, :n[b]].tolist() for b in range(B)]

    def transcribe_submit(self, mel, slot: int = 0, prompt: Sequence[int] = PROMPT, eot: int = EOT, max_loop: int = MAX_LOOP,
                          ignore_eot: bool = False, suppress_tokens: Sequence[int] = (), begin_suppress_tokens: Sequence[int] = (),
                          timestamps=None):
        """Pipelined form (wm_transcribe_submit): enqueue encoder + greedy loop for this batch on pipeline slot 0..7 and
        return at once; `transcribe_wait(slot)` collects the ids.  Submitting batch i+1 before waiting for batch i lets
        its encoder overlap batch i's decode."""
        if self._h is None:
            raise _lib.WhisperMiError("model not loaded")
        ptr, on_dev, B, keep = _mel_arg(mel, self.config)
        opts, keep2 = self._opts(prompt, eot, max_loop, ignore_eot, suppress_tokens, begin_suppress_tokens, timestamps)
        p = keep2[0]
        _lib.check(_lib.lib().wm_transcribe_submit(self._h, slot, ptr, on_dev, B, C.byref(opts)))
        self._pending = getattr(self, "_pending", {})
        self._pending[slot] = (B, len(p) + 1 + max_loop, keep)

    def transcribe_wait(self, slot: int = 0) -> List[List[int]]:
        B, total, _keep = self._pending.pop(slot)
        toks = np.zeros((B, total), np.int32)
        n = np.zeros(B, np.int32)
        _lib.check(_lib.lib().wm_transcribe_wait(self._h, slot, _ip(toks), _ip(n)))
        self.last_tokens, self.last_counts = toks, n
        return [toks[b, :n[b]].tolist() for b in range(B)]

    def transcribe_wait_device(self, slot: int, packed) -> None:
        """transcribe_wait with the ids left on the GPU as the multi-GPU gather buffer: `packed` is a torch int32 CUDA tensor
        [rows, 1 + stride] on this model's device; row r becomes [length, ids zero-padded] (dist.gather_tokens_device)."""
        B, total, _keep = self._pending.pop(slot)
        rows, width = int(packed.shape[0]), int(packed.shape[1])
        if not packed.is_cuda or packed.dtype.itemsize != 4 or not packed.is_contiguous():
            raise ValueError("packed must be a contiguous int32 CUDA tensor")
        _lib.check(_lib.lib().wm_transcribe_wait_device(self._h, slot, C.c_void_p(packed.data_ptr()), rows, width - 1))
        self.last_tokens = self.last_counts = None

    def loop_steps(self, slot: int = 0) -> int:
        """Loop iterations (whisper.mojo:205) enqueued for the slot's most recent completed pass: max_loop unless the early exit
        (every utterance emitted eot, whisper.mojo:206-207) cut the loop.  Slot 0 also serves transcribe_batch."""
        return int(_lib.lib().wm_transcribe_steps(self._h, slot))

    def transcribe(self, mel) -> List[int]:
        """whisper.mojo:184-223: mel [80, 3000] -> token ids."""
        return self.transcribe_batch(mel)[0]
