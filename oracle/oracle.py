"""ctypes binding of the CPU oracle (oracle/whisper_oracle.c).  TEST INFRASTRUCTURE ONLY: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg — never by the product package."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libwhisper_oracle.so")


class WoConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("d_model", "n_heads", "n_layers", "ffn", "n_mels", "n_audio_ctx",
                                       "n_text_ctx", "vocab", "gelu_mode")]


class WmDims(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("d_model", "n_heads", "n_layers", "ffn", "n_mels", "n_audio_ctx",
                                       "n_text_ctx", "vocab")]


def build(force: bool = False) -> str:
    """make -C oracle when the sources' CONTENT changed (sha256 kept in a .stamp file: mtimes do not survive the snapshot
    copy to the GPU box, and an mtime rule would recompile there)."""
    import hashlib
    srcs = [os.path.join(_HERE, f) for f in ("whisper_oracle.c", "oracle_synth.c", "Makefile")]
    srcs.append(os.path.join(_HERE, "..", "include", "wm_synth.h"))
    h = hashlib.sha256()
    for p in srcs:
        with open(p, "rb") as f:
            h.update(os.path.basename(p).encode() + b"\0" + f.read())
    digest, stamp = h.hexdigest(), _SO + ".stamp"
    try:
        fresh = os.path.exists(_SO) and open(stamp).read().strip() == digest
    except OSError:
        fresh = False
    if force or not fresh:
        subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
        with open(stamp, "w") as f:
            f.write(digest + "\n")
    return _SO


_lib = None


def usable_cpus() -> int:
    """CPUs this process may actually use: the cgroup quota when there is one (a GPU box shows 256 host cores to a job that
    owns 16), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def _cap_openmp_threads():
    """libgomp defaults to one thread per visible core; with 256 visible and 16 usable every small parallel region of the
    micro model spins against the quota (round 1's smoke() spent 90 s here).  OMP_NUM_THREADS, when set, wins."""
    if os.environ.get("OMP_NUM_THREADS"):
        return
    try:
        C.CDLL("libgomp.so.1").omp_set_num_threads(max(1, min(usable_cpus(), 16)))
    except OSError:
        pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int32)
        L.wo_matmul.argtypes = [fp, fp, fp, fp, C.c_int, C.c_int, C.c_int]
        L.wo_layer_norm.argtypes = [fp, fp, fp, fp, C.c_int, C.c_int, C.c_float]
        L.wo_gelu.argtypes = [fp, C.c_size_t, C.c_int]
        L.wo_softmax.argtypes = [fp, C.c_int, C.c_int]
        L.wo_transpose_conv_weights.argtypes = [fp, fp, C.c_int, C.c_int, C.c_int]
        L.wo_conv1d.argtypes = [fp, fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.wo_argmax.argtypes = [fp, C.c_int]
        L.wo_argmax.restype = C.c_int
        L.wo_weight_count.argtypes = [C.POINTER(WoConfig)]
        L.wo_weight_count.restype = C.c_size_t
        L.wo_model_from_memory.argtypes = [fp, C.c_size_t, C.POINTER(WoConfig)]
        L.wo_model_from_memory.restype = C.c_void_p
        L.wo_model_load.argtypes = [C.c_char_p, C.POINTER(WoConfig)]
        L.wo_model_load.restype = C.c_void_p
        L.wo_model_free.argtypes = [C.c_void_p]
        L.wo_cache_new.argtypes = [C.c_void_p, C.c_int]
        L.wo_cache_new.restype = C.c_void_p
        L.wo_cache_free.argtypes = [C.c_void_p]
        L.wo_cache_len.argtypes = [C.c_void_p]
        L.wo_cache_len.restype = C.c_int
        L.wo_encode.argtypes = [C.c_void_p, fp, fp]
        L.wo_encoder_stem.argtypes = [C.c_void_p, fp, fp]
        L.wo_decoder_forward.argtypes = [C.c_void_p, ip, C.c_int, fp, C.c_void_p, C.c_int, fp]
        L.wo_transcribe.argtypes = [C.c_void_p, fp, fp, ip, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, ip, fp]
        L.wo_transcribe.restype = C.c_int
        L.wo_transcribe_ex.argtypes = [C.c_void_p, fp, fp, ip, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, ip, C.c_int, ip, C.c_int, ip, fp]
        L.wo_transcribe_ex.restype = C.c_int
        L.wo_transcribe_ts.argtypes = [C.c_void_p, fp, fp, ip, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, ip, C.c_int, ip, C.c_int,
                                       C.c_int, C.c_int, C.c_int, ip, fp]
        L.wo_transcribe_ts.restype = C.c_int
        L.wo_timestamp_rules.argtypes = [fp, C.c_int, ip, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.wo_timestamp_rules.restype = None
        L.wo_teacher_forced.argtypes = [C.c_void_p, fp, ip, C.c_int, C.c_int, C.c_int, fp]
        L.wo_synth_count.argtypes = [C.POINTER(WmDims)]
        L.wo_synth_count.restype = C.c_size_t
        L.wo_synth_fill.argtypes = [C.POINTER(WmDims), C.c_uint64, fp]
        L.wo_synth_fill.restype = C.c_size_t
        L.wo_synth_mel.argtypes = [C.c_uint64, C.c_int, C.c_int, fp]
        _cap_openmp_threads()
        _lib = L
    return _lib


def _fp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _cfg(cfg, gelu_mode=0) -> WoConfig:
    return WoConfig(cfg.d_model, cfg.n_heads, cfg.n_layers, cfg.ffn, cfg.n_mels, cfg.n_audio_ctx, cfg.n_text_ctx,
                    cfg.vocab_size, gelu_mode)


def _dims(cfg) -> WmDims:
    return WmDims(cfg.d_model, cfg.n_heads, cfg.n_layers, cfg.ffn, cfg.n_mels, cfg.n_audio_ctx, cfg.n_text_ctx,
                  cfg.vocab_size)


# ---- ops (whisper_tensor.mojo) ------------------------------------------------------------------------
def matmul(A, B, bias=None):
    A, B = _f32(A), _f32(B)
    M, K = A.shape
    N = B.shape[0]
    out = np.empty((M, N), np.float32)
    b = None if bias is None else _f32(bias)
    lib().wo_matmul(_fp(out), _fp(A), _fp(B), _fp(b), M, N, K)
    return out


def layer_norm(x, gamma, beta, eps=1e-5):
    x, g, b = _f32(x), _f32(gamma), _f32(beta)
    out = np.empty_like(x)
    lib().wo_layer_norm(_fp(out), _fp(x), _fp(g), _fp(b), x.shape[0], x.shape[1], eps)
    return out


def gelu(x, mode=0):
    t = _f32(x).copy()
    lib().wo_gelu(_fp(t), t.size, mode)
    return t


def softmax(x):
    t = _f32(x).copy()
    lib().wo_softmax(_fp(t), t.shape[0], t.shape[1])
    return t


def transpose_conv_weights(w):
    w = _f32(w)
    co, ci, k = w.shape
    out = np.empty((co * k, ci), np.float32)
    lib().wo_transpose_conv_weights(_fp(out), _fp(w), co, ci, k)
    return out


def conv1d(inp, weight_T, bias, stride, padding, out_T=False):
    inp, w, b = _f32(inp), _f32(weight_T), _f32(bias)
    C_in, L_in = inp.shape
    C_out = w.shape[0] // 3
    L_out = (L_in + 2 * padding - 3) // stride + 1
    out = np.empty((L_out, C_out) if out_T else (C_out, L_out), np.float32)
    lib().wo_conv1d(_fp(out), _fp(inp), _fp(w), _fp(b), C_in, L_in, C_out, stride, padding, int(out_T))
    return out


def timestamp_rules(scores, seq, timestamp_begin, no_timestamps, eos, max_initial=None):
    """HF WhisperTimeStampLogitsProcessor.__call__ on one row of scores; returns the processed copy."""
    t = _f32(scores).copy()
    q = np.ascontiguousarray(np.asarray(list(seq) or [0], np.int32))
    lib().wo_timestamp_rules(_fp(t), t.size, _ip(q), len(seq), timestamp_begin, no_timestamps, eos, -1 if max_initial is None else max_initial)
    return t


def argmax(x):
    x = _f32(x).ravel()
    return int(lib().wo_argmax(_fp(x), x.size))


# ---- synthetic generator (C side) ----------------------------------------------------------------------
def synth_weights_c(cfg, seed=0):
    d = _dims(cfg)
    n = lib().wo_synth_count(C.byref(d))
    out = np.empty(n, np.float32)
    lib().wo_synth_fill(C.byref(d), seed, _fp(out))
    return out


def synth_mel_c(cfg, seed):
    out = np.empty((cfg.n_mels, 2 * cfg.n_audio_ctx), np.float32)
    lib().wo_synth_mel(seed, cfg.n_mels, 2 * cfg.n_audio_ctx, _fp(out))
    return out


# ---- model ---------------------------------------------------------------------------------------------
class OracleModel:
    """Restates Whisper (whisper.mojo:169-223) on the CPU."""

    def __init__(self, cfg, weights: np.ndarray, gelu_mode=0):
        self.cfg = cfg
        self.gelu_mode = gelu_mode
        w = _f32(weights).ravel()
        c = _cfg(cfg, gelu_mode)
        self._h = lib().wo_model_from_memory(_fp(w), w.size, C.byref(c))
        if not self._h:
            raise ValueError(f"weight image has {w.size} floats, expected {lib().wo_weight_count(C.byref(c))}")

    @classmethod
    def from_file(cls, cfg, path, gelu_mode=0):
        self = cls.__new__(cls)
        self.cfg, self.gelu_mode = cfg, gelu_mode
        c = _cfg(cfg, gelu_mode)
        self._h = lib().wo_model_load(path.encode(), C.byref(c))
        if not self._h:
            raise ValueError(f"cannot load {path} (missing, or byte size != expected)")
        return self

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib().wo_model_free(h)

    def encode(self, mel):
        mel = _f32(mel)
        assert mel.shape == (self.cfg.n_mels, 2 * self.cfg.n_audio_ctx)
        out = np.empty((self.cfg.n_audio_ctx, self.cfg.d_model), np.float32)
        lib().wo_encode(self._h, _fp(mel), _fp(out))
        return out

    def encoder_stem(self, mel):
        mel = _f32(mel)
        out = np.empty((self.cfg.n_audio_ctx, self.cfg.d_model), np.float32)
        lib().wo_encoder_stem(self._h, _fp(mel), _fp(out))
        return out

    def transcribe(self, mel=None, enc_out=None, prompt=(50258, 50259, 50359, 50363), eot=50257, max_loop=195,
                   pos_mode=0, ignore_eot=False, want_logits=False, suppress_tokens=(), begin_suppress_tokens=(), timestamps=None):
        """timestamps: None (the reference: raw argmax) or (timestamp_begin, no_timestamps_id, max_initial_timestamp_index | None):
        HF's WhisperTimeStampLogitsProcessor after the suppress masks."""
        p = np.asarray(prompt, np.int32)
        toks = np.zeros(len(p) + 1 + max_loop, np.int32)
        logits = np.zeros((1 + max_loop, self.cfg.vocab_size), np.float32) if want_logits else None
        m = None if mel is None else _f32(mel)
        e = None if enc_out is None else _f32(enc_out)
        sup = np.asarray(list(suppress_tokens) or [0], np.int32)
        bsup = np.asarray(list(begin_suppress_tokens) or [0], np.int32)
        tb, no_ts, max_init = (0, -1, -1) if timestamps is None else (timestamps[0], timestamps[1], -1 if timestamps[2] is None else timestamps[2])
        n = lib().wo_transcribe_ts(self._h, _fp(m), _fp(e), _ip(p), len(p), eot, max_loop, pos_mode, int(ignore_eot),
                                   _ip(sup), len(suppress_tokens), _ip(bsup), len(begin_suppress_tokens), tb, no_ts, max_init,
                                   _ip(toks), _fp(logits))
        toks = toks[:n].copy()
        if want_logits:
            return toks, logits[:n - len(p)].copy()
        return toks

    def teacher_forced(self, enc_out, forced, n_prompt=4, pos_mode=0):
        f = np.asarray(forced, np.int32)
        e = _f32(enc_out)
        logits = np.empty((len(f) - n_prompt + 1, self.cfg.vocab_size), np.float32)
        lib().wo_teacher_forced(self._h, _fp(e), _ip(f), n_prompt, len(f), pos_mode, _fp(logits))
        return logits
