"""numpy restatement of include/wm_synth.h: deterministic synthetic weights / mels in the reference's flat
fp32 file format (order of /root/reference/export_weights.py:19-90).  Integer-only recipe, so it is
bit-identical to the C header (tests/test_synth.py)."""
from __future__ import annotations

import numpy as np

from .config import WhisperConfig

K_WEIGHT, K_QK, K_BIAS, K_GAMMA, K_BETA, K_POS, K_EMB = range(7)
_INV_STD = np.float32(2.6428996e-05)
_SCALE = {K_WEIGHT: 0.02, K_QK: 0.04, K_BIAS: 0.02, K_GAMMA: 0.05, K_BETA: 0.05, K_POS: 0.05, K_EMB: 0.05}
_M64 = (1 << 64) - 1


def _mix64(z):
    z = (z + np.uint64(0x9E3779B97F4A7C15))
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def _ih4(seed: int, tensor: int, count: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        base = _mix64(np.uint64((seed * 0x100000001B3 + tensor) & _M64))
        h = _mix64(base + np.arange(count, dtype=np.uint64))
    m = np.uint64(0xFFFF)
    s = ((h & m) + ((h >> np.uint64(16)) & m) + ((h >> np.uint64(32)) & m) + ((h >> np.uint64(48)) & m))
    return s.astype(np.int64) - 131070


def tensor_table(cfg: WhisperConfig):
    """[(name, kind, shape)] in file order."""
    d, f = cfg.d_model, cfg.ffn
    t = []

    def attn(p):
        t.extend([(p + "q.w", K_QK, (d, d)), (p + "q.b", K_BIAS, (d,)), (p + "k.w", K_QK, (d, d)),
                  (p + "v.w", K_WEIGHT, (d, d)), (p + "v.b", K_BIAS, (d,)), (p + "o.w", K_WEIGHT, (d, d)),
                  (p + "o.b", K_BIAS, (d,))])

    def ln(p):
        t.extend([(p + ".w", K_GAMMA, (d,)), (p + ".b", K_BETA, (d,))])

    def mlp(p):
        t.extend([(p + "fc1.w", K_WEIGHT, (f, d)), (p + "fc1.b", K_BIAS, (f,)), (p + "fc2.w", K_WEIGHT, (d, f)),
                  (p + "fc2.b", K_BIAS, (d,))])

    t.extend([("enc.conv1.w", K_WEIGHT, (d, cfg.n_mels, 3)), ("enc.conv1.b", K_BIAS, (d,)),
              ("enc.conv2.w", K_WEIGHT, (d, d, 3)), ("enc.conv2.b", K_BIAS, (d,)),
              ("enc.pos", K_POS, (cfg.n_audio_ctx, d))])
    for l in range(cfg.n_layers):
        p = f"enc.{l}."
        attn(p + "attn."); ln(p + "ln1"); mlp(p); ln(p + "ln2")
    ln("enc.ln")
    t.extend([("dec.tok_emb", K_EMB, (cfg.vocab_size, d)), ("dec.pos", K_POS, (cfg.n_text_ctx, d))])
    for l in range(cfg.n_layers):
        p = f"dec.{l}."
        attn(p + "attn."); ln(p + "ln1"); attn(p + "cross."); ln(p + "lnx"); mlp(p); ln(p + "ln2")
    ln("dec.ln")
    return t


def synth_weights(cfg: WhisperConfig, seed: int = 0) -> np.ndarray:
    """The whole weight file image as one fp32 vector."""
    out = np.empty(cfg.weight_count(), dtype=np.float32)
    off = 0
    for ti, (_, kind, shape) in enumerate(tensor_table(cfg)):
        n = int(np.prod(shape))
        scale = np.float32(_SCALE[kind]) * _INV_STD
        v = _ih4(seed, ti, n).astype(np.float32) * scale
        if kind == K_GAMMA:
            v = np.float32(1.0) + v
        out[off:off + n] = v
        off += n
    assert off == out.size
    return out


def split_weights(cfg: WhisperConfig, flat: np.ndarray) -> dict:
    """name -> view of the flat image."""
    out, off = {}, 0
    for name, _, shape in tensor_table(cfg):
        n = int(np.prod(shape))
        out[name] = flat[off:off + n].reshape(shape)
        off += n
    assert off == flat.size
    return out


def synth_mel(cfg: WhisperConfig, seed: int) -> np.ndarray:
    """[n_mels, n_frames] fp32, clip(0.5*n, -1, 1.5)."""
    n = cfg.n_mels * cfg.n_frames
    v = _ih4(seed, 0x4D454C, n).astype(np.float32) * (np.float32(0.5) * _INV_STD)
    return np.clip(v, np.float32(-1.0), np.float32(1.5)).reshape(cfg.n_mels, cfg.n_frames)


def synth_mels(cfg: WhisperConfig, first_utt: int, count: int) -> np.ndarray:
    """[count, n_mels, n_frames]; utterance i uses seed 1000+i."""
    return np.stack([synth_mel(cfg, 1000 + first_utt + i) for i in range(count)])
