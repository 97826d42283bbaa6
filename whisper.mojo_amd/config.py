"""Model dimensions.  Mirrors WhisperConfig (/root/reference/whisper.mojo:15-31) and the compile-time
aliases of /root/reference/config.mojo:4-17, but as run-time parameters so that Whisper-base (BASELINE.json
config 5) and reduced-size test models use the same code.  head_dim is 64 for every Whisper size, which the
reference bakes into its decode path (layers.mojo:190-198)."""
from __future__ import annotations

import ctypes
from dataclasses import dataclass

GELU_TANH = 0  # reference: whisper_tensor.mojo:288-308
GELU_ERF = 1   # HF transformers (the model that produced expected_tokens.txt)
POS_REF = 0    # reference: start_pos = current_len - 1 (whisper.mojo:217)
POS_HF = 1     # HF: position = current_len

DT_F32 = 0
DT_BF16 = 1
DT_F16 = 2
DTYPE_NAMES = {DT_F32: "f32", DT_BF16: "bf16", DT_F16: "f16"}

PROMPT = (50258, 50259, 50359, 50363)  # whisper.mojo:187-191
EOT = 50257                            # whisper.mojo:206
MAX_LOOP = 195                         # whisper.mojo:205


class WmDims(ctypes.Structure):
    """== wm_dims in include/wm_synth.h / include/whisper_mi.h"""
    _fields_ = [(n, ctypes.c_int) for n in
                ("d_model", "n_heads", "n_layers", "ffn", "n_mels", "n_audio_ctx", "n_text_ctx", "vocab")]


@dataclass(frozen=True)
class WhisperConfig:
    d_model: int = 384
    n_heads: int = 6
    n_layers: int = 4
    vocab_size: int = 51865
    ffn: int = 1536
    n_mels: int = 80
    n_audio_ctx: int = 1500
    n_text_ctx: int = 448

    @staticmethod
    def tiny() -> "WhisperConfig":
        """whisper.mojo:30-31"""
        return WhisperConfig(384, 6, 4, 51865, 1536, 80, 1500, 448)

    @staticmethod
    def base() -> "WhisperConfig":
        """BASELINE.json config 5 (not supported by the reference)."""
        return WhisperConfig(512, 8, 6, 51865, 2048, 80, 1500, 448)

    @staticmethod
    def micro() -> "WhisperConfig":
        """Reduced test model: same structure, runs in seconds on a CPU."""
        return WhisperConfig(128, 2, 2, 1000, 512, 16, 100, 64)

    @property
    def n_frames(self) -> int:
        return 2 * self.n_audio_ctx

    def dims(self) -> WmDims:
        return WmDims(self.d_model, self.n_heads, self.n_layers, self.ffn, self.n_mels, self.n_audio_ctx,
                      self.n_text_ctx, self.vocab_size)

    def weight_count(self) -> int:
        d, f, L = self.d_model, self.ffn, self.n_layers
        attn, ln, mlp = 4 * d * d + 3 * d, 2 * d, 2 * f * d + f + d
        enc = d * self.n_mels * 3 + d + d * d * 3 + d + self.n_audio_ctx * d + L * (attn + ln + mlp + ln) + ln
        dec = self.vocab_size * d + self.n_text_ctx * d + L * (2 * (attn + ln) + mlp + ln) + ln
        return enc + dec
