"""whisper.mojo_amd — MI355X-native Whisper inference hot path behind the call surface of
antonvice/whisper.Mojo (load flat fp32 weights, 80x3000 log-mel in, greedy token ids out).

Host side (this package, Python) mirrors the reference's Mojo interface; all compute goes through the C-ABI
of csrc/libwhispermi.so (include/whisper_mi.h) into hand-written HIP kernels for gfx950.  There is no CPU
fallback: importing the compute modules without the built library raises."""
import os as _os

# One HIP stream per pipeline slot (wm_transcribe_submit): ROCm multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues
# (default 4) and passes that share a queue serialise.  Only effective if set before the HIP runtime initialises.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from .config import (WhisperConfig, GELU_TANH, GELU_ERF, POS_REF, POS_HF, DT_F32, DT_BF16, DT_F16, PROMPT, EOT,
                     MAX_LOOP)

__all__ = ["WhisperConfig", "GELU_TANH", "GELU_ERF", "POS_REF", "POS_HF", "DT_F32", "DT_BF16", "DT_F16", "PROMPT",
           "EOT", "MAX_LOOP"]
