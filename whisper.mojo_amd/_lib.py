"""ctypes binding of csrc/libwhispermi.so (include/whisper_mi.h).  Fails loudly when the HIP library is missing:
there is no CPU fallback in this package."""
from __future__ import annotations

import ctypes as C
import os

from .config import WmDims

_HERE = os.path.dirname(os.path.abspath(__file__))
# WM_USE_DEV_LIB=1 (developer tools only) selects the -DWM_DEV build with the A/B switches and debug chains
LIB_PATH = os.path.join(_HERE, "csrc", "libwhispermi_dev.so" if os.environ.get("WM_USE_DEV_LIB") else "libwhispermi.so")
if os.environ.get("WM_USE_DEV_LIB") and os.environ.get("WM_DEV_LIB_PATH"):  # a developer A/B build kept under another name
    LIB_PATH = os.environ["WM_DEV_LIB_PATH"]

# every symbol include/whisper_mi.h declares (tests/test_cabi_symbols.py checks the .so exports all of them)
SYMBOLS = [
    "wm_last_error", "wm_abi_version", "wm_model_load", "wm_model_load_memory", "wm_model_free", "wm_weight_count", "wm_weights_convert_v2", "wm_weights_read", "wm_state_new",
    "wm_state_reset", "wm_state_free", "wm_state_len", "wm_encode", "wm_state_set_encoder_output", "wm_decode_step",
    "wm_transcribe", "wm_transcribe_submit", "wm_transcribe_wait", "wm_transcribe_wait_device", "wm_transcribe_steps", "wm_log_mel", "wm_transcribe_pcm", "wm_op_matmul_nt", "wm_op_ln_matmul_nt", "wm_op_mlp_block", "wm_op_attention", "wm_op_attention_cached", "wm_op_layer_norm", "wm_op_gelu", "wm_op_softmax_rows", "wm_op_conv1d_k3",
    "wm_op_argmax", "wm_bench_kernel", "wm_bench_bytes", "wm_synth_weights", "wm_synth_mel_host",
]

ABI_VERSION = 4  # include/whisper_mi.h WM_ABI_VERSION: the struct layouts below are this version's
KERNEL_CROSS_ATTN, KERNEL_DECODE_STEP, KERNEL_ENCODER, KERNEL_DECODE_STEP_SHARED = 0, 1, 2, 3


class WmConfig(C.Structure):
    _fields_ = [("dims", WmDims), ("gelu_mode", C.c_int), ("compute_dtype", C.c_int), ("kv_dtype", C.c_int),
                ("max_batch", C.c_int), ("decoder_fp32", C.c_int), ("coalesce", C.c_int)]


class WmDecodeOpts(C.Structure):
    _fields_ = [("prompt", C.POINTER(C.c_int32)), ("n_prompt", C.c_int), ("eot", C.c_int), ("max_loop", C.c_int),
                ("pos_mode", C.c_int), ("ignore_eot", C.c_int),
                ("suppress_tokens", C.POINTER(C.c_int32)), ("n_suppress", C.c_int),
                ("begin_suppress_tokens", C.POINTER(C.c_int32)), ("n_begin_suppress", C.c_int),
                ("timestamp_begin", C.c_int), ("no_timestamps_token", C.c_int), ("max_initial_timestamp_index", C.c_int)]


class WhisperMiError(RuntimeError):
    pass


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise WhisperMiError(f"{LIB_PATH} is missing: build it with `python __graft_entry__.py build` "
                             "(hipcc --offload-arch=gfx950).  This package has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    fp, ip, vp = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.c_void_p
    L.wm_last_error.restype = C.c_char_p
    if not hasattr(L, "wm_abi_version") or L.wm_abi_version() != ABI_VERSION:
        got = L.wm_abi_version() if hasattr(L, "wm_abi_version") else "none"
        raise WhisperMiError(f"{LIB_PATH} speaks ABI version {got}, this binding {ABI_VERSION}: rebuild the library "
                             "(python __graft_entry__.py build)")
    L.wm_model_load.argtypes = [C.c_char_p, C.POINTER(WmConfig), C.c_int, C.POINTER(vp)]
    L.wm_model_load_memory.argtypes = [fp, C.c_size_t, C.POINTER(WmConfig), C.c_int, C.POINTER(vp)]
    L.wm_model_free.argtypes = [vp]
    L.wm_model_free.restype = None
    L.wm_weight_count.argtypes = [C.POINTER(WmDims)]
    L.wm_weight_count.restype = C.c_size_t
    L.wm_weights_convert_v2.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(WmDims), C.c_int, C.c_int]
    L.wm_weights_read.argtypes = [C.c_char_p, C.POINTER(WmDims), fp]
    L.wm_state_new.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.wm_state_reset.argtypes = [vp]
    L.wm_state_free.argtypes = [vp]
    L.wm_state_free.restype = None
    L.wm_state_len.argtypes = [vp]
    L.wm_encode.argtypes = [vp, vp, vp, C.c_int, C.c_int, fp]
    L.wm_state_set_encoder_output.argtypes = [vp, vp, fp, C.c_int]
    L.wm_decode_step.argtypes = [vp, vp, ip, C.c_int, ip, fp, ip]
    L.wm_transcribe.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(WmDecodeOpts), ip, ip]
    L.wm_transcribe_submit.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.POINTER(WmDecodeOpts)]
    L.wm_transcribe_wait.argtypes = [vp, C.c_int, ip, ip]
    L.wm_transcribe_steps.argtypes = [vp, C.c_int]
    L.wm_transcribe_wait_device.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int]
    L.wm_log_mel.argtypes = [vp, fp, ip, C.c_int, C.c_int, fp]
    L.wm_transcribe_pcm.argtypes = [vp, fp, ip, C.c_int, C.c_int, C.POINTER(WmDecodeOpts), ip, ip]
    L.wm_op_matmul_nt.argtypes = [fp, fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int]
    L.wm_op_mlp_block.argtypes = [fp] * 10 + [C.c_int] * 5
    L.wm_op_ln_matmul_nt.argtypes = [fp] * 6 + [C.c_int] * 5
    L.wm_op_attention.argtypes = [fp] * 4 + [C.c_int] * 3
    L.wm_op_attention_cached.argtypes = [fp] * 4 + [C.c_int] * 5
    L.wm_op_layer_norm.argtypes = [fp, fp, fp, fp, C.c_int, C.c_int, C.c_float]
    L.wm_op_gelu.argtypes = [fp, C.c_size_t, C.c_int]
    L.wm_op_softmax_rows.argtypes = [fp, C.c_int, C.c_int]
    L.wm_op_conv1d_k3.argtypes = [fp, fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.wm_op_argmax.argtypes = [fp, C.c_int, ip]
    L.wm_bench_kernel.argtypes = [vp, vp, C.c_int, C.c_int, fp]
    L.wm_bench_bytes.argtypes = [vp, vp, C.c_int, C.POINTER(C.c_double)]
    L.wm_synth_weights.argtypes = [C.POINTER(WmDims), C.c_uint64, fp]
    L.wm_synth_weights.restype = C.c_size_t
    L.wm_synth_mel_host.argtypes = [C.c_uint64, C.c_int, C.c_int, fp]
    L.wm_synth_mel_host.restype = None
    _lib = L
    return L


def check(rc: int):
    if rc != 0:
        raise WhisperMiError(f"libwhispermi error {rc}: {lib().wm_last_error().decode()}")
