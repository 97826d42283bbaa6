"""CPU-only: the C-ABI library loads and exports every symbol include/whisper_mi.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "whisper_mi.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(wm_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_the_boundary():
    syms = declared_symbols()
    for s in ("wm_model_load", "wm_transcribe", "wm_encode", "wm_decode_step", "wm_state_new", "wm_op_matmul_nt",
              "wm_op_layer_norm", "wm_op_gelu", "wm_op_softmax_rows", "wm_op_conv1d_k3", "wm_op_argmax"):
        assert s in syms


def test_library_exports_every_declared_symbol():
    from whisper_mojo_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(_lib.SYMBOLS) == declared_symbols()


def test_missing_library_fails_loudly(monkeypatch):
    from whisper_mojo_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libwhispermi.so")
    with pytest.raises(_lib.WhisperMiError):
        _lib.lib()


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under whisper.mojo_amd/ may reference it."""
    pkg = os.path.join(ROOT, "whisper.mojo_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.replace("test_oracle", ""), f"{f} mentions the oracle"


def test_package_import_sets_hw_queue_default():
    """Pipelined passes need one hardware queue per slot stream: the package sets GPU_MAX_HW_QUEUES before HIP
    initialises unless the user already chose a value."""
    import os
    import whisper_mojo_amd  # noqa: F401
    assert int(os.environ.get("GPU_MAX_HW_QUEUES", "0")) >= 8


def test_op_wrappers_validate_shapes_before_calling_the_library():
    """The host mirrors of the op entry points refuse wrong shapes themselves (no GPU, no library call needed to see it)."""
    import numpy as np
    import pytest
    from whisper_mojo_amd import whisper_tensor as wt
    x = np.zeros((4, 128), np.float32)
    with pytest.raises(ValueError):  # fc1 must be [ffn, d]
        wt.mlp_block(x, np.ones(128), np.zeros(128), np.zeros((256, 64), np.float32), np.zeros(256), np.zeros((128, 256), np.float32), np.zeros(128))
    with pytest.raises(ValueError):  # fc2 must be [d, ffn]
        wt.mlp_block(x, np.ones(128), np.zeros(128), np.zeros((256, 128), np.float32), np.zeros(256), np.zeros((128, 128), np.float32), np.zeros(128))
    with pytest.raises(ValueError):  # in place: x must be a C-contiguous float32 array
        wt.mlp_block(x.astype(np.float64), np.ones(128), np.zeros(128), np.zeros((256, 128), np.float32), np.zeros(256), np.zeros((128, 256), np.float32), np.zeros(128))
    q = np.zeros((10, 128), np.float32)
    with pytest.raises(ValueError):  # d must be 64 * n_heads
        wt.attention(np.zeros_like(q), q, q, q, 3)
    with pytest.raises(ValueError):  # k of another length
        wt.attention(np.zeros_like(q), q, np.zeros((9, 128), np.float32), q, 2)
    with pytest.raises(ValueError):  # out of the wrong shape
        wt.attention(np.zeros((10, 64), np.float32), q, q, q, 2)
    with pytest.raises(ValueError):  # cached keys must be [B, t, d]
        wt.attention_cached(np.zeros_like(q), q, q, q, 2)
    with pytest.raises(ValueError):  # another batch size in the cache
        wt.attention_cached(np.zeros_like(q), q, np.zeros((9, 5, 128), np.float32), np.zeros((9, 5, 128), np.float32), 2)
