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
