#!/usr/bin/env python3
"""Converts the reference's headerless fp32 weight file (v1, export_weights.py:19-90) to the v2 container (header + 16-bit
matrices).  Usage: python tools/convert_weights.py whisper_tiny_weights.bin out.wmi2 [--dtype bf16|f16|f32] [--model tiny|base]
[--emb-16bit]"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisper_mojo_amd import WhisperConfig, _lib

ap = argparse.ArgumentParser()
ap.add_argument("v1"); ap.add_argument("v2")
ap.add_argument("--dtype", default="bf16", choices=["f32", "bf16", "f16"])
ap.add_argument("--model", default="tiny", choices=["tiny", "base", "micro"])
ap.add_argument("--emb-16bit", action="store_true", help="store the token embedding in 16 bits too (embedding lookups then use rounded values)")
a = ap.parse_args()
cfg = getattr(WhisperConfig, a.model)()
d = cfg.dims()
_lib.check(_lib.lib().wm_weights_convert_v2(a.v1.encode(), a.v2.encode(), C.byref(d), {"f32": 0, "bf16": 1, "f16": 2}[a.dtype], 0 if a.emb_16bit else 1))
print(a.v2, os.path.getsize(a.v2), "bytes (v1:", os.path.getsize(a.v1), ")")
