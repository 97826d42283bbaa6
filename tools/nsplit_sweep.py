#!/usr/bin/env python3
"""Developer sweep: cross-attention launch time and whole decode step vs WM_NSPLIT (one process per value)."""
import os, subprocess, sys
code = r'''
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, %r)
from whisper_mojo_amd import WhisperConfig, _lib, DT_BF16
from whisper_mojo_amd.loader import WeightLoader
from whisper_mojo_amd.whisper import Whisper
L = _lib.lib(); cfg = WhisperConfig.tiny(); B = 64
w = np.empty(cfg.weight_count(), np.float32); d = cfg.dims()
L.wm_synth_weights(C.byref(d), 0, w.ctypes.data_as(C.POINTER(C.c_float)))
mel = np.zeros((B, 80, 3000), np.float32)
m = Whisper(cfg, compute_dtype=DT_BF16, max_batch=B); m.load(WeightLoader.from_array(w))
st = C.c_void_p(); _lib.check(L.wm_state_new(m._h, B, C.byref(st)))
_lib.check(L.wm_encode(m._h, st, mel.ctypes.data_as(C.c_void_p), 0, B, None))
us = C.c_float(); out = []
for which, reps in ((0, 400), (1, 100), (0, 400), (1, 100)):
    _lib.check(L.wm_bench_kernel(m._h, st, which, reps, C.byref(us))); out.append(us.value)
print("nsplit %%s: cross-attn %%.2f / %%.2f us, step %%.1f / %%.1f us" %% (os.environ.get("WM_NSPLIT"), out[0], out[2], out[1], out[3]))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for n in sys.argv[1:]:
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, WM_NSPLIT=n), check=True)
