#!/usr/bin/env python3
"""Developer experiment: one thread replays the cross-attention kernel back to back (wm_bench_kernel id 0) while another
replays a chain of one small decode launch (ids 20..29).  How slow does the small launch get under a saturated HBM?"""
import ctypes as C, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisper_mojo_amd import WhisperConfig, _lib, DT_BF16
from whisper_mojo_amd.loader import WeightLoader
from whisper_mojo_amd.whisper import Whisper
L = _lib.lib(); cfg = WhisperConfig.tiny(); B = 64
w = np.empty(cfg.weight_count(), np.float32); d = cfg.dims()
L.wm_synth_weights(C.byref(d), 0, w.ctypes.data_as(C.POINTER(C.c_float)))
mel = np.zeros((B, 80, 3000), np.float32)
def mk():
    m = Whisper(cfg, compute_dtype=DT_BF16, max_batch=B); m.load(WeightLoader.from_array(w))
    st = C.c_void_p(); _lib.check(L.wm_state_new(m._h, B, C.byref(st)))
    _lib.check(L.wm_encode(m._h, st, mel.ctypes.data_as(C.c_void_p), 0, B, None))
    return m, st
(ma, sa), (mb, sb) = mk(), mk()
def run(m, st, which, reps, out, key):
    us = C.c_float()
    _lib.check(L.wm_bench_kernel(m._h, st, which, reps, C.byref(us)))
    out[key] = us.value
ids = [int(a) for a in sys.argv[1:]] or [20, 21, 22, 25, 27, 29]
for small in ids:
    res = {}
    run(mb, sb, small, 2000, res, "alone")
    run(ma, sa, 0, 400, res, "x_alone")
    ta = threading.Thread(target=run, args=(ma, sa, 0, 4000, res, "x_conc"))
    tb = threading.Thread(target=run, args=(mb, sb, small, 2000, res, "conc"))
    ta.start(); time.sleep(0.01); tb.start(); tb.join(); ta.join()
    print(f"id {small}: alone {res['alone']:.2f} us, beside cross-attn {res['conc']:.2f} us   (cross-attn alone {res['x_alone']:.1f}, beside {res['x_conc']:.1f})")
