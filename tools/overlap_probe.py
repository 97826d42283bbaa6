#!/usr/bin/env python3
"""Developer experiment: do the encoder (MFMA-bound, big grids) and the decode step (latency / HBM-bound, small grids)
overlap when issued from two HIP streams?  Two model instances (own streams), two host threads."""
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
    us = C.c_float(); t0 = time.perf_counter()
    _lib.check(L.wm_bench_kernel(m._h, st, which, reps, C.byref(us)))
    out[key] = (time.perf_counter() - t0) * 1e3, us.value
res = {}
run(ma, sa, 2, 8, res, "enc alone"); run(mb, sb, 1, 200, res, "dec alone")
ta = threading.Thread(target=run, args=(ma, sa, 2, 8, res, "enc concurrent")); tb = threading.Thread(target=run, args=(mb, sb, 1, 200, res, "dec concurrent"))
t0 = time.perf_counter(); ta.start(); tb.start(); ta.join(); tb.join(); wall = (time.perf_counter() - t0) * 1e3
for k, v in res.items(): print(f"{k:16s} wall {v[0]:8.2f} ms   per-rep {v[1]:9.1f} us")
print(f"both concurrently: wall {wall:.2f} ms (sum of alone = {res['enc alone'][0] + res['dec alone'][0]:.2f} ms)")
