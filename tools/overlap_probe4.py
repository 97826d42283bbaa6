#!/usr/bin/env python3
"""Developer experiment: two decode chains started with a random phase offset — does the per-step time depend on the
offset (cross-attention phases interleaved vs coinciding)?"""
import ctypes as C, os, sys, threading, time, random
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
inst = [mk() for _ in range(2)]
def run(m, st, reps, out, key, delay):
    time.sleep(delay)
    us = C.c_float()
    _lib.check(L.wm_bench_kernel(m._h, st, 1, reps, C.byref(us)))
    out[key] = us.value
random.seed(3)
for trial in range(12):
    res = {}
    dl = random.random() * 2e-3
    th = [threading.Thread(target=run, args=(inst[i][0], inst[i][1], 600 if i == 0 else 300, res, i, 0 if i == 0 else 0.02 + dl)) for i in range(2)]
    [t.start() for t in th]; [t.join() for t in th]
    print(f"trial {trial}: delay {dl*1e3:.3f} ms: chain0 (600 steps, partly alone) {res[0]:.1f} us/step, chain1 (300 steps, always beside chain0) {res[1]:.1f} us/step")
