#!/usr/bin/env python3
"""Developer experiment: N decode chains (graph replays of one step each, own model + state + stream) issued
concurrently from N host threads — how well do latency-bound chains fill each other's holes?"""
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
NMAX = int(sys.argv[1]) if len(sys.argv) > 1 else 4
which = int(sys.argv[2]) if len(sys.argv) > 2 else 1
inst = [mk() for _ in range(NMAX)]
def run(m, st, reps, out, key):
    us = C.c_float()
    _lib.check(L.wm_bench_kernel(m._h, st, which, reps, C.byref(us)))
    out[key] = us.value
for n in range(1, NMAX + 1):
    res = {}
    th = [threading.Thread(target=run, args=(inst[i][0], inst[i][1], 400, res, i)) for i in range(n)]
    t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; wall = (time.perf_counter() - t0) * 1e3
    print(f"{n} chains: per-step {[round(res[i],1) for i in range(n)]} us; wall {wall:.1f} ms; aggregate {wall*1e3/400/n:.1f} us per (chain-)step")
