#!/usr/bin/env python3
"""Developer soak: the pipelined tiny / B=64 / bf16 workload for N groups of four passes; every pass must return exactly
the ids of the first one (same inputs) — a race in the GEMM ring, the decode graphs or the slot machinery shows up as a
differing id.  Args: [groups]"""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisper_mojo_amd import WhisperConfig, _lib, DT_BF16
from whisper_mojo_amd.loader import WeightLoader
from whisper_mojo_amd.whisper import Whisper
L = _lib.lib(); cfg = WhisperConfig.tiny(); B = 64
w = np.empty(cfg.weight_count(), np.float32); d = cfg.dims()
L.wm_synth_weights(C.byref(d), 0, w.ctypes.data_as(C.POINTER(C.c_float)))
mels = np.empty((B, 80, 3000), np.float32)
for i in range(B): L.wm_synth_mel_host(1000 + i, 80, 3000, mels[i].ctypes.data_as(C.POINTER(C.c_float)))
mel_dev = torch.from_numpy(mels).cuda()
m = Whisper(cfg, compute_dtype=DT_BF16, max_batch=B); m.load(WeightLoader.from_array(w))
want = m.transcribe_batch(mel_dev, max_loop=99, ignore_eot=True)
groups = int(sys.argv[1]) if len(sys.argv) > 1 else 50
bad = 0; t0 = time.time()
for g in range(groups):
    for s in range(4): m.transcribe_submit(mel_dev, slot=s, max_loop=99, ignore_eot=True)
    for s in range(4):
        got = m.transcribe_wait(s)
        if got != want:
            bad += 1
            diff = [(b, i) for b in range(B) for i in range(len(want[b])) if got[b][i] != want[b][i]]
            print(f"group {g} slot {s}: {len(diff)} differing ids, first {diff[:4]}")
    if g % 10 == 9: print(f"group {g + 1}/{groups}: {bad} bad passes, {time.time() - t0:.1f} s", flush=True)
print("soak done:", groups * 4, "passes,", bad, "bad")
sys.exit(1 if bad else 0)
