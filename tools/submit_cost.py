#!/usr/bin/env python3
"""Developer probe: host-side cost of wm_transcribe_submit / wm_transcribe_wait per group of eight coalesced submits (headline configuration)."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from whisper_mojo_amd import WhisperConfig, _lib, DT_BF16, DT_F32
from whisper_mojo_amd.loader import WeightLoader
from whisper_mojo_amd.whisper import Whisper
L = _lib.lib(); cfg = WhisperConfig.tiny(); B = 64
w = np.empty(cfg.weight_count(), np.float32); d = cfg.dims()
L.wm_synth_weights(C.byref(d), 0, w.ctypes.data_as(C.POINTER(C.c_float)))
mels = np.empty((B, 80, 3000), np.float32)
for i in range(B): L.wm_synth_mel_host(1000 + i, 80, 3000, mels[i].ctypes.data_as(C.POINTER(C.c_float)))
mel = torch.from_numpy(mels).cuda()
m = Whisper(cfg, compute_dtype=DT_BF16, kv_dtype=DT_F32, max_batch=B, decoder_fp32=True, coalesce=2); m.load(WeightLoader.from_array(w))
kw = dict(max_loop=99, ignore_eot=True)
for _ in range(2):
    for s in range(8): m.transcribe_submit(mel, slot=s, **kw)
    for s in range(8): m.transcribe_wait(s)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); ts = []
    for s in range(8):
        a = time.perf_counter(); m.transcribe_submit(mel, slot=s, **kw); ts.append((time.perf_counter() - a) * 1e3)
    t1 = time.perf_counter(); tw = []
    for s in range(8):
        a = time.perf_counter(); m.transcribe_wait(s); tw.append((time.perf_counter() - a) * 1e3)
    t2 = time.perf_counter()
    print("submits ms", [round(x, 2) for x in ts], "total", round((t1 - t0) * 1e3, 2), "| waits ms", [round(x, 1) for x in tw], "total", round((t2 - t1) * 1e3, 1), "| group", round((t2 - t0) * 1e3, 1))
