import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from whisper_mojo_amd import WhisperConfig, _lib, DT_BF16
from whisper_mojo_amd.loader import WeightLoader
from whisper_mojo_amd.whisper import Whisper
L = _lib.lib(); cfg = WhisperConfig.tiny(); B = 200
w = np.empty(cfg.weight_count(), np.float32); d = cfg.dims()
L.wm_synth_weights(C.byref(d), 0, w.ctypes.data_as(C.POINTER(C.c_float)))
mels = np.empty((B, 80, 3000), np.float32)
for i in range(B): L.wm_synth_mel_host(1000 + i, 80, 3000, mels[i].ctypes.data_as(C.POINTER(C.c_float)))
m = Whisper(cfg, compute_dtype=DT_BF16, max_batch=B); m.load(WeightLoader.from_array(w))
t0 = time.perf_counter(); out = m.transcribe_batch(mels, max_loop=30, ignore_eot=True); t1 = time.perf_counter()
print("B=200 pass", round((t1 - t0) * 1e3, 1), "ms; lens", set(len(o) for o in out))
for i in (0, 63, 64, 127, 199):
    assert m.transcribe_batch(mels[i], max_loop=30, ignore_eot=True)[0] == out[i], i
print("singles equal")
# natural stop rule at B=64 with an eot that some utterances emit
eot = out[0][10]
t0 = time.perf_counter(); nat = m.transcribe_batch(mels[:64], max_loop=195, eot=eot); t1 = time.perf_counter()
print("natural mode B=64 max_loop=195:", round((t1 - t0) * 1e3, 1), "ms; lengths min/max", min(map(len, nat)), max(map(len, nat)), "ended with eot:", sum(o[-1] == eot for o in nat))
