import time, sys, os, ctypes as C
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from whisper_mojo_amd import WhisperConfig, _lib, DT_BF16, dist as wdist
from whisper_mojo_amd.loader import WeightLoader
from whisper_mojo_amd.whisper import Whisper
L = _lib.lib(); cfg = WhisperConfig.tiny(); B = 64
w = np.empty(cfg.weight_count(), np.float32); d = cfg.dims()
L.wm_synth_weights(C.byref(d), 0, w.ctypes.data_as(C.POINTER(C.c_float)))
m = Whisper(cfg, compute_dtype=DT_BF16, max_batch=B); m.load(WeightLoader.from_array(w))
mel = torch.randn(B, 80, 3000, device="cuda").clamp(-1, 1.5)
for i in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = m.transcribe_batch(mel, max_loop=99, ignore_eot=True)
    t1 = time.perf_counter()
    g = wdist.gather_tokens(m.last_tokens, m.last_counts, B, 104)
    t2 = time.perf_counter()
    print(f"transcribe_batch {1e3*(t1-t0):.2f} ms, gather {1e3*(t2-t1):.3f} ms")
