#!/usr/bin/env python3
"""Developer measurement: wall time of the GPU log-mel front end and of PCM->tokens for 64 clips (host PCM upload included)."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisper_mojo_amd import WhisperConfig, _lib, DT_BF16, frontend
from whisper_mojo_amd.loader import WeightLoader
from whisper_mojo_amd.whisper import Whisper
L = _lib.lib(); cfg = WhisperConfig.tiny(); B = 64
w = np.empty(cfg.weight_count(), np.float32); d = cfg.dims()
L.wm_synth_weights(C.byref(d), 0, w.ctypes.data_as(C.POINTER(C.c_float)))
m = Whisper(cfg, compute_dtype=DT_BF16, max_batch=B); m.load(WeightLoader.from_array(w))
rng = np.random.default_rng(0)
audios = [(rng.standard_normal(480000) * 0.05).astype(np.float32) for _ in range(B)]
for i in range(4):
    t0 = time.perf_counter(); mel = frontend.log_mel(m, audios); t1 = time.perf_counter()
    toks = frontend.transcribe_audio(m, audios, max_loop=99, ignore_eot=True); t2 = time.perf_counter()
    print(f"log_mel (64 x 30 s, incl. 123 MB PCM upload + 61 MB mel download): {1e3*(t1-t0):.2f} ms;  PCM -> tokens: {1e3*(t2-t1):.2f} ms")
