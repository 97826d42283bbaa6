#!/usr/bin/env python3
"""Developer check: device memory is stable across many passes, state re-creations (changing B) and model reloads."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisper_mojo_amd import WhisperConfig, synth, DT_BF16
from whisper_mojo_amd.loader import WeightLoader
from whisper_mojo_amd.whisper import Whisper

def free_mb():
    torch.cuda.synchronize()
    f, t = torch.cuda.mem_get_info()
    return f / 2**20

cfg = WhisperConfig.micro()
w = synth.synth_weights(cfg, 0)
mels = synth.synth_mels(cfg, 0, 8)
torch.zeros(1, device="cuda")
base = free_mb()
for rnd in range(3):
    m = Whisper(cfg, compute_dtype=DT_BF16, max_batch=8)
    m.load(WeightLoader.from_array(w))
    marks = []
    for it in range(60):
        B = 1 + it % 8
        m.transcribe_batch(mels[:B], prompt=(1, 2, 3, 4), eot=-1, max_loop=5)   # B changes -> slot 0's state is re-created
        if it % 20 == 19:
            for s in range(8):
                m.transcribe_submit(mels[:3], slot=s, prompt=(1, 2, 3, 4), eot=-1, max_loop=5)
            for s in range(8):
                m.transcribe_wait(s)
            marks.append(round(free_mb(), 1))
    m.close()
    print(f"round {rnd}: free MB during {marks}, after close {free_mb():.1f} (start {base:.1f})")
