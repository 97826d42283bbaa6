#!/usr/bin/env python3
"""Developer experiment: per-kernel cost of 40-node graph chains of the small decode kernels (wm_bench_kernel ids 10-16)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisper_mojo_amd import WhisperConfig, _lib, DT_BF16, DT_F32
from whisper_mojo_amd.loader import WeightLoader
from whisper_mojo_amd.whisper import Whisper
L = _lib.lib()
cfg = WhisperConfig.tiny()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
w = np.empty(cfg.weight_count(), np.float32)
d = cfg.dims()
L.wm_synth_weights(C.byref(d), 0, w.ctypes.data_as(C.POINTER(C.c_float)))
DT = DT_F32 if os.environ.get('CHAIN_F32') else DT_BF16
m = Whisper(cfg, compute_dtype=DT, max_batch=B)
m.load(WeightLoader.from_array(w))
st = C.c_void_p()
_lib.check(L.wm_state_new(m._h, B, C.byref(st)))
mel = np.zeros((B, 80, 3000), np.float32)
_lib.check(L.wm_encode(m._h, st, mel.ctypes.data_as(C.c_void_p), 0, B, None))
names = {20: "L: LN1+QKV+append", 21: "L: self-attn (len 61)", 22: "L: o-proj+res", 23: "L: LNx+q", 24: "L: cross-attn", 25: "L: combine",
         26: "L: LN2+fc1+gelu", 27: "L: fc2+res", 28: "final LN+logits+argmax1", 29: "argmax2", 10: "embed x40", 11: "set_step x40", 12: "combine x40", 13: "dec_linear(no LN) x40", 14: "dec_linear(LN) x40",
         15: "embed/combine/set_step mix", 16: "dec_linear(LN)/combine alternating"}
ONLY = [int(a) for a in sys.argv[2:]]
for k, n in names.items():
    if ONLY and k not in ONLY:
        continue
    us = C.c_float()
    _lib.check(L.wm_bench_kernel(m._h, st, k, 50, C.byref(us)))
    print(f"{n:40s} {us.value:6.2f} us/kernel")
