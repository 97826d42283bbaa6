import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from whisper_mojo_amd import WhisperConfig, _lib, DT_BF16
from whisper_mojo_amd.loader import WeightLoader
from whisper_mojo_amd.whisper import Whisper
L = _lib.lib(); cfg = WhisperConfig.tiny(); B = int(os.environ.get('WM_PHASES_B', '64'))
w = np.empty(cfg.weight_count(), np.float32); d = cfg.dims()
L.wm_synth_weights(C.byref(d), 0, w.ctypes.data_as(C.POINTER(C.c_float)))
mel = np.empty((B, 80, 3000), np.float32)
for i in range(B): L.wm_synth_mel_host(1000 + i, 80, 3000, mel[i].ctypes.data_as(C.POINTER(C.c_float)))
m = Whisper(cfg, compute_dtype=DT_BF16, kv_dtype=(0 if os.environ.get('WM_DEC_F32') else DT_BF16), max_batch=B, decoder_fp32=bool(os.environ.get('WM_DEC_F32'))); m.load(WeightLoader.from_array(w))
st = C.c_void_p(); _lib.check(L.wm_state_new(m._h, B, C.byref(st)))
_lib.check(L.wm_encode(m._h, st, mel.ctypes.data_as(C.c_void_p), 0, B, None))
us = C.c_float()
for which in ((40, 41, 42, 43) if len(sys.argv) < 2 else [int(a) for a in sys.argv[1:]]):
    for _ in range(3):
        _lib.check(L.wm_bench_kernel(m._h, st, which, 1, C.byref(us)))
        print("event us", us.value)
