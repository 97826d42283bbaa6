#!/usr/bin/env python3
"""Developer soak on the round's headline configuration (tiny, B = 64, bf16 encoder, fp32 decoder + KV): N groups of eight
submits — coalesced by the library into four 128-row passes — alternating the fixed-length mode with the reference's stop rule
(loop fed by the library's pump thread, eot reachable for half of the utterances' copies).  Every pass must return exactly the
ids of an uncoalesced synchronous run on the same inputs: a race in the GEMM rings, the decode graphs, the pump thread, the
pairing / demultiplexing or the slot machinery shows up as a differing id.  Args: [groups]"""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from whisper_mojo_amd import WhisperConfig, _lib, DT_BF16, DT_F32
from whisper_mojo_amd.loader import WeightLoader
from whisper_mojo_amd.whisper import Whisper
L = _lib.lib(); cfg = WhisperConfig.tiny(); B = 64
w = np.empty(cfg.weight_count(), np.float32); d = cfg.dims()
L.wm_synth_weights(C.byref(d), 0, w.ctypes.data_as(C.POINTER(C.c_float)))
mels = np.empty((2 * B, 80, 3000), np.float32)
for i in range(2 * B): L.wm_synth_mel_host(1000 + i, 80, 3000, mels[i].ctypes.data_as(C.POINTER(C.c_float)))
mel_dev = [torch.from_numpy(mels[:B]).cuda(), torch.from_numpy(mels[B:]).cuda()]
kw = dict(compute_dtype=DT_BF16, kv_dtype=DT_F32, max_batch=B, decoder_fp32=True)
plain = Whisper(cfg, **kw); plain.load(WeightLoader.from_array(w))
fixed = dict(max_loop=99, ignore_eot=True)
free = plain.transcribe_batch(mel_dev[0], **fixed)
eot = free[0][4 + 40]
natural = dict(max_loop=195, eot=eot)
want = {("f", h): plain.transcribe_batch(mel_dev[h], **fixed) for h in (0, 1)}
want.update({("n", h): plain.transcribe_batch(mel_dev[h], **natural) for h in (0, 1)})
plain.close()
m = Whisper(cfg, coalesce=2, **kw); m.load(WeightLoader.from_array(w))
groups = int(sys.argv[1]) if len(sys.argv) > 1 else 50
bad = 0; t0 = time.time()
for g in range(groups):
    mode, opts = ("n", natural) if g % 4 == 3 else ("f", fixed)
    for s in range(8): m.transcribe_submit(mel_dev[s & 1], slot=s, **opts)
    for s in (range(8) if g % 2 else reversed(range(8))):
        got = m.transcribe_wait(s)
        ref = want[(mode, s & 1)]
        if got != ref:
            bad += 1
            diff = [(b, i) for b in range(B) for i in range(min(len(ref[b]), len(got[b]))) if got[b][i] != ref[b][i]]
            print(f"group {g} slot {s} mode {mode}: {len(diff)} differing ids, first {diff[:4]}")
    if g % 10 == 9: print(f"group {g + 1}/{groups}: {bad} bad passes, {time.time() - t0:.1f} s", flush=True)
print("soak done:", groups * 8, "passes,", bad, "bad")
sys.exit(1 if bad else 0)
