#!/usr/bin/env python3
"""Developer smoke script: op + stage parity vs the oracle with printed errors (not a test)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle
from whisper_mojo_amd import WhisperConfig, synth, whisper_tensor as wt, DT_F32, DT_BF16, DT_F16
from whisper_mojo_amd.loader import WeightLoader
from whisper_mojo_amd.whisper import Whisper, KVCache

rng = np.random.default_rng(0)
for (M, N, K) in [(5, 37, 64), (64, 256, 384), (200, 128, 96), (1, 51865 // 50, 384)]:
    A, B, b = rng.standard_normal((M, K), np.float32), rng.standard_normal((N, K), np.float32), rng.standard_normal(N, np.float32)
    out = np.zeros((M, N), np.float32)
    wt.matmul(out, A, B, b)
    print("matmul", M, N, K, np.abs(out - oracle.matmul(A, B, b)).max())
x = rng.standard_normal((9, 384), np.float32); g = rng.standard_normal(384, np.float32); be = rng.standard_normal(384, np.float32)
o = np.zeros_like(x); wt.layer_norm(o, x, g, be); print("ln", np.abs(o - oracle.layer_norm(x, g, be)).max())
t = rng.standard_normal((3, 67), np.float32); t2 = t.copy(); wt.gelu(t2); print("gelu", np.abs(t2 - oracle.gelu(t)).max())
t = rng.standard_normal((5, 1500), np.float32); t2 = t.copy(); wt.softmax(t2); print("softmax", np.abs(t2 - oracle.softmax(t)).max())
xin = rng.standard_normal((80, 300), np.float32); w = rng.standard_normal((128, 80, 3), np.float32) * 0.1; b = rng.standard_normal(128, np.float32)
for stride, oT in ((1, False), (2, True)):
    Lo = (300 + 2 - 3) // stride + 1
    out = np.zeros((Lo, 128) if oT else (128, Lo), np.float32)
    wt.conv1d(out, xin, w, b, stride, 1, oT)
    print("conv", stride, oT, np.abs(out - oracle.conv1d(xin, oracle.transpose_conv_weights(w), b, stride, 1, oT)).max())
v = np.zeros(1000, np.float32); v[[17, 500]] = 2; print("argmax", wt.argmax(v))

for name, cfg in (("micro", WhisperConfig.micro()), ("tiny", WhisperConfig.tiny())):
    wts = oracle.synth_weights_c(cfg, 0)
    mel = synth.synth_mel(cfg, 1000)
    ref = oracle.OracleModel(cfg, wts)
    enc_ref = ref.encode(mel)
    prompt = (50258, 50259, 50359, 50363) if cfg.vocab_size > 50363 else (1, 2, 3, 4)
    want, wl = ref.transcribe(enc_out=enc_ref, prompt=prompt, eot=-1, max_loop=16, want_logits=True)
    for dt in (DT_F32, DT_BF16, DT_F16):
        m = Whisper(cfg, compute_dtype=dt, max_batch=2)
        m.load(WeightLoader.from_array(wts))
        t0 = time.time(); enc = m.encoder.forward(mel); t1 = time.time() - t0
        print(name, dt, "enc err", np.abs(enc - enc_ref).max(), "rel", np.abs(enc - enc_ref).max() / np.abs(enc_ref).max(), f"{t1*1e3:.1f} ms")
        cache = KVCache(m, 1)
        lg = m.decoder.forward(list(prompt), enc_ref, cache, start_pos=0)
        print("   prefill logits err", np.abs(lg - wl[0]).max(), "argmax", lg.argmax(), wl[0].argmax())
        t0 = time.time(); got = m.transcribe_batch(mel, prompt=prompt, eot=-1, max_loop=16); t1 = time.time() - t0
        print("   tokens equal", got[0] == want.tolist(), f"{t1*1e3:.1f} ms", got[0][:10])
