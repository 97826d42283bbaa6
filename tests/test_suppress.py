"""Logit masks: suppress_tokens / begin_suppress_tokens with the semantics of HF generate's
processors.  Fixture: tests/golden/micro_suppress.npz (tools/make_golden_suppress.py, transformers' own processor classes)."""
import numpy as np
import pytest

from conftest import golden


def test_oracle_matches_hf_processors(micro_cfg, micro_weights):
    from oracle import oracle
    from whisper_mojo_amd import synth
    g = golden("micro_suppress")
    mel = synth.synth_mel(micro_cfg, 1000)
    M = oracle.OracleModel(micro_cfg, micro_weights)
    kw = dict(mel=mel, prompt=g["prompt"], eot=-1, max_loop=len(g["plain"]) - 5)
    assert np.array_equal(M.transcribe(**kw), g["plain"])
    assert np.array_equal(M.transcribe(suppress_tokens=g["suppress"], **kw), g["with_suppress"])
    assert np.array_equal(M.transcribe(suppress_tokens=g["suppress"], begin_suppress_tokens=g["begin_suppress"], **kw), g["with_both"])


@pytest.mark.gpu
def test_gpu_masks_in_fused_argmax(micro_cfg, micro_weights):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from whisper_mojo_amd import synth
    from whisper_mojo_amd.loader import WeightLoader
    from whisper_mojo_amd.whisper import Whisper
    g = golden("micro_suppress")
    mels = np.stack([synth.synth_mel(micro_cfg, 1000), synth.synth_mel(micro_cfg, 1001)])
    m = Whisper(micro_cfg, max_batch=2)
    m.load(WeightLoader.from_array(micro_weights))
    kw = dict(prompt=g["prompt"], eot=-1, max_loop=len(g["plain"]) - 5)
    assert m.transcribe_batch(mels, **kw)[0] == g["plain"].tolist()
    s1 = m.transcribe_batch(mels, suppress_tokens=g["suppress"], **kw)
    assert s1[0] == g["with_suppress"].tolist() and not set(s1[1][4:]) & set(g["suppress"].tolist())
    s2 = m.transcribe_batch(mels, suppress_tokens=g["suppress"], begin_suppress_tokens=g["begin_suppress"], **kw)
    assert s2[0] == g["with_both"].tolist()
    assert m.transcribe_batch(mels, **kw)[0] == g["plain"].tolist()  # masks are rebuilt when the lists change back
    # the raw logits of WhisperDecoder.forward are never masked (whisper.mojo:162-166)
    from whisper_mojo_amd.whisper import KVCache
    cache = KVCache(m, 1)
    m.encoder.forward(mels[0], cache)
    lg = m.decoder.forward(g["prompt"].tolist(), None, cache, start_pos=0)
    assert np.isfinite(lg).all() and int(lg.argmax()) == int(g["plain"][4])
