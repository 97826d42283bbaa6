"""Log-mel front end.  CPU: the numpy restatement vs the fixture generated from the transformers
WhisperFeatureExtractor the reference delegates to (export_weights.py:100-116).  GPU: the HIP front end vs both.
Tolerance on the normalised log-mel (range about [-1, 1.1]): the reference pipeline runs the FFT in float64; the GPU runs
an exact-fp32 MFMA DFT, so bins far below a frame's peak carry fp32 round-off — 2e-3 absolute, 1e-4 on the mean."""
import numpy as np
import pytest

from conftest import golden


def _cases():
    from oracle import logmel_oracle as lo
    g = golden("logmel")
    for i in range(4):
        audio = lo.synth_audio(int(g[f"seed{i}"]), int(g[f"n{i}"]))
        yield i, audio, g


def test_oracle_matches_hf_fixture():
    from oracle import logmel_oracle as lo
    g = golden("logmel")
    fb = lo.mel_filter_bank()
    assert np.abs(fb.sum(0) - g["mel_filters_colsum"]).max() < 1e-6
    assert np.abs(fb[::10, ::8] - g["mel_filters_sample"]).max() < 1e-7
    for i, audio, g in _cases():
        mel = lo.log_mel(audio)
        assert mel.shape == (80, 3000) and mel.dtype == np.float32
        assert np.abs(mel[:, g["cols"]] - g[f"mel{i}_cols"]).max() < 5e-5
        assert np.abs(mel.astype(np.float64).sum(1) - g[f"mel{i}_rowsum"]).max() < 5e-2
        assert np.abs(mel.astype(np.float64).sum(0) - g[f"mel{i}_colsum"]).max() < 2e-3


def test_synth_audio_is_reproducible():
    from oracle import logmel_oracle as lo
    a = lo.synth_audio(5, 20000)
    assert a.dtype == np.float32 and np.array_equal(a, lo.synth_audio(5, 20000)) and not np.array_equal(a, lo.synth_audio(6, 20000))
    assert np.abs(a).max() < 1.0
    assert np.allclose(np.abs(a[8000:12000]), 0.02)  # third 0.25 s segment: no noise, only the square tone
    assert (a[12000:16000] != 0).any() and np.abs(a[4000:8000]).max() < np.abs(a[0:4000]).max()


@pytest.mark.gpu
def test_gpu_front_end_matches_oracle_and_fixture(tiny_cfg, tiny_weights):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import logmel_oracle as lo
    from whisper_mojo_amd import frontend
    from whisper_mojo_amd.loader import WeightLoader
    from whisper_mojo_amd.whisper import Whisper
    m = Whisper(tiny_cfg, max_batch=4)
    m.load(WeightLoader.from_array(tiny_weights))
    audios = [a for _, a, _ in _cases()]
    g = golden("logmel")
    mels = frontend.log_mel(m, audios)  # one ragged batch: full, short, over-long, tiny
    assert mels.shape == (4, 80, 3000)
    for i, a in enumerate(audios):
        ref = lo.log_mel(a)
        err = np.abs(mels[i] - ref)
        assert err.max() < 2e-3 and err.mean() < 1e-4, (i, err.max(), err.mean())
        assert np.abs(mels[i][:, g["cols"]] - g[f"mel{i}_cols"]).max() < 2e-3
        assert np.array_equal(frontend.log_mel(m, [a])[0], mels[i])  # batch invariance
    # PCM -> tokens == mel -> tokens
    want = m.transcribe_batch(mels[:2], max_loop=8, ignore_eot=True)
    got = frontend.transcribe_audio(m, audios[:2], max_loop=8, ignore_eot=True)
    assert got == want


@pytest.mark.gpu
def test_gpu_front_end_micro_config(micro_cfg, micro_weights):
    """Other window lengths / mel counts (micro: 16 mels, 200 frames = 2 s)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import logmel_oracle as lo
    from whisper_mojo_amd import frontend
    from whisper_mojo_amd.loader import WeightLoader
    from whisper_mojo_amd.whisper import Whisper
    m = Whisper(micro_cfg, max_batch=2)
    m.load(WeightLoader.from_array(micro_weights))
    audios = [lo.synth_audio(9, 32000), lo.synth_audio(10, 5000)]
    mels = frontend.log_mel(m, audios)
    for i, a in enumerate(audios):
        ref = lo.log_mel(a, n_frames=micro_cfg.n_frames, n_mels=micro_cfg.n_mels)
        assert np.abs(mels[i] - ref).max() < 2e-3


@pytest.mark.gpu
def test_long_form_and_language_id(micro_cfg, micro_weights):
    """§8f rank 4, host level: long audio = consecutive full windows through the batch path (ids equal per-window
    transcription); language id = one decoder step on the start token, softmax over a range of ids (checked against the
    oracle's logits for the same step)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import logmel_oracle as lo, oracle
    from whisper_mojo_amd import frontend
    from whisper_mojo_amd.loader import WeightLoader
    from whisper_mojo_amd.whisper import Whisper
    m = Whisper(micro_cfg, max_batch=2)
    m.load(WeightLoader.from_array(micro_weights))
    win = micro_cfg.n_frames * frontend.HOP
    audio = lo.synth_audio(21, 3 * win + 5000)  # 3 full windows + a short tail -> 4 windows, 2 batches of 2
    kw = dict(prompt=(1, 2, 3, 4), eot=-1, max_loop=6)
    per_window, flat = frontend.transcribe_long(m, audio, **kw)
    assert len(per_window) == 4
    for i, ids in enumerate(per_window):
        assert ids == frontend.transcribe_audio(m, [audio[i * win:(i + 1) * win]], **kw)[0]
    assert flat == [t for ids in per_window for t in ids[4:]]
    # language id on the micro vocabulary: "languages" = ids 100..131
    mel = frontend.log_mel(m, [audio[:win]])[0]
    lid, p = frontend.detect_language(m, mel, sot=7, lang_first=100, lang_last=131)
    ref = oracle.OracleModel(micro_cfg, micro_weights)
    _, logits = ref.transcribe(mel=mel, prompt=(7,), eot=-1, max_loop=0, want_logits=True)
    rl = logits[0][100:132].astype(np.float64)
    rp = np.exp(rl - rl.max()); rp /= rp.sum()
    assert lid == 100 + int(rp.argmax()) and abs(p.sum() - 1) < 1e-9
    assert np.abs(p - rp).max() < 1e-4
    ids, pb = frontend.detect_language(m, np.stack([mel, mel]), sot=7, lang_first=100, lang_last=131)
    assert ids.tolist() == [lid, lid] and np.allclose(pb[0], p) and np.allclose(pb[1], p)
