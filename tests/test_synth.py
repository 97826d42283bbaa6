"""The synthetic weight/mel generator is bit-identical in numpy (whisper.mojo_amd/synth.py) and C
(include/wm_synth.h), and produces the reference's file size (151 042 560 B tiny, 290 375 680 B base)."""
import numpy as np

from oracle import oracle
from whisper_mojo_amd import WhisperConfig, synth


def test_file_sizes():
    assert WhisperConfig.tiny().weight_count() * 4 == 151_042_560
    assert WhisperConfig.base().weight_count() * 4 == 290_375_680
    assert len(synth.tensor_table(WhisperConfig.tiny())) == 167
    assert len(synth.tensor_table(WhisperConfig.base())) == 245


def test_float_offsets_match_survey():
    cfg = WhisperConfig.tiny()
    off, offs = 0, {}
    for name, _, shape in synth.tensor_table(cfg):
        offs[name] = off
        off += int(np.prod(shape))
    assert offs["enc.pos"] == 535_296
    assert offs["enc.0.attn.q.w"] == 1_111_296
    assert offs["enc.1.attn.q.w"] - offs["enc.0.attn.q.w"] == 1_774_080
    assert offs["enc.ln.w"] == 8_207_616
    assert offs["dec.tok_emb"] == 8_208_384
    assert offs["dec.pos"] == 28_124_544
    assert offs["dec.0.attn.q.w"] == 28_296_576
    assert offs["dec.1.attn.q.w"] - offs["dec.0.attn.q.w"] == 2_365_824
    assert offs["dec.0.cross.q.w"] - offs["dec.0.attn.q.w"] == 591_744
    assert offs["dec.0.fc1.w"] - offs["dec.0.attn.q.w"] == 1_183_488
    assert offs["dec.ln.w"] == 37_759_872


def test_numpy_equals_c_micro(micro_cfg, micro_weights):
    assert np.array_equal(micro_weights, oracle.synth_weights_c(micro_cfg, 0))
    assert not np.array_equal(micro_weights, oracle.synth_weights_c(micro_cfg, 1))


def test_numpy_equals_c_tiny(tiny_cfg, tiny_weights):
    assert np.array_equal(synth.synth_weights(tiny_cfg, 0), tiny_weights)


def test_mel_equal_and_range(tiny_cfg):
    a, b = synth.synth_mel(tiny_cfg, 1003), oracle.synth_mel_c(tiny_cfg, 1003)
    assert a.shape == (80, 3000) and np.array_equal(a, b)
    assert a.min() == -1.0 and a.max() == 1.5
    assert not np.array_equal(a, synth.synth_mel(tiny_cfg, 1004))
