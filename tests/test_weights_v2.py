"""Weight file format v2: host-side converter / reader (CPU) and loading through wm_model_load (GPU)."""
import ctypes as C
import os

import numpy as np
import pytest

from whisper_mojo_amd import _lib, synth


def _convert(cfg, w, tmp_path, dtype, emb_f32=1):
    v1, v2 = str(tmp_path / "w.bin"), str(tmp_path / f"w{dtype}{emb_f32}.wmi2")
    w.tofile(v1)
    d = cfg.dims()
    _lib.check(_lib.lib().wm_weights_convert_v2(v1.encode(), v2.encode(), C.byref(d), dtype, emb_f32))
    return v1, v2


def _read(cfg, path):
    out = np.empty(cfg.weight_count(), np.float32)
    d = cfg.dims()
    _lib.check(_lib.lib().wm_weights_read(path.encode(), C.byref(d), out.ctypes.data_as(C.POINTER(C.c_float))))
    return out


def test_roundtrip_and_sizes(micro_cfg, micro_weights, tmp_path):
    import torch
    v1, v2 = _convert(micro_cfg, micro_weights, tmp_path, 0)
    assert os.path.getsize(v2) == 64 + micro_weights.nbytes and open(v2, "rb").read(8) == b"WMIWGT2\0"
    assert np.array_equal(_read(micro_cfg, v2), micro_weights) and np.array_equal(_read(micro_cfg, v1), micro_weights)
    _, vb = _convert(micro_cfg, micro_weights, tmp_path, 1)
    back = _read(micro_cfg, vb)
    parts, ref = synth.split_weights(micro_cfg, back), synth.split_weights(micro_cfg, micro_weights)
    n_mat = 0
    for name, kind, shape in synth.tensor_table(micro_cfg):
        if kind in (synth.K_WEIGHT, synth.K_QK):
            n_mat += int(np.prod(shape))
            want = torch.from_numpy(ref[name].copy()).bfloat16().float().numpy()   # round-to-nearest-even
            assert np.array_equal(parts[name], want), name
        else:
            assert np.array_equal(parts[name], ref[name]), name                    # vectors, positions, embedding stay fp32
    assert os.path.getsize(vb) == 64 + micro_weights.nbytes - 2 * n_mat
    _, ve = _convert(micro_cfg, micro_weights, tmp_path, 2, emb_f32=0)
    assert os.path.getsize(ve) == os.path.getsize(vb) - 2 * micro_cfg.vocab_size * micro_cfg.d_model


def test_validation(micro_cfg, micro_weights, tiny_cfg, tmp_path):
    v1, v2 = _convert(micro_cfg, micro_weights, tmp_path, 1)
    with pytest.raises(_lib.WhisperMiError, match="other dims"):
        _read(tiny_cfg, v2)
    open(v2, "ab").write(b"\0")
    with pytest.raises(_lib.WhisperMiError, match="payload"):
        _read(micro_cfg, v2)
    micro_weights[:-1].tofile(v1)
    with pytest.raises(_lib.WhisperMiError, match="bytes"):
        _read(micro_cfg, v1)
    with pytest.raises(_lib.WhisperMiError, match="cannot open"):
        _read(micro_cfg, str(tmp_path / "missing"))


@pytest.mark.gpu
def test_gpu_v2_loads_to_the_same_model(micro_cfg, micro_weights, tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from whisper_mojo_amd.whisper import Whisper
    v1, v2 = _convert(micro_cfg, micro_weights, tmp_path, 1)
    mels = synth.synth_mels(micro_cfg, 0, 2)
    outs = []
    for path in (v1, v2):
        m = Whisper(micro_cfg, compute_dtype=1, max_batch=2)
        m.load_file(path)
        outs.append((m.transcribe_batch(mels, prompt=(1, 2, 3, 4), eot=-1, max_loop=12), m.encoder.forward(mels)))
    assert outs[0][0] == outs[1][0] and np.array_equal(outs[0][1], outs[1][1])
