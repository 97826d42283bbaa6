"""Timestamp logit rules: the semantics of HF generate's WhisperTimeStampLogitsProcessor, which the
reference lacks (whisper.mojo:198,219 take the raw argmax).  Fixture: tests/golden/micro_timestamps.npz, produced by
tools/make_golden_timestamps.py with transformers' own processor class — a known-answer table for the rule logic plus
greedy streams of the micro model with the processor (and the suppress processors) applied at every step."""
import numpy as np
import pytest

from conftest import golden


def _ts(g):
    return (int(g["timestamp_begin"]), int(g["no_timestamps"]), int(g["max_initial"]))


def test_oracle_rules_match_hf_processor_table():
    from oracle import oracle
    g = golden("micro_timestamps")
    tb, no_ts, max_init = _ts(g)
    for h, hist in enumerate(g["histories"]):
        seq = [int(t) for t in hist if t >= 0]
        for r in range(g["rows"].shape[1]):
            got = oracle.timestamp_rules(g["rows"][h, r], seq, tb, no_ts, int(g["eos"]), max_init)
            want = g["processed"][h, r]
            assert np.array_equal(np.isneginf(got), np.isneginf(want)), (h, r, seq)
            keep = ~np.isneginf(want)
            assert np.array_equal(got[keep], want[keep]) and keep.any()


def test_oracle_streams_match_hf_processor(micro_cfg, micro_weights):
    from oracle import oracle
    from whisper_mojo_amd import synth
    g = golden("micro_timestamps")
    M = oracle.OracleModel(micro_cfg, micro_weights)
    for i, seed in enumerate(g["mel_seeds"]):
        mel = synth.synth_mel(micro_cfg, int(seed))
        kw = dict(mel=mel, prompt=g["prompt"], eot=int(g["eos"]), ignore_eot=True, max_loop=g["plain"].shape[1] - 5)
        assert np.array_equal(M.transcribe(**kw), g["plain"][i])
        assert np.array_equal(M.transcribe(timestamps=_ts(g), **kw), g["with_rules"][i])
        assert np.array_equal(M.transcribe(timestamps=_ts(g), suppress_tokens=g["suppress"][i], begin_suppress_tokens=g["begin_suppress"][i], **kw),
                              g["with_all"][i])
    ts = g["with_rules"][0][4:]
    tb = int(g["timestamp_begin"])
    assert tb <= ts[0] <= tb + int(g["max_initial"]) and (ts >= tb).sum() >= 3  # the rules are actually exercised


@pytest.mark.gpu
def test_gpu_timestamp_rules_in_fused_argmax(micro_cfg, micro_weights):
    """The device-side greedy loop with the rules inside the fused argmax: ids equal the HF-processor fixture and the oracle."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import oracle
    from whisper_mojo_amd import synth
    from whisper_mojo_amd.loader import WeightLoader
    from whisper_mojo_amd.whisper import Whisper
    g = golden("micro_timestamps")
    seeds = [int(s) for s in g["mel_seeds"]] + [1002, 1003, 1004]
    mels = np.stack([synth.synth_mel(micro_cfg, s) for s in seeds])
    m = Whisper(micro_cfg, max_batch=len(seeds))
    m.load(WeightLoader.from_array(micro_weights))
    kw = dict(prompt=g["prompt"], eot=int(g["eos"]), ignore_eot=True, max_loop=g["plain"].shape[1] - 5)
    assert m.transcribe_batch(mels, **kw)[0] == g["plain"][0].tolist()
    got = m.transcribe_batch(mels, timestamps=_ts(g), **kw)
    for i in range(len(g["mel_seeds"])):
        assert got[i] == g["with_rules"][i].tolist()
    M = oracle.OracleModel(micro_cfg, micro_weights)
    for i in range(len(seeds)):
        assert got[i] == M.transcribe(mel=mels[i], timestamps=_ts(g), **kw).tolist()
    both = m.transcribe_batch(mels[:2], timestamps=_ts(g), suppress_tokens=g["suppress"][0], begin_suppress_tokens=g["begin_suppress"][0], **kw)
    assert both[0] == g["with_all"][0].tolist()
    assert m.transcribe_batch(mels, **kw)[0] == g["plain"][0].tolist()  # rules off again
    # pipelined form carries the rules too
    m.transcribe_submit(mels, slot=1, timestamps=_ts(g), **kw)
    assert m.transcribe_wait(1) == got
    # the reference's stop rule still applies on top: eos is emitted only where the rules allow it, then the utterance stops
    stop = m.transcribe_batch(mels, timestamps=_ts(g), prompt=g["prompt"], eot=int(g["eos"]), max_loop=g["plain"].shape[1] - 5)
    for i in range(len(seeds)):
        want = M.transcribe(mel=mels[i], timestamps=_ts(g), prompt=g["prompt"], eot=int(g["eos"]), max_loop=g["plain"].shape[1] - 5)
        assert stop[i] == want.tolist()
    # round 3: the rules, the suppress lists and the stop rule through COALESCED submits (two batches on one 2·B-row state, the
    # per-utterance rule state TsState for 2·B rows, the loop fed by the pump thread) — each batch still gets exactly its own ids
    c = Whisper(micro_cfg, max_batch=len(seeds), coalesce=2)
    c.load(WeightLoader.from_array(micro_weights))
    rev = mels[::-1].copy()
    kws = dict(timestamps=_ts(g), prompt=g["prompt"], eot=int(g["eos"]), max_loop=g["plain"].shape[1] - 5)
    c.transcribe_submit(mels, slot=0, **kws)
    c.transcribe_submit(rev, slot=1, **kws)
    assert c.transcribe_wait(0) == stop and c.transcribe_wait(1) == stop[::-1]
    kwa = dict(kw, timestamps=_ts(g), suppress_tokens=g["suppress"][0], begin_suppress_tokens=g["begin_suppress"][0])
    c.transcribe_submit(mels[:2], slot=2, **kwa)
    c.transcribe_submit(mels[:2], slot=3, **kwa)
    assert c.transcribe_wait(3) == both and c.transcribe_wait(2) == both
    c.close()


@pytest.mark.gpu
def test_gpu_timestamp_rules_tiny_vocab(tiny_cfg, tiny_weights):
    """Whisper-tiny's real id layout (eot 50257, <|notimestamps|> 50363, 1501 timestamp ids from 50364): three clips, 60 ids,
    against the oracle; every stream opens with a timestamp <= <|1.00|> and never emits <|notimestamps|>."""
    import ctypes as C
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import oracle
    from whisper_mojo_amd import _lib
    from whisper_mojo_amd.loader import WeightLoader
    from whisper_mojo_amd.whisper import Whisper
    L = _lib.lib()
    mels = np.empty((3, 80, 3000), np.float32)
    for i in range(3):
        L.wm_synth_mel_host(1000 + i, 80, 3000, mels[i].ctypes.data_as(C.POINTER(C.c_float)))
    m = Whisper(tiny_cfg, max_batch=3)
    m.load(WeightLoader.from_array(tiny_weights))
    ts = (50364, 50363, 50)
    prompt = (50258, 50259, 50359)  # no <|notimestamps|> in the prompt when timestamps are wanted
    got = m.transcribe_batch(mels, prompt=prompt, timestamps=ts, ignore_eot=True, max_loop=59)
    M = oracle.OracleModel(tiny_cfg, tiny_weights)
    for i in range(3):
        want, lg = M.transcribe(mel=mels[i], prompt=prompt, timestamps=ts, ignore_eot=True, max_loop=59, want_logits=True)
        assert 50364 <= got[i][3] <= 50364 + 50 and 50363 not in got[i][3:]
        d = next((k for k in range(len(want)) if got[i][k] != want[k]), None)
        if d is not None:  # only an fp32 near-tie of the oracle may part the streams (rule 5 compares two sums of ~1500 terms)
            s = np.sort(lg[d - 3])
            assert s[-1] - s[-2] < 1e-3 or abs(float(np.logaddexp.reduce(lg[d - 3][50364:].astype(np.float64))) - float(lg[d - 3][:50364].max())) < 1e-3
