"""Long and wide oracle-parity tests (-m gpu): the HIP path against the CPU oracle at the reference's full loop length
(195 iterations, whisper.mojo:205), up to the 448-row context edge (whisper.mojo:193), for several utterances in one batch, and
at the BENCHED configurations (config 3: tiny, B = 64, bf16 operands + bf16 KV, 100 positions; config 5: base, B = 64, f16).

16-bit modes cannot be token-exact against an fp32 oracle: they are teacher-forced on the ORACLE's greedy stream, and the
tests assert (1) a bound on the logit error at every position, (2) top-1 agreement wherever the oracle's top1-top2 margin
exceeds 4x that bound, with a FLOOR on how many positions that is (the check cannot go vacuous), and (3) that the device-side
greedy loop (fused argmax, graph replay) reproduces the argmax of those teacher-forced logits id for id up to the first
position where it leaves the oracle's stream.  Nothing here reads /root/reference."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

F32_TOL = 5e-5      # fp32 mode: |logit - oracle| (values O(1); different summation order)
MARGIN_TOL = 1e-3   # fp32 mode: a token may differ from the oracle's only where the oracle's own margin is below this
BF16_TOL = 0.06     # bf16 operands + bf16 KV over 100 positions (measured: 0.031 max over 8 utterances x 100 positions in round 2)
F16_TOL = 0.012     # f16 operands + f16 KV, base dims (measured: 0.0057 over 2 utterances x 40 positions)
ENC16_TOL = 0.03    # bf16 ENCODER GEMMs only, decoder + KV fp32 (BASELINE config 3 as written): measured 0.0162 over 64 utterances x 100 positions
PROMPT = (50258, 50259, 50359, 50363)


@pytest.fixture(scope="module")
def hip():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from whisper_mojo_amd import _lib
    _lib.lib()  # raises if the HIP library is missing: no fallback
    return True


@pytest.fixture(scope="module")
def oracle_mod():
    from oracle import oracle
    return oracle


def make_model(cfg, weights, **kw):
    from whisper_mojo_amd.loader import WeightLoader
    from whisper_mojo_amd.whisper import Whisper
    m = Whisper(cfg, **kw)
    m.load(WeightLoader.from_array(weights))
    return m


def synth_mels(cfg, seeds):
    from whisper_mojo_amd import _lib
    L = _lib.lib()
    out = np.empty((len(seeds), cfg.n_mels, cfg.n_frames), np.float32)
    for i, s in enumerate(seeds):
        L.wm_synth_mel_host(int(s), cfg.n_mels, cfg.n_frames, out[i].ctypes.data_as(C.POINTER(C.c_float)))
    return out


def first_divergence(got, want):
    n = min(len(got), len(want))
    for i in range(n):
        if got[i] != want[i]:
            return i
    return None if len(got) == len(want) else n


def margins_of(logits):
    s = np.partition(logits, -2, axis=1)
    return s[:, -1] - s[:, -2]


def teacher_forced(model, cache, streams, n_prompt, pos_mode=0):
    """WhisperDecoder.forward over `streams` [B, n] (prompt block first, then one token at a time, positions owned by this
    loop as in whisper.mojo:195,217).  Returns logits [n - n_prompt + 1, B, vocab]: row i predicts token n_prompt + i."""
    rows = [model.decoder.forward(streams[:, :n_prompt], None, cache, start_pos=0)]
    for i in range(n_prompt, streams.shape[1]):
        sp = cache.current_len - 1 if pos_mode == 0 else cache.current_len
        rows.append(model.decoder.forward(streams[:, i:i + 1], None, cache, start_pos=sp))
    return np.stack(rows)


# ------------------------------------------------------------------------------------------------ (a) tiny fp32, 195 steps
def test_tiny_fp32_195_steps_five_utterances(hip, oracle_mod, tiny_cfg, tiny_weights):
    """Config 2 arithmetic at the reference's full loop bound: five different clips in one batch, max_loop = 195 (whisper.mojo:205),
    cache lengths 4..199.  ids equal the oracle's (a difference is tolerated only at an oracle near-tie), and the logits of
    every one of the 196 positions, teacher-forced on the oracle's stream, are within 5e-5."""
    from whisper_mojo_amd.whisper import KVCache
    seeds = [1000, 1001, 1002, 1003, 1017]
    mels = synth_mels(tiny_cfg, seeds)
    ref = oracle_mod.OracleModel(tiny_cfg, tiny_weights)
    m = make_model(tiny_cfg, tiny_weights, max_batch=len(seeds))
    got = m.transcribe_batch(mels, max_loop=195, ignore_eot=True)
    want, wlog = [], []
    for b in range(len(seeds)):
        t, lg = ref.transcribe(mel=mels[b], max_loop=195, ignore_eot=True, want_logits=True)
        want.append(t)
        wlog.append(lg)
    for b in range(len(seeds)):
        assert len(got[b]) == len(want[b]) == 4 + 1 + 195
        i = first_divergence(got[b], want[b].tolist())
        if i is not None:
            mg = margins_of(wlog[b][i - 4:i - 3])[0]
            assert mg < MARGIN_TOL, f"utterance {b} token {i}: HIP {got[b][i]} vs oracle {want[b][i]}, oracle margin {mg}"
    # per-position logits on the oracle's streams (the oracle's free-running logits ARE teacher-forced on its own stream)
    streams = np.stack(want).astype(np.int32)[:, :-1]  # the last generated id is never fed back
    cache = KVCache(m, len(seeds))
    m.encoder.forward(mels, cache)
    lg = teacher_forced(m, cache, streams, 4)
    assert lg.shape[0] == 196 and cache.current_len == 199
    err = max(np.abs(lg[:, b] - wlog[b]).max() for b in range(len(seeds)))
    print(f"tiny fp32, 5 clips x 196 positions x 51 865 logits (split-bf16 logits kernel): max |logit - oracle| = {err:.2e}")
    assert err < F32_TOL, err
    for b in range(len(seeds)):
        clear = margins_of(wlog[b]) > MARGIN_TOL
        assert clear.sum() >= 190
        assert np.array_equal(lg[:, b].argmax(1)[clear], want[b][4:][clear])


def test_tiny_fp32_reachable_eot_long(hip, oracle_mod, tiny_cfg, tiny_weights):
    """The reference's stop rule (whisper.mojo:206-221) deep into the loop: eot is chosen so that one utterance stops after
    ~120 iterations, another later or never; every utterance's list equals the oracle's, eot included, and the others keep
    going to the bound."""
    seeds = [1000, 1001, 1002, 1003]
    mels = synth_mels(tiny_cfg, seeds)
    ref = oracle_mod.OracleModel(tiny_cfg, tiny_weights)
    m = make_model(tiny_cfg, tiny_weights, max_batch=len(seeds))
    free = ref.transcribe(mel=mels[0], max_loop=195, ignore_eot=True)
    eot = int(free[4 + 120])
    got = m.transcribe_batch(mels, eot=eot, max_loop=195)
    stopped = 0
    for b in range(len(seeds)):
        want, lg = ref.transcribe(mel=mels[b], eot=eot, max_loop=195, want_logits=True)
        want = want.tolist()
        i = first_divergence(got[b], want)
        if i is not None:
            # the streams part at an oracle near-tie: everything BEFORE it is still checked — same ids, and no stop before the
            # oracle stops (an eot in the common prefix would have ended both lists there)
            assert margins_of(lg[i - 4:i - 3])[0] < MARGIN_TOL, (b, i, got[b][i], want[i])
            assert got[b][:i] == want[:i] and eot not in got[b][4:i]
            # after the parting the HIP list still obeys the stop rule on its own ids
            assert got[b].count(eot) <= 1 and (eot not in got[b] or got[b][-1] == eot)
            assert len(got[b]) <= 4 + 1 + 195
            continue
        assert got[b] == want
        if want[-1] == eot and len(want) < 200:
            stopped += 1
            assert got[b][-1] == eot and got[b].count(eot) == 1
        else:
            assert len(got[b]) == 4 + 1 + 195 and eot not in got[b][4:]
    assert stopped >= 1 and len(got[0]) <= 4 + 1 + 120 + 1


def test_early_exit_sync_and_pipelined(hip, oracle_mod, tiny_cfg, tiny_weights):
    """The reference breaks its loop at eot (whisper.mojo:206-207).  Batched: the loop ends once EVERY utterance has emitted eot.
    Four copies of two clips, eot := the id clip 0 emits at iteration 60 and clip 1 somewhere else or never — (a) a batch that can
    finish (copies of clip 0 only): synchronous and pipelined entries return the oracle's ids and stop enqueueing within two
    sub-chunks (16 steps, + the one in progress) of the stop; (b) a batch in which one utterance never stops runs to the bound.
    The ids never depend on where the loop was cut."""
    mels = synth_mels(tiny_cfg, [1000, 1001])
    ref = oracle_mod.OracleModel(tiny_cfg, tiny_weights)
    free0 = ref.transcribe(mel=mels[0], max_loop=195, ignore_eot=True)
    eot = int(free0[4 + 60])
    want0 = ref.transcribe(mel=mels[0], eot=eot, max_loop=195).tolist()
    want1 = ref.transcribe(mel=mels[1], eot=eot, max_loop=195).tolist()
    assert want0[-1] == eot and len(want0) <= 4 + 1 + 60 + 1
    stop_iter = len(want0) - 5  # loop iterations until clip 0's eot is emitted
    m = make_model(tiny_cfg, tiny_weights, max_batch=4)
    same = np.stack([mels[0]] * 4)
    got = m.transcribe_batch(same, eot=eot, max_loop=195)  # synchronous entry
    assert got == [want0] * 4
    assert stop_iter <= m.loop_steps(0) <= stop_iter + 16 + 8 + 1, (stop_iter, m.loop_steps(0))
    for slot in (0, 1, 2):  # pipelined entry, three passes in flight: the library's pump thread feeds all of them
        m.transcribe_submit(same, slot=slot, eot=eot, max_loop=195)
    for slot in (0, 1, 2):
        assert m.transcribe_wait(slot) == [want0] * 4
        assert stop_iter <= m.loop_steps(slot) <= stop_iter + 16 + 8 + 1, (slot, stop_iter, m.loop_steps(slot))
    mixed = np.stack([mels[0], mels[1], mels[0], mels[1]])
    m.transcribe_submit(mixed, slot=3, eot=eot, max_loop=195)
    got = m.transcribe_wait(3)
    i = first_divergence(got[1], want1)
    if i is None:
        assert got == [want0, want1, want0, want1]
        long1 = len(want1) - 5
        assert min(195, long1) <= m.loop_steps(3) <= min(195, long1 + 25)
    else:  # clip 1 parted from the oracle at a near-tie: clip 0's rows are still exact
        assert got[0] == want0 and got[2] == want0
    # fixed mode is enqueued whole
    m.transcribe_batch(same, eot=eot, max_loop=40, ignore_eot=True)
    assert m.loop_steps(0) == 40
    m.close()


def test_tiny_fp32_context_edge_448(hip, oracle_mod, tiny_cfg, tiny_weights):
    """The 448-row decoder context (KVCache(n_layers, d_model, 448), whisper.mojo:193): the longest stream the cache holds —
    the 4 prompt rows + 444 fed-back ids fill rows 0..447 (the last generated id needs no row), 449 ids in all.  ids equal
    the oracle's up to an oracle near-tie; one iteration more is refused."""
    mel = synth_mels(tiny_cfg, [1005])
    ref = oracle_mod.OracleModel(tiny_cfg, tiny_weights)
    m = make_model(tiny_cfg, tiny_weights, max_batch=1)
    n = tiny_cfg.n_text_ctx - 4
    want, lg = ref.transcribe(mel=mel[0], max_loop=n, ignore_eot=True, want_logits=True)
    got = m.transcribe_batch(mel, max_loop=n, ignore_eot=True)[0]
    assert len(got) == len(want) == tiny_cfg.n_text_ctx + 1
    i = first_divergence(got, want.tolist())
    assert i is None or margins_of(lg[i - 4:i - 3])[0] < MARGIN_TOL, (i, got[i], want[i])
    from whisper_mojo_amd import _lib
    with pytest.raises(_lib.WhisperMiError, match="context"):
        m.transcribe_batch(mel, max_loop=n + 1, ignore_eot=True)


def test_micro_longest_stream_ids(hip, oracle_mod, micro_cfg, micro_weights):
    """The same context edge on the micro model (n_text_ctx = 64), three clips, ids against the oracle."""
    from whisper_mojo_amd import synth
    mels = synth.synth_mels(micro_cfg, 200, 3)
    ref = oracle_mod.OracleModel(micro_cfg, micro_weights)
    m = make_model(micro_cfg, micro_weights, max_batch=3)
    n = micro_cfg.n_text_ctx - 4
    got = m.transcribe_batch(mels, prompt=(1, 2, 3, 4), eot=-1, max_loop=n)
    for b in range(3):
        want, lg = ref.transcribe(mel=mels[b], prompt=(1, 2, 3, 4), eot=-1, max_loop=n, want_logits=True)
        assert len(got[b]) == micro_cfg.n_text_ctx + 1
        i = first_divergence(got[b], want.tolist())
        assert i is None or margins_of(lg[i - 4:i - 3])[0] < MARGIN_TOL


# ------------------------------------------------------------------------------------------------ (b) config 3 as benched
_ORACLE_RUNS = {}  # (d_model, n_layers, utterance, steps) -> (ids, logits) of the fp32 oracle on seed-0 weights, mel seed 1000 + utterance


def _sixteen_bit_case(oracle_mod, cfg, weights, dtype, tol, sampled, positions, min_clear, decoder_fp32=False, alone=None):
    """B = 64 in a 16-bit mode against the oracle for the `sampled` utterances (seeds 1000 + u, as bench.py).  decoder_fp32: the
    16-bit dtype applies to the encoder GEMMs only; decoder weights / operands / KV cache fp32 (BASELINE config 3 as written)."""
    from whisper_mojo_amd.whisper import KVCache
    B = 64
    mels = synth_mels(cfg, [1000 + u for u in range(B)])
    ref = oracle_mod.OracleModel(cfg, weights)
    steps = positions - 1  # 1 id from the prefill + `steps` loop iterations
    want, wlog = {}, {}
    for u in sampled:  # the oracle's streams / logits depend on (model dims, utterance, length) only: shared between the B = 64 tests
        key = (cfg.d_model, cfg.n_layers, u, steps)
        if key not in _ORACLE_RUNS:
            _ORACLE_RUNS[key] = ref.transcribe(mel=mels[u], max_loop=steps, ignore_eot=True, want_logits=True)
        want[u], wlog[u] = _ORACLE_RUNS[key]
    m = make_model(cfg, weights, compute_dtype=dtype, kv_dtype=0 if decoder_fp32 else dtype, max_batch=B, decoder_fp32=decoder_fp32)
    # teacher-forced on the oracle's streams: utterance u decodes the stream of sampled[u % len]; only the sampled rows are compared
    streams = np.stack([want[sampled[u % len(sampled)]] if u not in want else want[u] for u in range(B)]).astype(np.int32)[:, :-1]
    cache = KVCache(m, B)
    m.encoder.forward(mels, cache)
    lg = teacher_forced(m, cache, streams, 4)  # [positions, B, vocab]
    assert lg.shape[0] == positions
    n_clear, worst = 0, 0.0
    for u in sampled:
        e = np.abs(lg[:, u] - wlog[u]).max()
        worst = max(worst, float(e))
        clear = margins_of(wlog[u]) > 4 * tol
        n_clear += int(clear.sum())
        assert np.array_equal(lg[:, u].argmax(1)[clear], want[u][4:][clear]), f"utterance {u}: top-1 differs at a clear margin"
    assert worst < tol, worst
    assert n_clear >= min_clear, n_clear  # the agreement check above covered at least this many positions
    # the greedy loop as benched (graph replay, fused argmax, token feedback on the device): same ids as the argmax of the
    # teacher-forced logits, position for position, as long as it is still on the oracle's stream
    got = m.transcribe_batch(mels, max_loop=steps, ignore_eot=True)
    again = m.transcribe_batch(mels, max_loop=steps, ignore_eot=True)
    assert got == again
    for u in sampled:
        assert len(got[u]) == 4 + positions
        tf_ids = lg[:, u].argmax(1)
        i = first_divergence(got[u], want[u].tolist())
        upto = positions if i is None else i - 4 + 1
        assert got[u][4:4 + upto] == tf_ids[:upto].tolist()
    for u in (alone or (sampled[0], sampled[-1])):  # batch of 64 == the utterance alone, bit for bit
        assert m.transcribe_batch(mels[u], max_loop=steps, ignore_eot=True)[0] == got[u]
    assert all(0 <= t < cfg.vocab_size for row in got for t in row)
    return worst, n_clear


def test_config3_tiny_b64_bf16_against_oracle(hip, oracle_mod, tiny_cfg, tiny_weights):
    """The all-16-bit variant of config 3 (`value_all_16bit`, round 2's headline): 64 clips, bf16 operands + bf16 KV cache,
    1 prefill + 99 steps — all 64 utterances against the oracle (round 2 sampled 8)."""
    worst, n_clear = _sixteen_bit_case(oracle_mod, tiny_cfg, tiny_weights, 1, BF16_TOL, list(range(64)), 100, 5600, alone=(0, 31, 63))  # measured 0.0327, 6 019 clear
    print(f"all-16-bit tiny B=64: max |logit error| {worst:.4f} over 64 x 100 positions, {n_clear} positions with a clear margin")


def test_config3_literal_bf16_encoder_fp32_decoder_all_64_clips(hip, oracle_mod, tiny_cfg, tiny_weights):
    """BASELINE config 3 AS WRITTEN and as `python bench.py` runs it by default (workload tiny_b64_bf16enc_f32dec): 64 clips,
    bf16 encoder GEMMs on MFMA, decoder weights / operands / KV cache fp32, 1 prefill + 99 steps.  ALL 64 utterances against the
    oracle, 100 positions each, teacher-forced on the oracle's own greedy streams."""
    worst, n_clear = _sixteen_bit_case(oracle_mod, tiny_cfg, tiny_weights, 1, ENC16_TOL, list(range(64)), 100, 5900,
                                       decoder_fp32=True, alone=(0, 31, 63))  # measured 0.0162, 6 259 clear
    print(f"config 3 as written: max |logit error| {worst:.4f} over 64 x 100 positions, {n_clear} positions with a clear margin")


def test_config5_base_b64_f16_against_oracle(hip, oracle_mod):
    """BASELINE config 5: Whisper-base dims, 64 clips, f16 operands + f16 KV cache in HBM; 100 positions, eight sampled clips."""
    from whisper_mojo_amd import WhisperConfig
    cfg = WhisperConfig.base()
    w = oracle_mod.synth_weights_c(cfg, 0)
    worst, n_clear = _sixteen_bit_case(oracle_mod, cfg, w, 2, F16_TOL, [0, 3, 9, 17, 31, 42, 60, 63], 100, 740)  # measured 0.0062, 790 clear
    print(f"config 5: max |logit error| {worst:.4f} over 8 x 100 positions, {n_clear} positions with a clear margin")


# ------------------------------------------------------------------------------------------------ robustness
def test_resubmit_on_busy_slot_is_refused_for_any_batch_size(hip, micro_cfg, micro_weights):
    """A slot that holds an un-waited pass must refuse a new submit BEFORE touching its state, also when the batch size
    differs (re-creating the state would free graphs and arenas under running kernels); the first pass still completes."""
    from whisper_mojo_amd import _lib, synth
    m = make_model(micro_cfg, micro_weights, max_batch=2)
    a = synth.synth_mels(micro_cfg, 0, 2)
    kw = dict(prompt=(1, 2, 3, 4), eot=-1, max_loop=20)
    want = m.transcribe_batch(a, **kw)
    for slot in (0, 3):
        m.transcribe_submit(a, slot=slot, **kw)
        keep = m._pending[slot]
        with pytest.raises(_lib.WhisperMiError, match="not waited"):
            m.transcribe_submit(a[:1], slot=slot, **kw)
        m._pending[slot] = keep
        assert m.transcribe_wait(slot) == want
    m.transcribe_submit(a, slot=0, **kw)
    with pytest.raises(_lib.WhisperMiError, match="not waited"):
        m.transcribe_batch(a[:1], **kw)  # wm_transcribe shares slot 0
    assert m.transcribe_wait(0) == want
    m.transcribe_submit(a, slot=1, **kw)
    m.close()  # frees a state with a pass in flight: must synchronise first, not fault


def test_unsupported_dims_are_rejected(hip):
    """Dims the decode kernels are not instantiated for must fail at load, not compute garbage."""
    from whisper_mojo_amd import WhisperConfig, _lib, synth
    from whisper_mojo_amd.loader import WeightLoader
    from whisper_mojo_amd.whisper import Whisper
    for cfg, what in ((WhisperConfig(256, 4, 1, 300, 512, 16, 50, 32), "d_model"),
                      (WhisperConfig(128, 2, 1, 300, 2176, 16, 50, 32), "ffn"),
                      (WhisperConfig(128, 2, 1, 300, 2432, 16, 50, 32), "ffn")):
        w = synth.synth_weights(cfg, 0)
        m = Whisper(cfg, max_batch=1)
        with pytest.raises(_lib.WhisperMiError, match=what):
            m.load(WeightLoader.from_array(w))


def test_all_masked_and_nan_logits_do_not_fault(hip, oracle_mod, micro_cfg, micro_weights):
    """Every candidate -inf (suppress list = whole vocabulary) or NaN (a NaN in the final LayerNorm): the argmax has no
    winner; the reference's scan returns index 0 then (whisper_tensor.mojo:431-439), and so must the device loop — not an
    out-of-range id used as an embedding row."""
    from whisper_mojo_amd import synth
    mels = synth.synth_mels(micro_cfg, 0, 2)
    m = make_model(micro_cfg, micro_weights, max_batch=2)
    got = m.transcribe_batch(mels, prompt=(1, 2, 3, 4), eot=-1, max_loop=8, suppress_tokens=range(micro_cfg.vocab_size))
    assert got == [[1, 2, 3, 4] + [0] * 9] * 2
    ref = oracle_mod.OracleModel(micro_cfg, micro_weights)
    want = ref.transcribe(mel=mels[0], prompt=(1, 2, 3, 4), eot=-1, max_loop=8, suppress_tokens=range(micro_cfg.vocab_size))
    assert want.tolist() == got[0]
    w = micro_weights.copy()
    w[-1] = np.nan  # last float of the file = decoder ln.bias[-1] (export_weights.py:19-90 order): every logit becomes NaN
    bad = make_model(micro_cfg, w, max_batch=2)
    got = bad.transcribe_batch(mels, prompt=(1, 2, 3, 4), eot=-1, max_loop=8)
    assert got == [[1, 2, 3, 4] + [0] * 9] * 2


def test_kvcache_outliving_its_model(hip, micro_cfg, micro_weights):
    """A KVCache created on a model that is reloaded / closed: the library freed its arena with the model; using or
    deleting the stale cache is an error / no-op, never a use after free."""
    from whisper_mojo_amd import _lib, synth
    from whisper_mojo_amd.loader import WeightLoader
    from whisper_mojo_amd.whisper import KVCache
    mel = synth.synth_mel(micro_cfg, 1000)
    m = make_model(micro_cfg, micro_weights, max_batch=1)
    cache = KVCache(m, 1)
    m.encoder.forward(mel, cache)
    good = m.decoder.forward([1, 2, 3, 4], None, cache, start_pos=0)
    raw = cache._h
    m.load(WeightLoader.from_array(micro_weights))  # reload: wm_model_free + a new model (possibly at the same address)
    assert cache._h is None and cache.current_len == -1
    with pytest.raises(_lib.WhisperMiError):
        m.decoder.forward([1], None, cache, start_pos=4)
    L = _lib.lib()
    with pytest.raises(_lib.WhisperMiError, match="stale"):  # the raw C handle is rejected too
        _lib.check(L.wm_state_reset(raw))
    L.wm_state_free(raw)  # no-op
    del cache
    c2 = KVCache(m, 1)
    m.encoder.forward(mel, c2)
    assert np.array_equal(m.decoder.forward([1, 2, 3, 4], None, c2, start_pos=0), good)
    m.close()
    del c2  # after close: nothing to free
