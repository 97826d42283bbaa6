"""GPU edge cases (-m gpu): ragged batch sizes around the 16-utterance MFMA row block, degenerate decode options,
Whisper-base at full size, 1-vs-N-rank equivalence of the sharded path (ranks emulated on one GPU)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from whisper_mojo_amd import _lib
    _lib.lib()
    return True


def make_model(cfg, weights, **kw):
    from whisper_mojo_amd.loader import WeightLoader
    from whisper_mojo_amd.whisper import Whisper
    m = Whisper(cfg, **kw)
    m.load(WeightLoader.from_array(weights))
    return m


@pytest.mark.parametrize("dtype", [0, 1])
def test_ragged_batches_equal_singles(hip, micro_cfg, micro_weights, dtype):
    """B = 15 / 17 / 33 straddle the 16-row MFMA blocks of the decode kernels; every utterance must come out exactly as
    when transcribed alone (nothing in an utterance's arithmetic depends on its neighbours)."""
    from whisper_mojo_amd import synth
    mels = synth.synth_mels(micro_cfg, 0, 33)
    m = make_model(micro_cfg, micro_weights, compute_dtype=dtype, max_batch=33)
    prompt = (1, 2, 3, 4)
    full = m.transcribe_batch(mels, prompt=prompt, eot=-1, max_loop=10)
    singles = {i: m.transcribe_batch(mels[i], prompt=prompt, eot=-1, max_loop=10)[0] for i in (0, 14, 15, 16, 17, 31, 32)}
    for i, s in singles.items():
        assert full[i] == s
    for B in (15, 17):
        assert m.transcribe_batch(mels[:B], prompt=prompt, eot=-1, max_loop=10) == full[:B]


def test_degenerate_decode_options(hip, oracle_mod_, micro_cfg, micro_weights):
    from whisper_mojo_amd import synth
    mel = synth.synth_mel(micro_cfg, 1000)
    m = make_model(micro_cfg, micro_weights)
    ref = oracle_mod_.OracleModel(micro_cfg, micro_weights)
    # max_loop = 0: prompt + the one token the prefill yields (whisper.mojo:195-203)
    assert m.transcribe_batch(mel, prompt=(1, 2, 3, 4), eot=-1, max_loop=0)[0] == ref.transcribe(mel=mel, prompt=(1, 2, 3, 4), eot=-1, max_loop=0).tolist()
    # single-token prompt
    got = m.transcribe_batch(mel, prompt=(7,), eot=-1, max_loop=6)[0]
    assert got == ref.transcribe(mel=mel, prompt=(7,), eot=-1, max_loop=6).tolist() and len(got) == 8
    # first generated token is eot -> the list is prompt + [eot]
    first = got[1]
    assert m.transcribe_batch(mel, prompt=(7,), eot=first, max_loop=6)[0] == [7, first]
    # the longest stream the decoder context allows
    n = micro_cfg.n_text_ctx - 4
    long = m.transcribe_batch(mel, prompt=(1, 2, 3, 4), eot=-1, max_loop=n)[0]
    assert len(long) == 4 + 1 + n


@pytest.fixture(scope="module")
def oracle_mod_():
    from oracle import oracle
    return oracle


def test_base_full_size(hip, oracle_mod_):
    """BASELINE config 5 model (Whisper-base, full 1500-frame context and 51 865 vocabulary): fp32 tokens vs the oracle,
    f16 + f16 KV within 16-bit tolerance of the fp32 logits."""
    from whisper_mojo_amd import WhisperConfig, synth
    from whisper_mojo_amd.whisper import KVCache
    cfg = WhisperConfig.base()
    w = oracle_mod_.synth_weights_c(cfg, 0)
    mel = synth.synth_mel(cfg, 1000)
    ref = oracle_mod_.OracleModel(cfg, w)
    enc_ref = ref.encode(mel)
    m = make_model(cfg, w, max_batch=1)
    enc = m.encoder.forward(mel)
    assert np.abs(enc - enc_ref).max() < 5e-5
    want, logits = ref.transcribe(enc_out=enc_ref, max_loop=6, ignore_eot=True, want_logits=True)
    got = m.transcribe_batch(mel, max_loop=6, ignore_eot=True)[0]
    s = np.sort(logits, 1)
    margins = s[:, -1] - s[:, -2]
    first_bad = next((i for i in range(len(got)) if got[i] != want[i]), None)
    assert first_bad is None or margins[first_bad - 4] < 1e-3
    h = make_model(cfg, w, compute_dtype=2, kv_dtype=2, max_batch=1)
    cache = KVCache(h, 1)
    h.encoder.forward(mel, cache)
    lg = h.decoder.forward([50258, 50259, 50359, 50363], None, cache, start_pos=0)
    assert np.abs(lg - logits[0]).max() < 0.02


def test_one_rank_equals_two_ranks(hip, micro_cfg, micro_weights):
    """the same global batch on 1 vs N ranks gives identical token buffers.  Ranks are emulated on one GPU:
    each 'rank' runs its shard_range of the utterances through its own model instance, results are packed exactly as
    dist.gather_tokens packs them."""
    from whisper_mojo_amd import dist as wdist, synth
    total = 7
    mels = synth.synth_mels(micro_cfg, 0, total)
    prompt = (1, 2, 3, 4)
    one = make_model(micro_cfg, micro_weights, max_batch=total)
    ref = one.transcribe_batch(mels, prompt=prompt, eot=-1, max_loop=12)
    stride = 4 + 1 + 12
    bufs = []
    for r in range(2):
        first, count = wdist.shard_range(total, r, 2)
        mr = make_model(micro_cfg, micro_weights, max_batch=count)
        mr.transcribe_batch(mels[first:first + count], prompt=prompt, eot=-1, max_loop=12)
        bufs.append(wdist.pack_tokens(mr.last_tokens, mr.last_counts, stride, (total + 1) // 2)[:count])
    got = wdist.unpack_tokens(np.concatenate(bufs))
    assert got == ref


def test_pipelined_submit_wait_equals_synchronous(hip, micro_cfg, micro_weights):
    """wm_transcribe_submit / _wait on the two pipeline slots give exactly wm_transcribe's ids, in any interleaving, also in
    the eot-stopping mode (where the pipelined form enqueues all steps and finished utterances stop recording)."""
    from whisper_mojo_amd import _lib, synth
    m = make_model(micro_cfg, micro_weights, max_batch=3)
    a, b = synth.synth_mels(micro_cfg, 0, 3), synth.synth_mels(micro_cfg, 10, 3)
    kw = dict(prompt=(1, 2, 3, 4), eot=-1, max_loop=15)
    ra, rb = m.transcribe_batch(a, **kw), m.transcribe_batch(b, **kw)
    m.transcribe_submit(a, slot=0, **kw)
    m.transcribe_submit(b, slot=1, **kw)
    assert m.transcribe_wait(0) == ra
    m.transcribe_submit(a, slot=0, **kw)   # slot 0 reused while slot 1 is still in flight
    assert m.transcribe_wait(1) == rb
    assert m.transcribe_wait(0) == ra
    eot = ra[0][7]
    kw2 = dict(prompt=(1, 2, 3, 4), eot=eot, max_loop=15)
    want = m.transcribe_batch(a, **kw2)
    m.transcribe_submit(a, slot=1, **kw2)
    assert m.transcribe_wait(1) == want
    m.transcribe_submit(a, slot=0, **kw)
    with pytest.raises(_lib.WhisperMiError, match="not waited"):
        m.transcribe_submit(b, slot=0, **kw)
    m.transcribe_wait(0)
    with pytest.raises(_lib.WhisperMiError, match="nothing was submitted"):
        _lib.check(_lib.lib().wm_transcribe_wait(m._h, 0, None, None) if False else _lib.lib().wm_transcribe_wait(
            m._h, 0, np.zeros(4, np.int32).ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_int32)),
            np.zeros(4, np.int32).ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_int32))))


def test_eight_slots_in_flight(hip, micro_cfg, micro_weights):
    """All eight pipeline slots in flight at once, different batches and batch sizes per slot, waited out of order:
    every slot returns exactly the synchronous call's ids; slot 8 does not exist."""
    from whisper_mojo_amd import _lib, synth
    m = make_model(micro_cfg, micro_weights, max_batch=4)
    kw = dict(prompt=(1, 2, 3, 4), eot=-1, max_loop=10)
    mels = [synth.synth_mels(micro_cfg, 100 + 7 * s, 1 + s % 4) for s in range(8)]
    want = [m.transcribe_batch(x, **kw) for x in mels]
    for rnd in range(2):  # second round reuses every slot's state and graph
        for s in range(8):
            m.transcribe_submit(mels[s], slot=s, **kw)
        for s in (3, 0, 7, 1, 6, 2, 5, 4):
            assert m.transcribe_wait(s) == want[s], (rnd, s)
    with pytest.raises(_lib.WhisperMiError, match="slot"):
        m.transcribe_submit(mels[0], slot=8, **kw)


@pytest.mark.parametrize("q_len", [2, 3, 4, 5, 16])
def test_block_prefill_equals_stepwise(hip, micro_cfg, micro_weights, q_len):
    """WhisperDecoder.forward on a q_len block (whisper.mojo:195) = the same tokens fed one by one: same logits bit for
    bit (the block pass runs position-major rows through the same kernels; q_len = 4 also takes the multi-query
    cross-attention), and the cache ends in the same state (the next step agrees too)."""
    from whisper_mojo_amd import synth
    from whisper_mojo_amd.whisper import KVCache
    m = make_model(micro_cfg, micro_weights, max_batch=3)
    mels = synth.synth_mels(micro_cfg, 40, 3)
    r = np.random.default_rng(q_len)
    toks = r.integers(0, micro_cfg.vocab_size, (3, q_len + 1)).astype(np.int32)
    ca, cb = KVCache(m, 3), KVCache(m, 3)
    m.encoder.forward(mels, ca)
    m.encoder.forward(mels, cb)
    block = m.decoder.forward(toks[:, :q_len], None, ca, start_pos=0)
    for i in range(q_len):
        step = m.decoder.forward(toks[:, i:i + 1], None, cb, start_pos=i)
    assert np.array_equal(block, step)
    assert ca.current_len == cb.current_len == q_len
    na = m.decoder.forward(toks[:, q_len:], None, ca, start_pos=q_len)
    nb = m.decoder.forward(toks[:, q_len:], None, cb, start_pos=q_len)
    assert np.array_equal(na, nb)


# ------------------------------------------------------------------------------------------------ coalesced submits
@pytest.mark.parametrize("dtype", [0, 1])
def test_coalesced_submits_return_each_batch_its_own_ids(hip, micro_cfg, micro_weights, dtype):
    """wm_config.coalesce = 2: two consecutive wm_transcribe_submit calls share one 2·B-row decode state.  Every call must
    still return exactly what it returns uncoalesced (bit for bit: nothing in an utterance's arithmetic depends on the batch
    it rides in), whatever the pairing: pairs, a leftover that runs alone at its wait, waits in reverse order, a partner
    whose options differ (no pairing), the reference's stop rule with different stop points in the two halves."""
    from whisper_mojo_amd import _lib, synth
    B = 3
    mels = [synth.synth_mels(micro_cfg, 10 * k, B) for k in range(5)]
    kw = dict(prompt=(1, 2, 3, 4), eot=-1, max_loop=20)
    plain = make_model(micro_cfg, micro_weights, compute_dtype=dtype, max_batch=B)
    want = [plain.transcribe_batch(x, **kw) for x in mels]
    m = make_model(micro_cfg, micro_weights, compute_dtype=dtype, max_batch=B, coalesce=2)
    # two pairs + a leftover, collected in submit order
    for k in range(5):
        m.transcribe_submit(mels[k], slot=k, **kw)
    assert [m.transcribe_wait(k) for k in range(5)] == want
    # collected in reverse order; slot numbers unrelated to the pairing
    for k, slot in enumerate((6, 2, 7, 0)):
        m.transcribe_submit(mels[k], slot=slot, **kw)
    got = {slot: m.transcribe_wait(slot) for slot in (0, 7, 2, 6)}
    assert [got[6], got[2], got[7], got[0]] == want[:4]
    # a partner with other options is not paired: both run alone, both right
    kw2 = dict(kw, max_loop=12)
    want2 = plain.transcribe_batch(mels[1], **kw2)
    m.transcribe_submit(mels[0], slot=0, **kw)
    m.transcribe_submit(mels[1], slot=1, **kw2)
    assert m.transcribe_wait(1) == want2 and m.transcribe_wait(0) == want[0]
    # a busy slot is refused while its batch is only held, too
    m.transcribe_submit(mels[0], slot=3, **kw)
    with pytest.raises(_lib.WhisperMiError, match="not waited"):
        m.transcribe_submit(mels[1], slot=3, **kw)
    # the synchronous call flushes the held batch first and is unaffected by it
    assert m.transcribe_batch(mels[2], **kw) == want[2]
    assert m.transcribe_wait(3) == want[0]
    # the reference's stop rule: halves that stop at different iterations (the pair runs until BOTH are done)
    free = want[0][0]
    eot = free[4 + 6]
    kw3 = dict(prompt=(1, 2, 3, 4), eot=eot, max_loop=40)
    w0, w1 = plain.transcribe_batch(mels[0], **kw3), plain.transcribe_batch(mels[1], **kw3)
    m.transcribe_submit(mels[0], slot=0, **kw3)
    m.transcribe_submit(mels[1], slot=1, **kw3)
    assert m.transcribe_wait(0) == w0 and m.transcribe_wait(1) == w1
    m.close()
    plain.close()


@pytest.mark.parametrize("dec_fp32", [True, False])
def test_coalesced_tiny_b64_pairs_equal_uncoalesced(hip, tiny_cfg, tiny_weights, dec_fp32):
    """The benched form: BASELINE config 3 as written (bf16 encoder, fp32 decoder + KV) — and the all-16-bit variant —, 64 clips per
    submit, coalesce = 2, four submits in flight = two 128-row passes (128-row logits kernels).  Each submit's ids equal the
    uncoalesced model's, bit for bit, all 64 rows."""
    import ctypes as C
    from whisper_mojo_amd import _lib
    L = _lib.lib()
    mels = np.empty((128, 80, 3000), np.float32)
    for i in range(128):
        L.wm_synth_mel_host(1000 + i, 80, 3000, mels[i].ctypes.data_as(C.POINTER(C.c_float)))
    kw = dict(max_loop=30, ignore_eot=True)
    mk = dict(compute_dtype=1, kv_dtype=0 if dec_fp32 else 1, max_batch=64, decoder_fp32=dec_fp32)
    plain = make_model(tiny_cfg, tiny_weights, **mk)
    want = [plain.transcribe_batch(mels[:64], **kw), plain.transcribe_batch(mels[64:], **kw)]
    plain.close()
    m = make_model(tiny_cfg, tiny_weights, coalesce=2, **mk)
    for slot, half in ((0, 0), (1, 1), (2, 1), (3, 0)):
        m.transcribe_submit(mels[64 * half:64 * half + 64], slot=slot, **kw)
    assert m.transcribe_wait(0) == want[0] and m.transcribe_wait(1) == want[1]
    assert m.transcribe_wait(2) == want[1] and m.transcribe_wait(3) == want[0]
    m.close()


@pytest.mark.parametrize("dtype", [0, 1, 2])
def test_logits_128_row_kernel_equals_64_row_kernel(hip, micro_cfg, micro_weights, dtype):
    """Decode states of more than 64 rows take the 128-rows-per-workgroup logits kernel (fp32 decoder: two K passes over three
    bf16 images; LayerNorm statistics summed in the 64-row kernel's order).  Its ids must equal the 64-row kernel's bit for
    bit: B = 80 in one state against the same utterances in batches of 40 (64-row kernel), and 40 + 40 coalesced."""
    from whisper_mojo_amd import synth
    mels = synth.synth_mels(micro_cfg, 300, 80)
    kw = dict(prompt=(1, 2, 3, 4), eot=-1, max_loop=24)
    small = make_model(micro_cfg, micro_weights, compute_dtype=dtype, max_batch=40)
    want = small.transcribe_batch(mels[:40], **kw) + small.transcribe_batch(mels[40:], **kw)
    small.close()
    big = make_model(micro_cfg, micro_weights, compute_dtype=dtype, max_batch=80)
    assert big.transcribe_batch(mels, **kw) == want
    big.close()
    if dtype == 0:  # two row blocks of the 128-row kernel, the second one ragged (150 = 128 + 22 rows)
        more = np.concatenate([mels, mels[:70]])
        huge = make_model(micro_cfg, micro_weights, compute_dtype=dtype, max_batch=150)
        assert huge.transcribe_batch(more, **kw) == want + want[:70]
        huge.close()
    pair = make_model(micro_cfg, micro_weights, compute_dtype=dtype, max_batch=40, coalesce=2)
    pair.transcribe_submit(mels[:40], slot=0, **kw)
    pair.transcribe_submit(mels[40:], slot=1, **kw)
    assert pair.transcribe_wait(0) + pair.transcribe_wait(1) == want
    pair.close()
