"""GPU parity tests (-m gpu): the HIP path, called through the C-ABI, against the CPU oracle and the committed
golden fixtures on the same seeded inputs.

Bars (the path is IEEE fp32 except token ids / argmax indices):
  * token ids and argmax indices: exact, wherever the oracle's top1-top2 margin exceeds the fp32 logit tolerance
    (greedy argmax is discontinuous; a different summation order may only flip a near-tie, and the test reports it);
  * fp32 mode: encoder rows and logits within 5e-5 abs of the oracle (values are O(1); different reduction order);
  * bf16 / f16 operand modes: within 16-bit tolerances written next to each test.
Nothing here reads /root/reference."""
import os

import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu

F32_TOL = 5e-5
MARGIN_TOL = 1e-3


@pytest.fixture(scope="module")
def hip():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from whisper_mojo_amd import _lib
    _lib.lib()  # raises if the HIP library is missing: no fallback
    import whisper_mojo_amd as pkg
    return pkg


@pytest.fixture(scope="module")
def oracle_mod():
    from oracle import oracle
    return oracle


def make_model(cfg, weights, dtype=0, kv=None, gelu=0, pos=0, max_batch=4):
    from whisper_mojo_amd.loader import WeightLoader
    from whisper_mojo_amd.whisper import Whisper
    m = Whisper(cfg, compute_dtype=dtype, kv_dtype=kv, gelu_mode=gelu, pos_mode=pos, max_batch=max_batch)
    m.load(WeightLoader.from_array(weights))
    return m


# ---------------------------------------------------------------------------------------------- op-level KATs
rng = np.random.default_rng(42)


@pytest.mark.parametrize("M,N,K", [(1, 37, 64), (4, 16, 96), (5, 1037, 384), (64, 256, 384), (200, 128, 96), (130, 384, 1536),
                                   (17, 50, 32), (70, 24, 128)])
def test_op_matmul(hip, oracle_mod, M, N, K):
    from whisper_mojo_amd import whisper_tensor as wt
    A, B = rng.standard_normal((M, K), np.float32), rng.standard_normal((N, K), np.float32)
    b = rng.standard_normal(N, np.float32)
    for bias in (None, b):
        out = wt.Tensor(M, N)
        wt.matmul(out, A, B, bias)
        ref = oracle_mod.matmul(A, B, bias)
        assert np.abs(out - ref).max() < 1e-5 * K ** 0.5 * 8


def test_op_matmul_wide_ragged_repeat(hip, oracle_mod):
    """N >= 1024 takes two 16-column tiles per workgroup; with round_up(N, 16) % 32 == 16 the last workgroup has only one.
    (A store of the missing tile would race with the next row's first columns — repeat to give a race the chance.)"""
    from whisper_mojo_amd import whisper_tensor as wt
    A, B = rng.standard_normal((9, 256), np.float32), rng.standard_normal((1037, 256), np.float32)
    ref = oracle_mod.matmul(A, B, None)
    for _ in range(20):
        out = wt.Tensor(9, 1037)
        wt.matmul(out, A, B, None)
        assert np.abs(out - ref).max() < 1e-3


def test_op_matmul_bf16_rounding(hip):
    """16-bit operand mode = exact products of the rounded operands (fp32 accumulate)."""
    import torch
    from whisper_mojo_amd import whisper_tensor as wt, DT_BF16
    A, B = rng.standard_normal((64, 384), np.float32), rng.standard_normal((128, 384), np.float32)
    out = wt.Tensor(64, 128)
    wt.matmul(out, A, B, None, dtype=DT_BF16)
    Ab = torch.from_numpy(A).bfloat16().double().numpy()
    Bb = torch.from_numpy(B).bfloat16().double().numpy()
    assert np.abs(out - Ab @ Bb.T).max() < 1e-4


@pytest.mark.parametrize("M,N", [(4741, 1152), (20000, 3072), (130, 128)])
def test_op_matmul_rowpanel_bf16(hip, M, N):
    """K = 384 with 16-bit operands runs on the A-stationary row-panel kernel (counted-vmcnt LDS ring, several units and
    panel switches per workgroup, ragged last panel): exact products of the rounded operands, and the same bits every time."""
    import torch
    from whisper_mojo_amd import whisper_tensor as wt, DT_BF16
    r = np.random.default_rng(M + N)
    A, B = r.standard_normal((M, 384), np.float32), r.standard_normal((N, 384), np.float32)
    bias = r.standard_normal(N, np.float32)
    Ab = torch.from_numpy(A).bfloat16().float().numpy()
    Bb = torch.from_numpy(B).bfloat16().float().numpy()
    ref = Ab.astype(np.float64) @ Bb.T.astype(np.float64) + bias
    first = None
    for _ in range(3):
        out = wt.Tensor(M, N)
        wt.matmul(out, A, B, bias, dtype=DT_BF16)
        assert np.abs(out - ref).max() < 2e-4
        if first is None:
            first = out.copy()
        else:
            assert np.array_equal(out, first)


@pytest.mark.parametrize("M,K", [(4741, 1536), (130, 384), (1500, 1152)])
def test_op_matmul_fullrow_bf16(hip, M, K):
    """N = 384 with 16-bit operands and fp32 output runs on the full-row kernel (half-stage LDS ring, asm fragment pipeline,
    ragged last panel, 6 / 18 / 24 k64 steps): exact products of the rounded operands, and the same bits every time."""
    import torch
    from whisper_mojo_amd import whisper_tensor as wt, DT_BF16
    r = np.random.default_rng(M + K)
    A, B = r.standard_normal((M, K), np.float32), r.standard_normal((384, K), np.float32)
    bias = r.standard_normal(384, np.float32)
    Ab = torch.from_numpy(A).bfloat16().float().numpy()
    Bb = torch.from_numpy(B).bfloat16().float().numpy()
    ref = Ab.astype(np.float64) @ Bb.T.astype(np.float64) + bias
    first = None
    for _ in range(3):
        out = wt.Tensor(M, 384)
        wt.matmul(out, A, B, bias, dtype=DT_BF16)
        assert np.abs(out - ref).max() < 4e-4 * np.sqrt(K / 384.0)
        if first is None:
            first = out.copy()
        else:
            assert np.array_equal(out, first)


def _bf16_round(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).bfloat16().float().numpy()


def _mlp_block_ref(x, g1, b1, w1, bb1, w2, bb2, g2, b2, rnd):
    """layers.mojo:489-517 in float64, with the operand roundings of the 16-bit path (rnd) where the kernels round: the
    LayerNorm output fed to fc1, the hidden activations fed to fc2, the weights, and the returned next-LayerNorm rows."""
    def ln(v, g, b):
        v32 = v.astype(np.float32)
        mean = v32.mean(axis=1, keepdims=True, dtype=np.float64)
        var = (v32.astype(np.float64) ** 2).mean(axis=1, keepdims=True) - mean * mean  # one-pass variance (whisper_tensor.mojo:273)
        return (v32 - mean) / np.sqrt(var + 1e-5) * g + b
    xn = rnd(ln(x, g1, b1)).astype(np.float64)
    h = xn @ rnd(w1).T.astype(np.float64) + bb1
    h = 0.5 * h * (1.0 + np.tanh(0.79788456 * (h + 0.044715 * h ** 3)))  # whisper_tensor.mojo:288-308
    h = rnd(h).astype(np.float64)
    out = x.astype(np.float64) + h @ rnd(w2).T.astype(np.float64) + bb2
    return out, ln(out, g2, b2)


@pytest.mark.parametrize("M", [300, 4741])
def test_op_mlp_block_bf16(hip, M):
    """The MLP half of a block on the encoder's 16-bit kernels: LayerNorm inside the fc1 GEMM's A load, tanh-GELU epilogue,
    fc2 on the full-row kernel with the residual add and the NEXT LayerNorm in its epilogue — against float64 with the same
    operand roundings.  Differences left: fp32 accumulation order and single-ulp flips of rounded intermediates."""
    from whisper_mojo_amd import whisper_tensor as wt, DT_BF16
    r = np.random.default_rng(M)
    d, ffn = 384, 1536
    x = (r.standard_normal((M, d)) * 1.5 + 0.3).astype(np.float32)
    g1, b1 = (1.0 + 0.2 * r.standard_normal(d)).astype(np.float32), (0.1 * r.standard_normal(d)).astype(np.float32)
    g2, b2 = (1.0 + 0.2 * r.standard_normal(d)).astype(np.float32), (0.1 * r.standard_normal(d)).astype(np.float32)
    w1, bb1 = (r.standard_normal((ffn, d)) / np.sqrt(d)).astype(np.float32), (0.1 * r.standard_normal(ffn)).astype(np.float32)
    w2, bb2 = (r.standard_normal((d, ffn)) / np.sqrt(ffn)).astype(np.float32), (0.1 * r.standard_normal(d)).astype(np.float32)
    ref_x, ref_xn = _mlp_block_ref(x, g1, b1, w1, bb1, w2, bb2, g2, b2, _bf16_round)
    got = x.copy()
    xn = wt.mlp_block(got, g1, b1, w1, bb1, w2, bb2, next_ln=(g2, b2), dtype=DT_BF16)
    assert np.abs(got - ref_x).max() < 1e-2          # measured 1.0e-3 / 4.7e-3 (M = 300 / 4741): a bf16 ulp flip of a hidden value moves its sum
    assert np.abs(got - ref_x).mean() < 5e-5         # measured 2e-6 / 4e-6
    assert np.array_equal(_bf16_round(xn), xn)       # really 16-bit values
    err = np.abs(xn - ref_xn)
    assert err.max() < 0.04                          # one bf16 ulp at |v| < 8 (measured 0.0156: one ulp at |v| in [2, 4))
    assert (err > 0.004 * (1 + np.abs(ref_xn))).mean() < 0.001  # measured 0: beyond half-ulp rounding only where a tie flipped
    again = x.copy()
    xn2 = wt.mlp_block(again, g1, b1, w1, bb1, w2, bb2, next_ln=(g2, b2), dtype=DT_BF16)
    assert np.array_equal(again, got) and np.array_equal(xn2, xn)


def test_op_mlp_block_fp32(hip):
    """The same entry with exact fp32 operands (plain LayerNorm + GEMM kernels): float64 reference, no roundings."""
    from whisper_mojo_amd import whisper_tensor as wt, DT_F32
    r = np.random.default_rng(7)
    M, d, ffn = 200, 384, 1536
    x = r.standard_normal((M, d)).astype(np.float32)
    g1, b1 = (1.0 + 0.2 * r.standard_normal(d)).astype(np.float32), (0.1 * r.standard_normal(d)).astype(np.float32)
    g2, b2 = (1.0 + 0.2 * r.standard_normal(d)).astype(np.float32), (0.1 * r.standard_normal(d)).astype(np.float32)
    w1, bb1 = (r.standard_normal((ffn, d)) / np.sqrt(d)).astype(np.float32), (0.1 * r.standard_normal(ffn)).astype(np.float32)
    w2, bb2 = (r.standard_normal((d, ffn)) / np.sqrt(ffn)).astype(np.float32), (0.1 * r.standard_normal(d)).astype(np.float32)
    ref_x, ref_xn = _mlp_block_ref(x, g1, b1, w1, bb1, w2, bb2, g2, b2, lambda a: np.asarray(a, np.float32))
    got = x.copy()
    xn = wt.mlp_block(got, g1, b1, w1, bb1, w2, bb2, next_ln=(g2, b2), dtype=DT_F32)
    assert np.abs(got - ref_x).max() < 2e-5
    assert np.abs(xn - ref_xn).max() < 5e-5


def _attention_ref(q, k, v, n_heads, rnd):
    """layers.mojo:273-342 in float64 on the operands as the kernel sees them (rnd = operand rounding): per head
    softmax(q_h·k_hᵀ · 0.125)·v_h — scale after the product (Q4), no mask."""
    q, k, v = (rnd(a).astype(np.float64) for a in (q, k, v))
    out = np.empty_like(q)
    for h in range(n_heads):
        sl = slice(64 * h, 64 * h + 64)
        s_ = (q[:, sl] @ k[:, sl].T) * 0.125
        s_ -= s_.max(axis=1, keepdims=True)
        p = np.exp(s_)
        out[:, sl] = (p / p.sum(axis=1, keepdims=True)) @ v[:, sl]
    return out


def _attention_case(name, n_ctx, n_heads, r):
    d = 64 * n_heads
    q, k, v = (r.standard_normal((n_ctx, d)).astype(np.float32) for _ in range(3))
    if name == "growing":
        # scores that keep outgrowing any earlier row maximum (and, for half the rows, keep falling): key j = u·(0.4 j), so
        # s_ij = (q_i·u)·0.05 j reaches ±several hundred — the lazily refreshed softmax reference has to be refreshed many times
        u = r.standard_normal(d).astype(np.float32)
        u /= np.linalg.norm(u.reshape(n_heads, 64), axis=1).repeat(64)
        k = (u[None, :] * (0.4 * np.arange(n_ctx, dtype=np.float32))[:, None]).astype(np.float32) + 0.1 * k
    elif name == "outlier":
        # one key in a middle tile beats everything by ~2^100, another row block sees it as hugely negative
        j = n_ctx // 2 + 3
        k[j] = 12.0 * np.sign(q[n_ctx // 3])
    return q, k, v


@pytest.mark.parametrize("case,n_ctx,n_heads", [("random", 1500, 6), ("growing", 1500, 6), ("outlier", 700, 8), ("random", 70, 6),
                                                ("random", 64, 6), ("random", 1, 6), ("growing", 200, 2)])
def test_op_attention_16bit(hip, case, n_ctx, n_heads):
    """Encoder attention (fused, never writes S; softmax against a lazily refreshed reference maximum; key mask in a peeled last
    tile) against float64 on the rounded operands: random scores, scores that keep growing / falling along the keys (the
    reference maximum must be refreshed again and again), a 2^100 outlier in a middle tile, ragged and single-tile lengths."""
    import torch
    from whisper_mojo_amd import whisper_tensor as wt, DT_BF16, DT_F16
    r = np.random.default_rng(n_ctx * 31 + n_heads)
    q, k, v = _attention_case(case, n_ctx, n_heads, r)
    for dt, rnd, tol in ((DT_BF16, _bf16_round, 8e-3), (DT_F16, lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).half().float().numpy(), 1e-3)):
        ref = _attention_ref(q, k, v, n_heads, rnd)
        out = wt.Tensor(n_ctx, 64 * n_heads)
        wt.attention(out, q, k, v, n_heads, dtype=dt)
        assert np.isfinite(out).all()
        # probabilities and the output are rounded to 16 bits: ~2^-8 (bf16) / 2^-11 (f16) of the output's magnitude
        # (measured, worst of the seven cases: 3.9e-3 bf16, 4.6e-4 f16 of max |out|)
        assert np.abs(out - ref).max() < tol * max(1.0, np.abs(ref).max()), (case, dt, np.abs(out - ref).max())
        again = wt.Tensor(n_ctx, 64 * n_heads)
        wt.attention(again, q, k, v, n_heads, dtype=dt)
        assert np.array_equal(out, again)


def test_op_attention_fp32(hip):
    from whisper_mojo_amd import whisper_tensor as wt, DT_F32
    r = np.random.default_rng(5)
    for case, n_ctx in (("random", 300), ("growing", 200)):
        q, k, v = _attention_case(case, n_ctx, 6, r)
        ref = _attention_ref(q, k, v, 6, lambda a: np.asarray(a, np.float32))
        out = wt.Tensor(n_ctx, 384)
        wt.attention(out, q, k, v, 6, dtype=DT_F32)
        assert np.abs(out - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())


def _attention_cached_ref(q, k, v, n_heads, rnd):
    """layers.mojo:186-272 in float64: per utterance and head s_j = (q·K_j)·0.125, softmax over the cached rows, o = Σ p_j V_j."""
    q = q.astype(np.float64)
    k, v = rnd(k).astype(np.float64), rnd(v).astype(np.float64)
    out = np.empty_like(q)
    for h in range(n_heads):
        sl = slice(64 * h, 64 * h + 64)
        s_ = np.einsum("bd,btd->bt", q[:, sl], k[:, :, sl]) * 0.125
        s_ -= s_.max(axis=1, keepdims=True)
        p = np.exp(s_)
        out[:, sl] = np.einsum("bt,btd->bd", p / p.sum(axis=1, keepdims=True), v[:, :, sl])
    return out


@pytest.mark.parametrize("B,t,chunks", [(1, 1, 1), (3, 7, 1), (5, 200, 1), (2, 448, 1), (1, 1500, 16), (5, 1500, 16), (3, 1500, 47), (2, 97, 3)])
def test_op_attention_cached(hip, B, t, chunks):
    """The decode attention (q_len == 1 over cached K/V: the single-workgroup self-attention form and the chunked + merged
    cross-attention form) against float64 on the cache as stored, for fp32 / bf16 / f16 caches; queries stay fp32."""
    import torch
    from whisper_mojo_amd import whisper_tensor as wt, DT_F32, DT_BF16, DT_F16
    r = np.random.default_rng(B * 1000 + t)
    H = 6
    q = (r.standard_normal((B, 64 * H)) * 1.5).astype(np.float32)
    k = r.standard_normal((B, t, 64 * H)).astype(np.float32)
    v = r.standard_normal((B, t, 64 * H)).astype(np.float32)
    if t > 4:
        k[:, t // 2] = 3.0 * np.sign(q)  # one dominant key: the running-max path matters
    h16 = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).half().float().numpy()
    for dt, rnd, tol in ((DT_F32, lambda a: a, 2e-5), (DT_BF16, _bf16_round, 2e-5), (DT_F16, h16, 2e-5)):
        ref = _attention_cached_ref(q, k, v, H, rnd)
        out = wt.Tensor(B, 64 * H)
        wt.attention_cached(out, q, k, v, H, kv_dtype=dt, n_chunks=chunks)
        # the cache rounding is in the reference; what is left is fp32 arithmetic
        assert np.abs(out - ref).max() < tol * max(1.0, np.abs(ref).max()), (dt, np.abs(out - ref).max())


def test_op_layer_norm(hip, oracle_mod):
    from whisper_mojo_amd import whisper_tensor as wt
    for rows, cols in ((9, 384), (1, 128), (70, 512)):
        x = rng.standard_normal((rows, cols), np.float32) * 3 + 1
        g, b = rng.standard_normal(cols, np.float32), rng.standard_normal(cols, np.float32)
        out = wt.Tensor(rows, cols)
        wt.layer_norm(out, x, g, b)
        assert np.abs(out - oracle_mod.layer_norm(x, g, b)).max() < 1e-5


def test_op_gelu_softmax_argmax(hip, oracle_mod):
    from whisper_mojo_amd import whisper_tensor as wt
    t = rng.standard_normal((3, 67), np.float32) * 2
    for mode in (0, 1):
        t2 = t.copy()
        wt.gelu(t2, mode)
        assert np.abs(t2 - oracle_mod.gelu(t, mode)).max() < 1e-6
        assert np.array_equal(t2.ravel()[200:], t.ravel()[200:])  # 201 % 8 = 1: tail untouched like the reference
    for cols in (3, 13, 1500):
        s = rng.standard_normal((5, cols), np.float32) * 4
        s2 = s.copy()
        wt.softmax(s2)
        assert np.abs(s2 - oracle_mod.softmax(s)).max() < 1e-6
    v = rng.standard_normal(51865).astype(np.float32)
    assert wt.argmax(v) == oracle_mod.argmax(v)
    v[[17, 40000, 51864]] = 9.0  # ties: lowest index wins (whisper_tensor.mojo:436)
    assert wt.argmax(v) == 17


@pytest.mark.parametrize("C_in,L,stride,out_T", [(80, 300, 1, False), (128, 301, 2, True), (16, 64, 2, False), (40, 50, 1, True)])
def test_op_conv1d(hip, oracle_mod, C_in, L, stride, out_T):
    from whisper_mojo_amd import whisper_tensor as wt
    x = rng.standard_normal((C_in, L), np.float32)
    w = rng.standard_normal((128, C_in, 3), np.float32) * 0.1
    b = rng.standard_normal(128, np.float32)
    L_out = (L + 2 - 3) // stride + 1
    out = wt.Tensor(*((L_out, 128) if out_T else (128, L_out)))
    wt.conv1d(out, x, w, b, stride, 1, out_T)
    ref = oracle_mod.conv1d(x, oracle_mod.transpose_conv_weights(w), b, stride, 1, out_T)
    assert np.abs(out - ref).max() < 2e-5


@pytest.mark.parametrize("C_in,L,stride", [(384, 3000, 2), (384, 601, 2), (384, 257, 1)])
def test_op_conv1d_bf16_full_row(hip, oracle_mod, C_in, L, stride):
    """conv2's shape (C_out = 384, K = 3·384) with 16-bit operands: the implicit GEMM over overlapping input rows runs on the
    full-row kernel (ragged last panel, lda = stride·C_in).  Against the oracle on the rounded operands (fp32 accumulate)."""
    from whisper_mojo_amd import whisper_tensor as wt, DT_BF16
    r = np.random.default_rng(C_in + L)
    x = r.standard_normal((C_in, L)).astype(np.float32)
    w = (r.standard_normal((384, C_in, 3)) / np.sqrt(3 * C_in)).astype(np.float32)
    b = r.standard_normal(384).astype(np.float32)
    L_out = (L + 2 - 3) // stride + 1
    out = wt.Tensor(L_out, 384)
    wt.conv1d(out, x, w, b, stride, 1, True, dtype=DT_BF16)
    ref = oracle_mod.conv1d(_bf16_round(x), oracle_mod.transpose_conv_weights(_bf16_round(w)), b, stride, 1, True)
    assert np.abs(out - ref).max() < 2e-4, np.abs(out - ref).max()


# ---------------------------------------------------------------------------------------------- model-level
def _first_divergence(got, want):
    n = min(len(got), len(want))
    for i in range(n):
        if got[i] != want[i]:
            return i
    return None if len(got) == len(want) else n


def _assert_tokens(got, want, logits_ref, n_prompt):
    """Token-exact unless the oracle itself was at a near-tie where the streams part."""
    i = _first_divergence(list(got), list(want))
    if i is None:
        return
    s = np.sort(logits_ref[i - n_prompt])
    margin = s[-1] - s[-2]
    assert margin < MARGIN_TOL, f"token {i}: HIP {got[i]} vs oracle {want[i]} with oracle margin {margin}"


@pytest.mark.parametrize("mode", ["ref", "hf"])
def test_micro_against_oracle_and_golden(hip, oracle_mod, micro_cfg, micro_weights, mode):
    from whisper_mojo_amd import synth
    from whisper_mojo_amd.whisper import KVCache
    gm, pm = {"ref": (0, 0), "hf": (1, 1)}[mode]
    g = golden(f"micro_{mode}")
    mel = synth.synth_mel(micro_cfg, 1000)
    ref = oracle_mod.OracleModel(micro_cfg, micro_weights, gelu_mode=gm)
    m = make_model(micro_cfg, micro_weights, gelu=gm, pos=pm)
    enc = m.encoder.forward(mel)
    enc_ref = ref.encode(mel)
    assert np.abs(enc - enc_ref).max() < F32_TOL
    assert np.abs(enc - g["enc_out"]).max() < F32_TOL
    # teacher-forced logits, step by step through WhisperDecoder.forward with caller-owned positions
    forced = g["forced_tokens"]
    cache = KVCache(m, 1)
    rows = [m.decoder.forward(forced[:4].tolist(), enc_ref, cache, start_pos=0)]
    for i in range(4, len(forced)):
        sp = cache.current_len - 1 if pm == 0 else cache.current_len  # whisper.mojo:217 lives in the host loop
        rows.append(m.decoder.forward([int(forced[i])], None, cache, start_pos=sp))
    lg = np.stack(rows)
    assert np.abs(lg - g["forced_logits"]).max() < F32_TOL
    assert np.abs(lg - ref.teacher_forced(enc_ref, forced, 4, pm)).max() < F32_TOL
    assert np.array_equal(lg.argmax(1), g["forced_top_idx"][:, 0])
    # free-running greedy loop on the device
    steps = len(forced) - 4
    got = m.transcribe_batch(mel, prompt=g["prompt"], eot=-1, max_loop=steps)[0]
    _assert_tokens(got, g["greedy_tokens"], g["greedy_logits"], 4)


@pytest.mark.parametrize("mode", ["ref", "hf"])
def test_tiny_b1_fp32_token_exact(hip, oracle_mod, tiny_cfg, tiny_weights, mode):
    """BASELINE config 2: Whisper-tiny, batch 1, fp32, greedy — token-exact vs the oracle and the HF-generated golden."""
    from whisper_mojo_amd import synth
    gm, pm = {"ref": (0, 0), "hf": (1, 1)}[mode]
    g = golden(f"tiny_{mode}")
    mel = synth.synth_mel(tiny_cfg, 1000)
    m = make_model(tiny_cfg, tiny_weights, gelu=gm, pos=pm, max_batch=1)
    enc = m.encoder.forward(mel)
    rows = g["enc_rows"]
    assert np.abs(enc[rows] - g["enc_out_rows"]).max() < F32_TOL
    assert np.abs(enc.astype(np.float64).sum(1) - g["enc_out_rowsum"]).max() < 1e-3
    steps = len(g["forced_tokens"]) - 4
    got = m.transcribe_batch(mel, max_loop=steps, ignore_eot=True)[0]
    assert got == g["greedy_tokens"].tolist()  # fixture margins >= 0.014 >> fp32 logit error
    ref = oracle_mod.OracleModel(tiny_cfg, tiny_weights, gelu_mode=gm)
    want = ref.transcribe(mel=mel, max_loop=steps, pos_mode=pm, ignore_eot=True)
    assert got == want.tolist()
    # teacher-forced top-8 logits vs the golden
    from whisper_mojo_amd.whisper import KVCache
    forced = g["forced_tokens"]
    cache = KVCache(m, 1)
    m.encoder.forward(mel, cache)
    lg = [m.decoder.forward(forced[:4].tolist(), None, cache, start_pos=0)]
    for i in range(4, len(forced)):
        sp = cache.current_len - 1 if pm == 0 else cache.current_len
        lg.append(m.decoder.forward([int(forced[i])], None, cache, start_pos=sp))
    lg = np.stack(lg)
    assert np.abs(np.take_along_axis(lg, g["forced_top_idx"], 1) - g["forced_top_val"]).max() < F32_TOL
    assert np.abs(lg[:, :64] - g["forced_logit_slice"]).max() < F32_TOL
    assert np.array_equal(lg.argmax(1), g["forced_top_idx"][:, 0])


def test_batch_equals_singles_and_oracle(hip, oracle_mod, micro_cfg, micro_weights):
    """The batch dimension the reference lacks: a batch of different mels gives, per utterance, the single-utterance
    result bit for bit (no cross-utterance state), ragged batch sizes included, and matches the oracle."""
    from whisper_mojo_amd import synth
    mels = synth.synth_mels(micro_cfg, 0, 5)
    m = make_model(micro_cfg, micro_weights, max_batch=5)
    ref = oracle_mod.OracleModel(micro_cfg, micro_weights)
    prompt = (1, 2, 3, 4)
    batch = m.transcribe_batch(mels, prompt=prompt, eot=-1, max_loop=20)
    enc_b = m.encoder.forward(mels)
    for b in range(5):
        single = m.transcribe_batch(mels[b], prompt=prompt, eot=-1, max_loop=20)[0]
        assert single == batch[b]
        assert np.array_equal(m.encoder.forward(mels[b]), enc_b[b])
        want, logits = ref.transcribe(mel=mels[b], prompt=prompt, eot=-1, max_loop=20, want_logits=True)
        _assert_tokens(batch[b], want, logits, 4)
    three = m.transcribe_batch(mels[:3], prompt=prompt, eot=-1, max_loop=20)
    assert three == batch[:3]


def test_eot_stop_rule_per_utterance(hip, oracle_mod, micro_cfg, micro_weights):
    """whisper.mojo:205-221: eot is appended, then that utterance stops; others continue; <= n_prompt+1+max_loop ids."""
    from whisper_mojo_amd import synth
    mels = synth.synth_mels(micro_cfg, 0, 4)
    m = make_model(micro_cfg, micro_weights, max_batch=4)
    ref = oracle_mod.OracleModel(micro_cfg, micro_weights)
    prompt = (1, 2, 3, 4)
    free = m.transcribe_batch(mels, prompt=prompt, eot=-1, max_loop=30)
    assert all(len(f) == 35 for f in free)
    eot = free[0][8]  # a token the first utterance emits early
    got = m.transcribe_batch(mels, prompt=prompt, eot=eot, max_loop=30)
    for b in range(4):
        want = ref.transcribe(mel=mels[b], prompt=prompt, eot=eot, max_loop=30)
        assert got[b] == want.tolist()
        if eot in free[b][4:]:
            assert got[b][-1] == eot and got[b].count(eot) == 1 + free[b][:4].count(eot)
    assert len(got[0]) < 35


def test_fixed_mode_ignores_eot(hip, micro_cfg, micro_weights):
    from whisper_mojo_amd import synth
    mel = synth.synth_mel(micro_cfg, 1000)
    m = make_model(micro_cfg, micro_weights)
    free = m.transcribe_batch(mel, prompt=(1, 2, 3, 4), eot=-1, max_loop=12)[0]
    fixed = m.transcribe_batch(mel, prompt=(1, 2, 3, 4), eot=free[5], max_loop=12, ignore_eot=True)[0]
    assert fixed == free


def test_device_resident_mel_equals_host_mel(hip, micro_cfg, micro_weights):
    import torch
    from whisper_mojo_amd import synth
    mels = synth.synth_mels(micro_cfg, 3, 2)
    m = make_model(micro_cfg, micro_weights, max_batch=2)
    a = m.transcribe_batch(mels, prompt=(1, 2, 3, 4), eot=-1, max_loop=10)
    b = m.transcribe_batch(torch.from_numpy(mels).cuda(), prompt=(1, 2, 3, 4), eot=-1, max_loop=10)
    assert a == b


@pytest.mark.parametrize("dtype,kv,enc_tol,logit_tol", [(1, 1, 0.06, 0.06), (1, 0, 0.06, 0.06), (2, 2, 0.008, 0.008)])
def test_tiny_16bit_modes_within_tolerance(hip, oracle_mod, tiny_cfg, tiny_weights, dtype, kv, enc_tol, logit_tol):
    """BASELINE configs 3 / 5 numerics: 16-bit GEMM operands (bf16: 8-bit mantissa -> ~4e-3 relative per operand;
    f16: 11-bit) with fp32 accumulation.  Tolerances are absolute on O(1) values (|enc_out| <= 4.3, logit std 1)."""
    from whisper_mojo_amd import synth
    from whisper_mojo_amd.whisper import KVCache
    g = golden("tiny_ref")
    mel = synth.synth_mel(tiny_cfg, 1000)
    m = make_model(tiny_cfg, tiny_weights, dtype=dtype, kv=kv, max_batch=1)
    enc = m.encoder.forward(mel)
    rows = g["enc_rows"]
    assert np.abs(enc[rows] - g["enc_out_rows"]).max() < enc_tol
    forced = g["forced_tokens"]
    cache = KVCache(m, 1)
    m.encoder.forward(mel, cache)
    lg = [m.decoder.forward(forced[:4].tolist(), None, cache, start_pos=0)]
    for i in range(4, len(forced)):
        lg.append(m.decoder.forward([int(forced[i])], None, cache, start_pos=cache.current_len - 1))
    lg = np.stack(lg)
    err = np.abs(np.take_along_axis(lg, g["forced_top_idx"], 1) - g["forced_top_val"]).max()
    assert err < logit_tol
    margins = g["forced_top_val"][:, 0] - g["forced_top_val"][:, 1]
    clear = margins > 4 * logit_tol
    assert np.array_equal(lg.argmax(1)[clear], g["forced_top_idx"][clear, 0])


def test_tiny_bf16_encoder_fp32_decoder(hip, oracle_mod, tiny_cfg, tiny_weights):
    """BASELINE config 3 read literally — bf16 ENCODER GEMMs, the decoder's weights / operands / KV cache fp32
    (wm_config.decoder_fp32): the only 16-bit rounding left is in the encoder output and the cross-K/V projection, so the
    teacher-forced logits sit well inside the all-bf16 bound and every clear-margin top-1 agrees."""
    from whisper_mojo_amd import synth
    from whisper_mojo_amd.loader import WeightLoader
    from whisper_mojo_amd.whisper import KVCache, Whisper
    g = golden("tiny_ref")
    mel = synth.synth_mel(tiny_cfg, 1000)
    m = Whisper(tiny_cfg, compute_dtype=1, kv_dtype=0, max_batch=1, decoder_fp32=True)
    m.load(WeightLoader.from_array(tiny_weights))
    enc = m.encoder.forward(mel)
    assert np.abs(enc[g["enc_rows"]] - g["enc_out_rows"]).max() < 0.06
    forced = g["forced_tokens"]
    cache = KVCache(m, 1)
    m.encoder.forward(mel, cache)
    lg = [m.decoder.forward(forced[:4].tolist(), None, cache, start_pos=0)]
    for i in range(4, len(forced)):
        lg.append(m.decoder.forward([int(forced[i])], None, cache, start_pos=cache.current_len - 1))
    lg = np.stack(lg)
    err = np.abs(np.take_along_axis(lg, g["forced_top_idx"], 1) - g["forced_top_val"]).max()
    print(f"bf16 encoder + fp32 decoder: max |logit error| {err:.4f} over {len(lg)} positions")
    assert err < 0.03, err  # measured 0.013 over the 25 positions (same fixture, all-bf16 model: 0.031 over 100 positions)
    margins = g["forced_top_val"][:, 0] - g["forced_top_val"][:, 1]
    clear = margins > 0.12
    assert clear.sum() >= 12
    assert np.array_equal(lg.argmax(1)[clear], g["forced_top_idx"][clear, 0])
    got = m.transcribe_batch(mel, max_loop=20, ignore_eot=True)[0]
    assert len(got) == 25 and got[:4] == [50258, 50259, 50359, 50363]


def test_base_dims_fp32(hip, oracle_mod):
    """BASELINE config 5 dims (d_model 512, 8 heads, 6+6 layers, ffn 2048) — the reference cannot run this; a reduced
    context / vocab keeps the oracle fast."""
    from whisper_mojo_amd import WhisperConfig, synth
    cfg = WhisperConfig(512, 8, 6, 3000, 2048, 80, 200, 64)
    w = synth.synth_weights(cfg, 3)
    mel = synth.synth_mel(cfg, 1001)
    ref = oracle_mod.OracleModel(cfg, w)
    m = make_model(cfg, w, max_batch=1)
    assert np.abs(m.encoder.forward(mel) - ref.encode(mel)).max() < F32_TOL
    want, logits = ref.transcribe(mel=mel, prompt=(1, 2, 3, 4), eot=-1, max_loop=16, want_logits=True)
    got = m.transcribe_batch(mel, prompt=(1, 2, 3, 4), eot=-1, max_loop=16)[0]
    _assert_tokens(got, want, logits, 4)


def test_error_behaviour(hip, micro_cfg, micro_weights):
    from whisper_mojo_amd import _lib, synth
    from whisper_mojo_amd.loader import WeightLoader
    from whisper_mojo_amd.whisper import KVCache, Whisper
    m = Whisper(micro_cfg)
    with pytest.raises(_lib.WhisperMiError, match="floats"):
        m.load(WeightLoader.from_array(micro_weights[:-1]))  # the reference would read past the end (loader.mojo:21-27)
    with pytest.raises(_lib.WhisperMiError):
        m.load_file("/nonexistent/whisper_tiny_weights.bin")  # raises like loader.mojo:10-11
    m = make_model(micro_cfg, micro_weights, max_batch=2)
    with pytest.raises(_lib.WhisperMiError, match="max_batch"):
        KVCache(m, 3)
    cache = KVCache(m, 1)
    with pytest.raises(_lib.WhisperMiError, match="encoder"):
        m.decoder.forward([1], None, cache, start_pos=0)
    with pytest.raises(ValueError):
        m.transcribe_batch(np.zeros((1, 3, 5), np.float32))
    mel = synth.synth_mel(micro_cfg, 1000)
    with pytest.raises(_lib.WhisperMiError, match="context"):
        m.transcribe_batch(mel, max_loop=500)


def test_full_size_properties_b64(hip, tiny_cfg, tiny_weights):
    """BASELINE config 3 size (B=64, bf16): size-independent properties — batch results equal single-utterance results
    bit for bit for sampled utterances, every stream has the fixed length, ids are in range, repeat runs are identical."""
    import ctypes as C
    from whisper_mojo_amd import _lib
    L = _lib.lib()
    mels = np.empty((64, 80, 3000), np.float32)
    for i in range(64):
        L.wm_synth_mel_host(1000 + i, 80, 3000, mels[i].ctypes.data_as(C.POINTER(C.c_float)))
    m = make_model(tiny_cfg, tiny_weights, dtype=1, kv=1, max_batch=64)
    a = m.transcribe_batch(mels, max_loop=40, ignore_eot=True)
    b = m.transcribe_batch(mels, max_loop=40, ignore_eot=True)
    assert a == b
    assert all(len(t) == 45 and t[:4] == [50258, 50259, 50359, 50363] and max(t) < 51865 and min(t) >= 0 for t in a)
    for i in (0, 17, 63):
        assert m.transcribe_batch(mels[i], max_loop=40, ignore_eot=True)[0] == a[i]


@pytest.mark.skipif(not os.environ.get("WHISPER_REAL_DIR"), reason="real whisper_tiny_weights.bin + sample_input.bin are "
                    "git-ignored upstream and need network to create; set WHISPER_REAL_DIR to a dir holding them")
@pytest.mark.parametrize("mode", ["ref", "hf"])
def test_real_weights_expected_tokens_hip(hip, tiny_cfg, mode):
    """The reference's one golden (expected_tokens.txt, main.mojo:23-37) on the HIP path — the twin of
    tests/test_oracle_golden.py::test_real_weights_expected_tokens: Whisper.load(whisper_tiny_weights.bin),
    Whisper.transcribe(sample_input.bin) in fp32, in the reference's semantics (tanh GELU, positions current_len - 1) and in
    HF's (erf GELU, positions current_len — the model that WROTE the golden, export_weights.py:125-131); tokens[4:-1] must be
    the 89 ids.  README claims the reference reproduces them (readme.md:19); HF mode must by construction."""
    import re
    from whisper_mojo_amd.loader import WeightLoader
    from whisper_mojo_amd.whisper import Whisper
    d = os.environ["WHISPER_REAL_DIR"]
    ids = [int(x) for x in re.findall(r"\((\d+)\)", open(os.path.join(os.path.dirname(__file__), "golden", "expected_tokens.txt")).read())]
    mel = np.fromfile(os.path.join(d, "sample_input.bin"), np.float32).reshape(80, 3000)
    m = Whisper(tiny_cfg, gelu_mode=0 if mode == "ref" else 1, pos_mode=0 if mode == "ref" else 1, max_batch=1)
    m.load(WeightLoader(os.path.join(d, "whisper_tiny_weights.bin")))
    toks = m.transcribe(mel)
    assert toks[:4] == [50258, 50259, 50359, 50363] and toks[-1] == 50257
    assert toks[4:-1] == ids


def test_ln_matmul_fused_and_refused_shapes(hip):
    """The LayerNorm -> projection pair (layers.mojo:449-455) through wm_op_ln_matmul_nt.  (1) K = 384 with 16-bit operands: the
    one-kernel form (LayerNorm in the row-panel GEMM's A load) equals float64 on the operands it rounds, and the unfused form.
    (2) Shapes the fused kernel does not take — K = 256, fp32 operands, N = 3200 — asked for WITH require_fused are REFUSED by the
    launcher: the C-ABI returns WM_E_ARG with a message, launches nothing, and the output buffer keeps its contents (round 2's
    launchers printed to stderr and returned WM_OK with stale output)."""
    from whisper_mojo_amd import _lib, whisper_tensor as wt
    rng = np.random.default_rng(11)

    def case(M, N, K):
        A = (rng.standard_normal((M, K)) * 1.5 + 0.3).astype(np.float32)
        g = (1 + 0.1 * rng.standard_normal(K)).astype(np.float32)
        b = (0.1 * rng.standard_normal(K)).astype(np.float32)
        W = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
        bias = (rng.standard_normal(N) * 0.1).astype(np.float32)
        return A, g, b, W, bias

    def bf16_round(x):
        u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
        u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
        return u.astype(np.uint32).view(np.float32)

    A, g, b, W, bias = case(300, 768, 384)
    mean = A.mean(1, keepdims=True, dtype=np.float64)
    var = (A.astype(np.float64) ** 2).mean(1, keepdims=True) - mean ** 2
    xn = ((A - mean) / np.sqrt(var + 1e-5) * g + b).astype(np.float32)
    want = bf16_round(xn).astype(np.float64) @ bf16_round(W).astype(np.float64).T + bias
    fused = np.full((300, 768), 7.0, np.float32)
    wt.ln_matmul(fused, A, g, b, W, bias, dtype=1, require_fused=True)
    plain = np.empty_like(fused)
    wt.ln_matmul(plain, A, g, b, W, bias, dtype=1)
    # the LayerNorm itself is fp32 (vs float64 here): a bf16 ulp of an operand flips now and then -> a few 1e-3 on O(1) sums
    assert np.abs(fused - want).max() < 2e-2 and np.abs(plain - want).max() < 2e-2
    assert np.abs(fused - plain).max() < 2e-2
    for (M, N, K, dt, why) in ((64, 256, 256, 1, "K = 256"), (64, 256, 384, 0, "fp32 operands"), (64, 3200, 384, 1, "N > 3072")):
        A, g, b, W, bias = case(M, N, K)
        out = np.full((M, N), 7.0, np.float32)
        with pytest.raises(_lib.WhisperMiError, match="LayerNorm fused into the A load") as ei:
            wt.ln_matmul(out, A, g, b, W, bias, dtype=dt, require_fused=True)
        assert "-1" in str(ei.value).split(":")[0], why  # WM_E_ARG
        assert np.all(out == 7.0), why                    # nothing was written
        wt.ln_matmul(out, A, g, b, W, bias, dtype=dt)     # the same shape without the demand: served by LayerNorm + plain GEMM
        assert np.all(out != 7.0)
    # the skinny path: a K the decode kernel cannot split over its waves is refused, not truncated
    out = np.full((2, 10), 7.0, np.float32)
    with pytest.raises(_lib.WhisperMiError, match="K too large"):
        wt.matmul(out, rng.standard_normal((2, 4096)).astype(np.float32), rng.standard_normal((10, 4096)).astype(np.float32))
    assert np.all(out == 7.0)
