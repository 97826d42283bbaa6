"""Known-answer tests of the oracle's op restatements (whisper_tensor.mojo) against independent float64 numpy."""
import numpy as np
import pytest

from oracle import oracle

rng = np.random.default_rng(0)


@pytest.mark.parametrize("M,N,K", [(1, 37, 384), (4, 16, 70), (5, 9, 64), (33, 40, 100), (2, 3, 5)])
def test_matmul(M, N, K):
    A, B, b = rng.standard_normal((M, K), np.float32), rng.standard_normal((N, K), np.float32), rng.standard_normal(N, np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64).T
    assert np.abs(oracle.matmul(A, B) - ref).max() < 1e-4
    assert np.abs(oracle.matmul(A, B, b) - (ref + b)).max() < 1e-4


def test_layer_norm_one_pass_variance():
    x = rng.standard_normal((7, 128), np.float32) * 3 + 1
    g, b = rng.standard_normal(128, np.float32), rng.standard_normal(128, np.float32)
    x64 = x.astype(np.float64)
    mean = x64.mean(1, keepdims=True)
    var = (x64 * x64).mean(1, keepdims=True) - mean * mean
    ref = (x64 - mean) / np.sqrt(var + 1e-5) * g + b
    assert np.abs(oracle.layer_norm(x, g, b) - ref).max() < 1e-4


def test_gelu_tanh_vs_erf_and_tail():
    x = np.linspace(-4, 4, 67, dtype=np.float32)  # 67 % 8 = 3: the reference leaves the tail untouched
    t = oracle.gelu(x, 0)
    x64 = x.astype(np.float64)
    ref = 0.5 * x64 * (1 + np.tanh(0.79788456 * (x64 + 0.044715 * x64 ** 3)))
    assert np.abs(t[:64] - ref[:64]).max() < 1e-6
    assert np.array_equal(t[64:], x[64:])
    e = oracle.gelu(x, 1)
    assert 1e-4 < np.abs(e[:64] - t[:64]).max() < 1e-2


@pytest.mark.parametrize("cols", [3, 8, 13, 1500])
def test_softmax(cols):
    x = rng.standard_normal((5, cols), np.float32) * 4
    x64 = x.astype(np.float64)
    e = np.exp(x64 - x64.max(1, keepdims=True))
    assert np.abs(oracle.softmax(x) - e / e.sum(1, keepdims=True)).max() < 1e-6


@pytest.mark.parametrize("stride,out_T", [(1, False), (2, True), (2, False), (1, True)])
def test_conv1d(stride, out_T):
    C_in, L, C_out = 16, 50, 24
    x, w, b = rng.standard_normal((C_in, L), np.float32), rng.standard_normal((C_out, C_in, 3), np.float32), rng.standard_normal(C_out, np.float32)
    wT = oracle.transpose_conv_weights(w)
    assert np.array_equal(wT.reshape(C_out, 3, C_in), w.transpose(0, 2, 1))
    out = oracle.conv1d(x, wT, b, stride, 1, out_T)
    xp = np.pad(x.astype(np.float64), ((0, 0), (1, 1)))
    L_out = (L + 2 - 3) // stride + 1
    ref = np.stack([np.einsum("oik,ik->o", w.astype(np.float64), xp[:, t * stride:t * stride + 3]) for t in range(L_out)], 1) + b[:, None]
    got = out.T if out_T else out
    assert got.shape == ref.shape and np.abs(got - ref).max() < 1e-4


def test_argmax_lowest_index_wins():
    x = np.zeros(100, np.float32)
    x[[17, 40, 99]] = 3.0
    assert oracle.argmax(x) == 17
