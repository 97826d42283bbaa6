"""Host-side mirror of /root/reference/whisper_tensor.mojo: the op functions with the reference's names and
argument meaning (out-param first), each a thin call through the C-ABI into a HIP kernel.  `Tensor` is a
row-major fp32 numpy array [rows, cols] (whisper_tensor.mojo:10-15)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .config import DT_F32, GELU_TANH


def Tensor(rows: int, cols: int) -> np.ndarray:
    """whisper_tensor.mojo:17-23: zero-filled [rows, cols] fp32."""
    return np.zeros((rows, cols), np.float32)


def _fp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))


def _chk_out(out, shape):
    if out.dtype != np.float32 or not out.flags.c_contiguous or out.shape != tuple(shape):
        raise ValueError(f"out must be a C-contiguous float32 array of shape {tuple(shape)}")


def matmul(C_out: np.ndarray, A, B, bias=None, dtype=DT_F32):
    """whisper_tensor.mojo:151-246: C = A·Bᵀ (+bias); B in HF [out,in] layout."""
    A, B = np.ascontiguousarray(A, np.float32), np.ascontiguousarray(B, np.float32)
    b = None if bias is None or bias.size == 0 else np.ascontiguousarray(bias, np.float32).ravel()
    M, K = A.shape
    N = B.shape[0]
    _chk_out(C_out, (M, N))
    _lib.check(_lib.lib().wm_op_matmul_nt(_fp(C_out), _fp(A), _fp(B), _fp(b), M, N, K, dtype))


def ln_matmul(C_out: np.ndarray, A, ln_g, ln_b, B, bias=None, dtype=DT_F32, require_fused: bool = False):
    """layers.mojo:449-455 / 489-497: C = layer_norm(A)·Bᵀ (+bias).  require_fused: demand the one-kernel form (the LayerNorm applied
    while the GEMM loads A); a shape that kernel does not take raises (WM_E_ARG) and leaves C_out untouched."""
    f = lambda a: np.ascontiguousarray(a, np.float32)
    A, B, ln_g, ln_b = f(A), f(B), f(ln_g).ravel(), f(ln_b).ravel()
    b = None if bias is None or bias.size == 0 else f(bias).ravel()
    M, K = A.shape
    N = B.shape[0]
    _chk_out(C_out, (M, N))
    _lib.check(_lib.lib().wm_op_ln_matmul_nt(_fp(C_out), _fp(A), _fp(ln_g), _fp(ln_b), _fp(B), _fp(b), M, N, K, dtype, int(require_fused)))


def mlp_block(x: np.ndarray, ln_g, ln_b, fc1_w, fc1_b, fc2_w, fc2_b, next_ln=None, dtype=DT_F32, gelu_mode: int = GELU_TANH):
    """layers.mojo:489-517 (the MLP half of ResidualAttentionBlock.forward): x += fc2(gelu(fc1(layer_norm(x)))), in place.
    next_ln = (gamma, beta): also returns layer_norm(x_new) rounded to the operand dtype — the rows the next projection reads."""
    _chk_out(x, x.shape)
    M, d = x.shape
    f = lambda a: np.ascontiguousarray(a, np.float32)
    ln_g, ln_b, fc1_w, fc1_b, fc2_w, fc2_b = f(ln_g).ravel(), f(ln_b).ravel(), f(fc1_w), f(fc1_b).ravel(), f(fc2_w), f(fc2_b).ravel()
    ffn = fc1_w.shape[0]
    if fc1_w.shape != (ffn, d) or fc2_w.shape != (d, ffn):
        raise ValueError("fc1_w must be [ffn, d] and fc2_w [d, ffn]")
    ng = nb = xn = None
    if next_ln is not None:
        ng, nb = f(next_ln[0]).ravel(), f(next_ln[1]).ravel()
        xn = np.empty((M, d), np.float32)
    _lib.check(_lib.lib().wm_op_mlp_block(_fp(x), _fp(ln_g), _fp(ln_b), _fp(fc1_w), _fp(fc1_b), _fp(fc2_w), _fp(fc2_b), _fp(ng), _fp(nb),
                                          _fp(xn), M, d, ffn, dtype, gelu_mode))
    return xn


def attention(out: np.ndarray, q, k, v, n_heads: int, dtype=DT_F32):
    """layers.mojo:273-342, the block path without cache or mask (the encoder's): per head softmax(q_h·k_hᵀ / 8)·v_h."""
    f = lambda a: np.ascontiguousarray(a, np.float32)
    q, k, v = f(q), f(k), f(v)
    n_ctx, d = q.shape
    if d != 64 * n_heads or k.shape != q.shape or v.shape != q.shape:
        raise ValueError("q, k, v must be [n_ctx, 64 * n_heads]")
    _chk_out(out, q.shape)
    _lib.check(_lib.lib().wm_op_attention(_fp(out), _fp(q), _fp(k), _fp(v), n_ctx, n_heads, dtype))


def attention_cached(out: np.ndarray, q, k, v, n_heads: int, kv_dtype=DT_F32, n_chunks: int = 1):
    """layers.mojo:186-272, the q_len == 1 path over cached rows: q [B, d], k / v [B, t, d] -> out [B, d].
    n_chunks > 1: the cross-attention form (keys swept by several workgroups and merged)."""
    f = lambda a: np.ascontiguousarray(a, np.float32)
    q, k, v = f(q), f(k), f(v)
    B, d = q.shape
    if d != 64 * n_heads or k.ndim != 3 or k.shape[0] != B or k.shape[2] != d or v.shape != k.shape:
        raise ValueError("q must be [B, 64 * n_heads], k and v [B, t, 64 * n_heads]")
    _chk_out(out, q.shape)
    _lib.check(_lib.lib().wm_op_attention_cached(_fp(out), _fp(q), _fp(k), _fp(v), B, k.shape[1], n_heads, kv_dtype, n_chunks))


def layer_norm(out: np.ndarray, inp, gamma, beta, eps: float = 1e-5):
    """whisper_tensor.mojo:249-285"""
    x = np.ascontiguousarray(inp, np.float32)
    g, b = np.ascontiguousarray(gamma, np.float32).ravel(), np.ascontiguousarray(beta, np.float32).ravel()
    _chk_out(out, x.shape)
    _lib.check(_lib.lib().wm_op_layer_norm(_fp(out), _fp(x), _fp(g), _fp(b), x.shape[0], x.shape[1], eps))


def gelu(t: np.ndarray, mode: int = GELU_TANH):
    """whisper_tensor.mojo:288-308 — in place."""
    _chk_out(t, t.shape)
    _lib.check(_lib.lib().wm_op_gelu(_fp(t), t.size, mode))


def softmax(t: np.ndarray):
    """whisper_tensor.mojo:311-355 — rows, in place."""
    _chk_out(t, t.shape)
    _lib.check(_lib.lib().wm_op_softmax_rows(_fp(t), t.shape[0], t.shape[1]))


def conv1d(out: np.ndarray, inp, weight, bias, stride: int, padding: int = 1, out_T: bool = False, dtype=DT_F32):
    """whisper_tensor.mojo:367-428 (K=3).  `weight` is the file-layout [C_out, C_in, 3] tensor; the reference's
    transpose_conv_weights (:358-364) re-layout happens inside the library."""
    if padding != 1:
        raise ValueError("the reference only ever uses padding=1 (whisper.mojo:74,79)")
    x, w = np.ascontiguousarray(inp, np.float32), np.ascontiguousarray(weight, np.float32)
    b = np.ascontiguousarray(bias, np.float32).ravel()
    C_in, L_in = x.shape
    C_out = w.shape[0]
    L_out = (L_in + 2 - 3) // stride + 1
    _chk_out(out, (L_out, C_out) if out_T else (C_out, L_out))
    _lib.check(_lib.lib().wm_op_conv1d_k3(_fp(out), _fp(x), _fp(w), _fp(b), C_in, L_in, C_out, stride, int(out_T), dtype))


def argmax(t) -> int:
    """whisper_tensor.mojo:431-439: lowest index wins ties."""
    x = np.ascontiguousarray(t, np.float32).ravel()
    idx = C.c_int32(0)
    _lib.check(_lib.lib().wm_op_argmax(_fp(x), x.size, C.byref(idx)))
    return int(idx.value)
