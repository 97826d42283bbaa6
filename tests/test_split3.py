"""The arithmetic claim behind the fp32 logits kernels (kernels_decoder.hip split3 / dec_logits_split*_kernel): every fp32 value is
EXACTLY the sum of three bf16 values taken by truncation (8 + 8 + 8 significant bits), and the six products kept of the nine are
within 2^-21 of the exact product.  numpy restatement of the device code's bit operations; no GPU."""
import numpy as np


def split3(x):
    x = np.ascontiguousarray(x, np.float32)
    uh = x.view(np.uint32) & np.uint32(0xFFFF0000)
    h = uh.view(np.float32)
    r1 = x - h  # exact: the low 16 mantissa bits of x
    um = r1.view(np.uint32) & np.uint32(0xFFFF0000)
    m = um.view(np.float32)
    r2 = r1 - m  # exact: at most 8 significant bits left
    return h, m, r2


def test_three_bf16_terms_reproduce_fp32_exactly():
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.standard_normal(200000).astype(np.float32) * np.float32(10.0) ** rng.integers(-30, 30, 200000).astype(np.float32),
                        np.float32([0.0, -0.0, 1.0, -1.0, 3.4e38, -3.4e38, 1.17549435e-38, 1e-30, np.float32(1) + np.float32(2 ** -23)])])
    h, m, l = split3(x)
    for part in (h, m, l):  # each part is a bf16 value: its low 16 bits are zero
        assert not np.any(part.view(np.uint32) & np.uint32(0xFFFF))
    assert np.array_equal((h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64)).astype(np.float32), x)
    assert np.array_equal(h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64), x.astype(np.float64))  # exactly, not just after rounding


def test_six_kept_products_are_at_fp32_rounding_level():
    rng = np.random.default_rng(4)
    w = (rng.standard_normal(100000) * 0.05).astype(np.float32)
    x = (rng.standard_normal(100000) * 1.5).astype(np.float32)
    (wh, wm, wl), (xh, xm, xl) = split3(w), split3(x)
    f = lambda a: a.astype(np.float64)
    kept = f(wl) * f(xh) + f(wh) * f(xl) + f(wm) * f(xm) + f(wm) * f(xh) + f(wh) * f(xm) + f(wh) * f(xh)
    exact = f(w) * f(x)
    rel = np.abs(kept - exact) / np.maximum(np.abs(exact), 1e-300)
    # dropped: wm*xl and wl*xm (each < 2^-7 * 2^-15 of |w*x|: truncation leaves |m| < 2^-7 |x|, |l| < 2^-15 |x|) and wl*xl (< 2^-30)
    assert rel.max() < 2.0 ** -21 and np.median(rel) < 2.0 ** -24
    # a K = 384 dot product of such terms stays far inside the fp32 parity bar of the logits (5e-5 on O(1) values)
    k = 384
    dots_kept, dots_exact = kept[: 260 * k].reshape(260, k).sum(1), exact[: 260 * k].reshape(260, k).sum(1)
    assert np.abs(dots_kept - dots_exact).max() < 1e-6  # (typical 1e-7: the dropped terms are one-signed per product, random across k)
