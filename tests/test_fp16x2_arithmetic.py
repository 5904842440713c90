"""The arithmetic of the fp16 two-plane GEMM (aps_amd/csrc/gemm_fp16x2.hip), emulated exactly on the
CPU (scripts/split_fp16_emulation.py: plane products are exact in float64): with a power-of-two scale
per operand row the three-product form loses no more than a plain fp32 evaluation does -- on
well-scaled operands, on rows 12 orders of magnitude apart, with an outlier column, at the edges of
the fp32 range; on heavy-tailed elements, where single products dominate an output, it is within
1.5 x of it -- every output within 2^-20.5 of sum |a| |w| -- and the same planes WITHOUT the scale
are not (why the scale exists).  No GPU."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "scripts"))
import split_fp16_emulation as emu  # noqa: E402


def _cases(rng, M, N, K):
    g = lambda *s: rng.standard_normal(s).astype(np.float32)
    rows = np.exp(rng.uniform(np.log(1e-6), np.log(1e6), (M, 1))).astype(np.float32)
    a_out = g(M, K)
    a_out[:, 7] *= 1e4
    return {
        "normal": (g(M, K), g(N, K)),
        "tiny": (1e-4 * g(M, K), g(N, K)),
        "row scales 1e-6..1e6": (rows * g(M, K), g(N, K)),
        "lognormal(0,3)": (np.exp(3 * g(M, K)) * np.sign(g(M, K)), g(N, K)),
        "outlier column": (a_out, g(N, K)),
        "offset rows": (20 + g(M, K), g(N, K) / np.sqrt(K)),
        "range edges": (1e30 * g(M, K), 1e-30 * g(N, K)),
    }


def _err(c, ref):
    return np.abs(c - ref) / np.sqrt(np.mean(ref ** 2))


@pytest.mark.parametrize("K", [256, 1024])
def test_scaled_two_plane_product_is_as_accurate_as_fp32(K):
    rng = np.random.default_rng(K)
    for name, (a, w) in _cases(rng, 48, 40, K).items():
        ref = a.astype(np.float64) @ w.astype(np.float64).T
        e32 = _err(emu.gemm_f32(a, w), ref)
        e16 = _err(emu.gemm_fp16x3(a, w, True), ref)
        e6 = _err(emu.gemm_bf16x6(a, w), ref)
        assert np.isfinite(e16).all(), name
        # rms against the plain fp32 evaluation (which carries the accumulation error the emulated
        # split forms do not: the margin of the kernel is larger than this comparison shows); the
        # maximum of a single draw is noisy on heavy-tailed operands (an output dominated by ONE
        # product sees that product's 2^-21 representation error where fp32 rounds once): over many
        # draws the maxima are equal (scripts/split_fp16_emulation.py), here only bounded
        heavy = name == "lognormal(0,3)"  # (outputs dominated by single products: 2^-22 vs one 2^-24 rounding)
        assert np.sqrt(np.mean(e16 ** 2)) <= (1.5 if heavy else 1.05) * np.sqrt(np.mean(e32 ** 2)), name
        assert e16.max() <= 4e-6 and e6.max() <= 4e-6, name
        # component-wise: every output within 2^-20.5 of sum |a| |w| (fp32 itself: 2^-21.0 measured)
        bound = np.abs(a).astype(np.float64) @ np.abs(w).astype(np.float64).T
        assert (np.abs(emu.gemm_fp16x3(a, w, True) - ref) <= 2.0 ** -20.5 * bound).all(), name
        assert (np.abs(emu.gemm_bf16x6(a, w) - ref) <= 2.0 ** -20.5 * bound).all(), name


def test_unscaled_planes_fail_outside_the_fp16_range():
    rng = np.random.default_rng(3)
    cases = _cases(rng, 48, 40, 256)
    a, w = cases["tiny"]
    ref = a.astype(np.float64) @ w.astype(np.float64).T
    with np.errstate(over="ignore", invalid="ignore"):
        assert _err(emu.gemm_fp16x3(a, w, False), ref).max() > 1e-4      # the 1e-4 parity bar is gone
        a, w = cases["row scales 1e-6..1e6"]
        assert not np.isfinite(emu.gemm_fp16x3(a, w, False)).all()       # overflow


def test_row_exponent_rule():
    """141 - biased exponent, clamped: the row maximum lands in [2^14, 2^15); zero / subnormal rows
    count as the smallest normal, inf as the largest finite (gemm_fp16x2.hip:scale_exponent)"""
    x = np.array([[1.0, -3.0], [0.0, 0.0], [1e-45, 0.0], [6e4, 7e4], [np.inf, 1.0], [2.0 ** -126, 0],
                  [3.4e38, 1.0]], dtype=np.float32)
    e = emu.row_exponent(x)
    mx = np.max(np.abs(x), axis=1)
    for i in (0, 3, 5, 6):
        scaled = np.ldexp(np.float64(mx[i]), int(e[i]))
        assert 2.0 ** 14 <= scaled < 2.0 ** 15
    assert e[1] == 140 and e[2] == 140      # zero and subnormal rows
    assert e[4] == 141 - 254                # inf: the exponent of the largest finite
