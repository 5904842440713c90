"""The arithmetic of the fp16 two-plane GEMM (aps_amd/csrc/gemm_fp16x2.hip), emulated exactly on the
CPU (scripts/split_fp16_emulation.py: plane products are exact in float64).  With a power-of-two scale
per operand row, the low plane carrying the residue times 2^11, the cross terms in their own
accumulator and the rows the planes cannot hold recomputed in fp32, every output lies within
2^-20.5 of sum |a| |w| -- on well-scaled operands, on rows 12 orders of magnitude apart, with an
outlier column (also when that column meets a ZERO weight column, the case that broke round 2's
single-accumulator form), at the edges of the fp32 range; elements that are not recomputed keep a
relative error of at most 2^-19 each.  The same planes WITHOUT the row scale fail (why it exists).
No GPU."""
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
        # the panel form (gemm_panel.hip): a power of two per row AND K chunk of 256 / 128
        for chunk in (256, 128):
            assert (np.abs(emu.gemm_fp16x3_chunked(a, w, chunk) - ref) <= 2.0 ** -20.5 * bound).all(), (name, chunk)
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


@pytest.mark.parametrize("in_row_range", [1e5, 1e6, 1e7, 1e8, 1e9, 1e10])
def test_outlier_column_meeting_a_zero_weight_column(in_row_range):
    """the round-2 verdict's counter-example: the row maximum meets a zero weight, so it bounds no
    output and the error of the small elements shows.  The shipped form holds the component-wise
    bound at every in-row range (rows whose small elements leave the planes' range are recomputed in
    fp32, and they are exactly the rows it flags); round 2's form does not (kept as the witness)."""
    rng = np.random.default_rng(int(np.log10(in_row_range)))
    a, w = emu.outlier_zero_weight_case(rng, 48, 40, 512, in_row_range)
    c, wide_a, wide_w = emu.gemm_fp16x3(a, w, return_wide=True)
    assert emu.componentwise_log2(c, a, w) <= -20.5
    assert not wide_w.any()
    if in_row_range >= 1e8:
        assert wide_a.all()          # small elements below 2^-16 after scaling: every row recomputed
    if in_row_range <= 1e5:
        # nearly everything inside the planes (a Gaussian row may hold a chance value near zero)
        assert wide_a.mean() <= 0.1
        assert emu.componentwise_log2(emu.gemm_fp16x3(a, w, guard=False), a, w) <= -19.0
    if in_row_range >= 1e7:
        assert emu.componentwise_log2(emu.gemm_fp16x3_round2(a, w), a, w) > -20.5   # what round 2 shipped
    # the panel form: only the chunk that holds the outlier column can be out of range, and a row it
    # flags there is recomputed whole -- same bound, never more rows than the whole-row rule flags
    cc, wide_c, _ = emu.gemm_fp16x3_chunked(a, w, 256, return_wide=True)
    assert emu.componentwise_log2(cc, a, w) <= -20.5
    assert not (wide_c & ~wide_a).any()


def test_elements_the_guard_lets_through_keep_2_pow_minus_19():  # (each within 2^-20)
    """worst case for an element that is NOT recomputed: just above 2^-30 of its row maximum, the only
    element its weight column looks at.  Its relative error is bounded by 2^-36 / 2^-16 = 2^-20 (+ the
    weight's own 2^-22 + the dropped l l term: 2^-19 is the bound the kernel is held to)."""
    rng = np.random.default_rng(11)
    M, N, K = 32, 16, 64
    a = np.zeros((M, K), np.float32)
    a[:, 0] = 3.0e4 * (1 + rng.random(M))                       # the row maximum
    a[:, 1:] = (a[:, :1].astype(np.float64) * 2.0 ** -29.5 * (1 + 0.4 * rng.random((M, K - 1)))).astype(np.float32)
    w = np.zeros((N, K), np.float32)
    w[:, 1:] = rng.standard_normal((N, K - 1)).astype(np.float32)   # nothing looks at column 0
    c, wide_a, _ = emu.gemm_fp16x3(a, w, return_wide=True)
    assert not wide_a.any()                                      # inside the guard's range
    assert emu.componentwise_log2(c, a, w) <= -19.0
    # one binade further down the guard fires and the rows are exact fp32 again
    a2 = a.copy()
    a2[:, 1:] *= np.float32(0.25)
    c2, wide2, _ = emu.gemm_fp16x3(a2, w, return_wide=True)
    assert wide2.all() and emu.componentwise_log2(c2, a2, w) <= -22.0


def test_fit_rule_edges():
    """0 fits; 2^-16 <= |x'| < 2^15 fits; anything else (a stale row-maximum hint: >= 2^15) does not;
    inf / NaN are not the guard's business (they propagate like in fp32)"""
    e = np.zeros(1, np.int64)
    for v, want in ((0.0, False), (2.0 ** -16, False), (np.nextafter(np.float32(2.0 ** -16), 0), True),
                    (-2.0 ** -20, True), (2.0 ** 15, True), (np.nextafter(np.float32(2.0 ** 15), 0), False),
                    (np.inf, False), (np.nan, False)):
        x = np.array([[1.0, v]], np.float32)
        assert bool(emu.planes_fp16_low_scaled(x, e)[2][0]) == want, v
