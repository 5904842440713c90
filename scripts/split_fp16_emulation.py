"""Representation error of the fp32-GEMM-on-the-matrix-pipe schemes, emulated exactly in numpy (CPU):

  bf16 x 3 planes, 6 products   (csrc/gemm_split.hip, range-safe by construction)
  fp16 x 2 planes, 3 products   the shipped form (csrc/gemm_fp16x2.hip, round 3): a power-of-two scale
                                per A row and per W row puts the row maximum in [2^14, 2^15), the low
                                plane holds the residue times 2^11, the cross terms have their own
                                accumulator, and rows with an element the planes cannot hold (non-zero
                                and more than 2^30 below the row maximum) are recomputed in fp32
  fp16 x 2 planes, round 2      the same scale but an UNSCALED low plane in one accumulator: loses the
                                low plane of every element more than 2^17 below its row maximum
                                (kept here to show why it was replaced: `outlier_zero_weight_case`)
  fp16 x 2 planes, unscaled     (what scripts/split_probe.py measured on the GPU)

Every product of two plane values is exact in float64, so summing them in float64 isolates what the
split itself loses (rounding of the planes + dropped cross terms) from the fp32 accumulation error
that all forms, and the fp32 MFMA, share.  Printed: max and rms error of C = A W^T in units of the
rms of C, next to the error of a plain fp32 evaluation (float32 products, pairwise float32 sums), and
the component-wise figure log2 max |C - C_exact| / sum |a| |w|.

    python scripts/split_fp16_emulation.py
"""
import numpy as np


def bf16_trunc(x):
    u = x.astype(np.float32).view(np.uint32) & np.uint32(0xFFFF0000)
    return u.view(np.float32)


def planes_bf16(x):
    h = bf16_trunc(x)
    r = (x - h).astype(np.float32)
    m = bf16_trunc(r)
    l = bf16_trunc((r - m).astype(np.float32))
    return h, m, l


def row_exponent(x):
    """e with max|row| * 2^e in [2^14, 2^15); zero / subnormal rows as the smallest normal"""
    mx = np.max(np.abs(x), axis=1).astype(np.float32)
    be = (mx.view(np.uint32) >> 23) & 0xFF
    be = np.clip(be, 1, 254).astype(np.int64)
    return 141 - be


def planes_fp16(x, e=None):
    xs = x.astype(np.float32) if e is None else np.ldexp(x.astype(np.float32), e[:, None].astype(np.int32))
    with np.errstate(over="ignore"):
        h = xs.astype(np.float16)                       # round to nearest even
        l = (xs - h.astype(np.float32)).astype(np.float16)
    return h.astype(np.float64), l.astype(np.float64)


def gemm_bf16x6(a, w):
    ah, am, al = (p.astype(np.float64) for p in planes_bf16(a))
    wh, wm, wl = (p.astype(np.float64) for p in planes_bf16(w))
    return am @ wm.T + ah @ wl.T + al @ wh.T + ah @ wm.T + am @ wh.T + ah @ wh.T


LOW_SHIFT = 11       # the low plane holds (x' - h) 2^11   (gemm_fp16x2.hip: kLowShift)
FIT_LO, FIT_HI = -16, 15  # a scaled element fits iff it is 0 or 2^-16 <= |x'| < 2^15 (kFitBias / kFitMax)


def planes_fp16_low_scaled(x, e):
    """(h, l, wide): h = rn_f16(x'), l = rn_f16((x' - h) 2^11), wide[row] = some element does not fit"""
    xs = np.ldexp(x.astype(np.float32), e[:, None].astype(np.int32))
    with np.errstate(over="ignore", invalid="ignore"):
        h = xs.astype(np.float16)
        r = (xs - h.astype(np.float32)).astype(np.float32)
        l = (r * np.float32(2.0 ** LOW_SHIFT)).astype(np.float16)
    mag = np.abs(xs)
    wide = (((mag < 2.0 ** FIT_LO) & (xs != 0)) | ((mag >= 2.0 ** FIT_HI) & np.isfinite(xs))).any(axis=1)
    return h.astype(np.float64), l.astype(np.float64), wide


def gemm_fp16x3(a, w, scaled=True, guard=True, return_wide=False):
    """the shipped arithmetic (scaled=True): main + 2^-11 cross, rows / columns with an element that
    does not fit recomputed as a plain fp32 evaluation (guard).  scaled=False: raw planes, no scale."""
    if not scaled:
        ah, al = planes_fp16(a, None)
        wh, wl = planes_fp16(w, None)
        return ah @ wl.T + al @ wh.T + ah @ wh.T
    ea, ew = row_exponent(a), row_exponent(w)
    ah, al, wa = planes_fp16_low_scaled(a, ea)
    wh, wl, ww = planes_fp16_low_scaled(w, ew)
    main = ah @ wh.T
    cross = ah @ wl.T + al @ wh.T
    c = np.ldexp(main + cross * 2.0 ** -LOW_SHIFT, -(ea[:, None] + ew[None, :]).astype(np.int32))
    if guard and (wa.any() or ww.any()):
        c32 = gemm_f32(a, w)
        c[wa, :] = c32[wa, :]
        c[:, ww] = c32[:, ww]
    return (c, wa, ww) if return_wide else c


def gemm_fp16x3_chunked(a, w, chunk=256, guard=True, return_wide=False):
    """the PANEL form (csrc/gemm_panel.hip, round 4): the same planes with a power of two per A row
    AND K chunk -- each chunk's (main + 2^-11 cross) is brought back with its own exponent and the
    chunks meet in an fp32 sum (emulated in float64 like the accumulators); a row is `wide` when any
    of its chunks holds an element the planes cannot hold; W keeps one exponent per row"""
    M, K = a.shape
    ew = row_exponent(w)
    wh, wl, ww = planes_fp16_low_scaled(w, ew)
    c = np.zeros((M, w.shape[0]))
    wa = np.zeros(M, bool)
    for k0 in range(0, K, chunk):
        ac = a[:, k0:k0 + chunk]
        ea = row_exponent(ac)
        ah, al, wide = planes_fp16_low_scaled(ac, ea)
        wa |= wide
        main = ah @ wh[:, k0:k0 + chunk].T
        cross = ah @ wl[:, k0:k0 + chunk].T + al @ wh[:, k0:k0 + chunk].T
        c += np.ldexp(main + cross * 2.0 ** -LOW_SHIFT, -(ea[:, None] + ew[None, :]).astype(np.int32))
    if guard and (wa.any() or ww.any()):
        c32 = gemm_f32(a, w)
        c[wa, :] = c32[wa, :]
        c[:, ww] = c32[:, ww]
    return (c, wa, ww) if return_wide else c


def gemm_fp16x3_round2(a, w):
    """round 2's form: unscaled low plane, one accumulator, no guard"""
    ea, ew = row_exponent(a), row_exponent(w)
    ah, al = planes_fp16(a, ea)
    wh, wl = planes_fp16(w, ew)
    c = ah @ wl.T + al @ wh.T + ah @ wh.T
    return np.ldexp(c, -(ea[:, None] + ew[None, :]).astype(np.int32))


def outlier_zero_weight_case(rng, M, N, K, in_row_range):
    """A ~ 1e-3 N(0,1) with one column `in_row_range` times larger, meeting a ZERO weight column: the
    row maximum then dominates no output, and whatever the split loses on the small elements shows"""
    a = (1e-3 * rng.standard_normal((M, K))).astype(np.float32)
    a[:, 3] = (1e-3 * in_row_range * np.sign(rng.standard_normal(M))).astype(np.float32)
    w = rng.standard_normal((N, K)).astype(np.float32)
    w[:, 3] = 0
    return a, w


def componentwise_log2(c, a, w):
    ref = a.astype(np.float64) @ w.astype(np.float64).T
    bound = np.abs(a).astype(np.float64) @ np.abs(w).astype(np.float64).T
    with np.errstate(divide="ignore", invalid="ignore"):
        q = np.abs(c - ref) / bound
    q = q[bound > 0]
    return float(np.log2(q.max())) if q.size and q.max() > 0 else -np.inf


def gemm_f32(a, w):
    prod = a.astype(np.float32)[:, None, :] * w.astype(np.float32)[None, :, :]
    return np.sum(prod, axis=2, dtype=np.float32).astype(np.float64)  # numpy: pairwise fp32 sums


def report(name, a, w):
    ref = a.astype(np.float64) @ w.astype(np.float64).T
    scale = np.sqrt(np.mean(ref ** 2))
    out = [f"{name:34s}"]
    for label, fn in (("f32", lambda: gemm_f32(a, w)), ("bf16x6", lambda: gemm_bf16x6(a, w)),
                      ("fp16x3", lambda: gemm_fp16x3(a, w)),
                      ("fp16x3 chunked", lambda: gemm_fp16x3_chunked(a, w)),
                      ("fp16x3 r2", lambda: gemm_fp16x3_round2(a, w)),
                      ("fp16x3 raw", lambda: gemm_fp16x3(a, w, False))):
        with np.errstate(invalid="ignore", over="ignore"):
            c = fn()
        err = np.abs(c - ref) / scale
        mx = np.nanmax(err) if np.isfinite(c).all() else float("inf")
        cw = componentwise_log2(c, a, w) if np.isfinite(c).all() else float("inf")
        out.append(f"{label} {mx:8.1e}/{np.sqrt(np.nanmean(err ** 2)):8.1e} 2^{cw:6.1f}")
    print("  ".join(out), flush=True)


def main():
    rng = np.random.default_rng(0)
    M, N = 96, 64
    for K in (512, 2048):
        print(f"K = {K}   (max / rms error in units of rms(C), log2 of max |err| / sum |a||w|)")
        g = lambda *s: rng.standard_normal(s).astype(np.float32)
        report("N(0,1) x N(0,1)", g(M, K), g(N, K))
        report("3e3 N(0,1) x N(0,1)", 3e3 * g(M, K), g(N, K))
        report("1e-4 N(0,1) x N(0,1)", 1e-4 * g(M, K), g(N, K))
        rs = np.exp(rng.uniform(np.log(1e-6), np.log(1e6), (M, 1))).astype(np.float32)
        report("rows of scale 1e-6 .. 1e6", rs * g(M, K), g(N, K))
        report("lognormal(0, 3) elements", np.exp(3 * g(M, K)) * np.sign(g(M, K)), g(N, K))
        a = g(M, K)
        a[:, 7] *= 1e4
        report("one 1e4 outlier column in A", a, g(N, K))
        report("layer-norm-like: 20 + N(0,1)", 20 + g(M, K), g(N, K) / np.sqrt(K))
        report("1e30 N(0,1) x 1e-30 N(0,1)", 1e30 * g(M, K), 1e-30 * g(N, K))
        report("weights N(0, 0.02)", g(M, K), 0.02 * g(N, K))
        for rg in (1e5, 1e6, 1e7, 1e8, 1e9, 1e10):
            a, w = outlier_zero_weight_case(rng, M, N, K, rg)
            _, wa, _ = gemm_fp16x3(a, w, return_wide=True)
            report(f"outlier x zero weight, range {rg:.0e} ({int(wa.sum())} rows in fp32)", a, w)


if __name__ == "__main__":
    main()
