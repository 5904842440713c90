"""Representation error of the fp32-GEMM-on-the-matrix-pipe schemes, emulated exactly in numpy (CPU):

  bf16 x 3 planes, 6 products   (shipped: csrc/gemm_split.hip, range-safe by construction)
  fp16 x 2 planes, 3 products   with a power-of-two scale per A row and per W row that puts the
                                row maximum in [2^14, 2^15)  (layout 2 of aps_linear_split)
  fp16 x 2 planes, 3 products   unscaled (what scripts/split_probe.py measured on the GPU)

Every product of two plane values is exact in float64, so summing them in float64 isolates what the
split itself loses (rounding of the planes + dropped cross terms) from the fp32 accumulation error
that all forms, and the fp32 MFMA, share.  Printed: max and rms error of C = A W^T in units of the
rms of C, next to the error of a plain fp32 evaluation (float32 products, pairwise float32 sums).

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


def gemm_fp16x3(a, w, scaled):
    ea = row_exponent(a) if scaled else None
    ew = row_exponent(w) if scaled else None
    ah, al = planes_fp16(a, ea)
    wh, wl = planes_fp16(w, ew)
    c = ah @ wl.T + al @ wh.T + ah @ wh.T
    if scaled:
        c = np.ldexp(c, -(ea[:, None] + ew[None, :]).astype(np.int32))
    return c


def gemm_f32(a, w):
    prod = a.astype(np.float32)[:, None, :] * w.astype(np.float32)[None, :, :]
    return np.sum(prod, axis=2, dtype=np.float32).astype(np.float64)  # numpy: pairwise fp32 sums


def report(name, a, w):
    ref = a.astype(np.float64) @ w.astype(np.float64).T
    scale = np.sqrt(np.mean(ref ** 2))
    out = [f"{name:34s}"]
    for label, fn in (("f32", lambda: gemm_f32(a, w)), ("bf16x6", lambda: gemm_bf16x6(a, w)),
                      ("fp16x3 scaled", lambda: gemm_fp16x3(a, w, True)),
                      ("fp16x3 raw", lambda: gemm_fp16x3(a, w, False))):
        with np.errstate(invalid="ignore", over="ignore"):
            c = fn()
        err = np.abs(c - ref) / scale
        mx = np.nanmax(err) if np.isfinite(c).all() else float("inf")
        out.append(f"{label} {mx:8.1e}/{np.sqrt(np.nanmean(err ** 2)):8.1e}")
    print("  ".join(out), flush=True)


def main():
    rng = np.random.default_rng(0)
    M, N = 96, 64
    for K in (512, 2048):
        print(f"K = {K}   (max / rms error in units of rms(C))")
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


if __name__ == "__main__":
    main()
