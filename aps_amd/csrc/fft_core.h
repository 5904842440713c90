// Host/device math shared by the STFT kernels.  Everything here is plain C++ so the index math
// and butterflies can be unit-tested on the CPU (tests/csrc/test_fft_core.cc) before they ever
// run on a wavefront.
#ifndef APS_AMD_FFT_CORE_H_
#define APS_AMD_FFT_CORE_H_

#if defined(__HIPCC__)
#define APS_HD __host__ __device__ __forceinline__
#else
#define APS_HD inline
#endif

namespace aps {

struct cf {
  float re, im;
};

APS_HD cf operator+(cf a, cf b) { return {a.re + b.re, a.im + b.im}; }
APS_HD cf operator-(cf a, cf b) { return {a.re - b.re, a.im - b.im}; }
APS_HD cf cmul(cf a, cf b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
APS_HD cf cconj(cf a) { return {a.re, -a.im}; }
APS_HD cf cscale(cf a, float s) { return {a.re * s, a.im * s}; }

// cos / sin of 2*pi*m/16, m = 0..15
#define APS_C16                                                                                  \
  {1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f, 0.0f,                 \
   -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f, -1.0f,                   \
   -0.92387953251128674f, -0.70710678118654752f, -0.38268343236508977f, 0.0f,                    \
   0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f}
#define APS_S16                                                                                  \
  {0.0f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f, 1.0f,                 \
   0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f, 0.0f,                       \
   -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f, -1.0f,                   \
   -0.92387953251128674f, -0.70710678118654752f, -0.38268343236508977f}

// 4-point DFT, natural order in / natural order out.  INV selects exp(+i..).
template <bool INV>
APS_HD void dft4(cf& a0, cf& a1, cf& a2, cf& a3) {
  cf t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, t3 = a1 - a3;
  // forward: -i * t3 ; inverse: +i * t3
  cf r = INV ? cf{-t3.im, t3.re} : cf{t3.im, -t3.re};
  a0 = t0 + t2;
  a2 = t0 - t2;
  a1 = t1 + r;
  a3 = t1 - r;
}

// 16-point DFT held in registers: x[16] natural order -> natural order (4 x 4 Cooley-Tukey).
// n = 4*n1 + n2, k = k1 + 4*k2.
template <bool INV>
APS_HD void dft16(cf (&x)[16]) {
  constexpr float c16[16] = APS_C16;
  constexpr float s16[16] = APS_S16;
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2) dft4<INV>(x[n2], x[4 + n2], x[8 + n2], x[12 + n2]);
  // now x[4*k1 + n2] = sum_n1 x[4 n1 + n2] W4^(n1 k1); twiddle by W16^(n2 k1)
#pragma unroll
  for (int k1 = 1; k1 < 4; ++k1) {
#pragma unroll
    for (int n2 = 1; n2 < 4; ++n2) {
      const int m = n2 * k1;
      cf w = {c16[m], INV ? s16[m] : -s16[m]};
      x[4 * k1 + n2] = cmul(x[4 * k1 + n2], w);
    }
  }
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) dft4<INV>(x[4 * k1], x[4 * k1 + 1], x[4 * k1 + 2], x[4 * k1 + 3]);
  // x[4*k1 + k2] holds X[k1 + 4*k2]: transpose the 4x4 index grid (register renaming only)
  cf y[16];
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1)
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) y[k1 + 4 * k2] = x[4 * k1 + k2];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = y[i];
}

// Real-FFT split: given Z = DFT_M(z), z[m] = x[2m] + i x[2m+1], M = W/2, produce X[k], 0<=k<=M.
//   zk = Z[k mod M], zc = Z[(M - k) mod M], w = exp(-2 pi i k / W)
APS_HD cf r2c_split(cf zk, cf zc, cf w) {
  cf c = cconj(zc);
  cf e = cscale(zk + c, 0.5f);
  cf d = zk - c;                      // O = d / (2i) = (-i/2) d = (d.im/2, -d.re/2)
  cf o = {0.5f * d.im, -0.5f * d.re};
  return e + cmul(w, o);
}

// Inverse of the split: from X[k], X[M-k] (one-sided spectrum of a real signal, W = 2M) build
// Z[k] such that IDFT_M(Z)[m] = x[2m] + i x[2m+1] (up to the 1/M the caller folds in).
//   E = (X[k] + conj X[M-k]) / 2,  O = (X[k] - conj X[M-k]) / 2 * exp(+2 pi i k / W),  Z = E + i O
APS_HD cf c2r_merge(cf xk, cf xmk, cf w_pos) {
  cf c = cconj(xmk);
  cf e = cscale(xk + c, 0.5f);
  cf o = cmul(cscale(xk - c, 0.5f), w_pos);
  return {e.re - o.im, e.im + o.re};
}

}  // namespace aps
#endif  // APS_AMD_FFT_CORE_H_
