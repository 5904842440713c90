// Shared by conv.hip (fp32 MFMA / direct kernels) and gemm_split.hip (the bf16-split form):
// the convolution's argument block and its coordinate helpers.
#pragma once
#include <stdint.h>

#include "common.h"

namespace aps {

struct ConvArgs {
  const float* x;
  const float* w;
  const float* scale;     // [Co] per-channel multiplier (BatchNorm affine) or null
  const float* shift;     // [Co] per-channel offset (bias / BatchNorm) or null
  const float* residual;  // [M, Co] added after the activation or null
  float* y;
  int32_t N, H, W, Ci, Ho, Wo, Co;
  int32_t KH, KW, sh, sw, ph, pw;
  int32_t transposed;  // 1: y[ho, wo] gathers x[(ho + ph - kh) / sh, (wo + pw - kw) / sw]
  int32_t act;         // 0 none, 1 relu, 5 leaky relu (slope)
  float slope;
  int64_t M;
  int32_t direct_pix;  // conv_direct_kernel: output pixels per workgroup (multiple of 4)
  int32_t by_class;    // conv_mfma_kernel: rows ordered by stride residue class (transposed, stride > 1)
};

// rows of residue class (qh, qw): ho = qh + sh jh, wo = qw + sw jw
__host__ __device__ __forceinline__ int64_t class_rows(int N, int Ho, int Wo, int sh, int sw, int qh,
                                                       int qw, int& Hc, int& Wc) {
  Hc = qh < Ho ? (Ho - qh + sh - 1) / sh : 0;
  Wc = qw < Wo ? (Wo - qw + sw - 1) / sw : 0;
  return (int64_t)N * Hc * Wc;
}

__device__ __forceinline__ float conv_act(float v, int act, float slope) {
  if (act == 1) v = fmaxf(v, 0.f);
  if (act == 5) v = v > 0.f ? v : v * slope;
  return v;
}

// input coordinate of output coordinate o for tap k; returns false when the tap reads padding
__device__ __forceinline__ bool tap_coord(int o, int k, int stride, int pad, int size,
                                          int transposed, int& i) {
  if (!transposed) {
    i = o * stride + k - pad;
    return (unsigned)i < (unsigned)size;
  }
  const int t = o + pad - k;
  i = t / stride;
  return t >= 0 && t == i * stride && i < size;
}

}  // namespace aps
