// Mask-based MVDR on the bin-fastest spectrogram store (gfx950).
//
//   covariance : lanes own bins, frames are strided over the 8 half-waves of a workgroup, so a
//                (n, f) covariance is 2*C*C register accumulators per lane and the only cross-lane
//                traffic is one LDS reduction at the end.  Mask padding / max-normalisation /
//                transposition of _process_mask are folded into the same pass.
//   attention  : |off-diagonal mean of Rs| -> Linear -> tanh -> Linear, split over chunks of 64
//                hidden units so the tiny GEMV still covers the chip; softmax in a finalise step.
//   weight     : one lane per (n, f): in-register complex Gaussian elimination (partial pivoting)
//                of (Rn + eps I) Y = Rs, trace normalisation, projection on u.
//   beamform   : y = sum_c conj(w_c) x_c, lanes along bins, weights held in registers.
//
// Replaces aps/asr/filter/mvdr.py:19-174 and the ComplexTensor algebra it uses
// (aps/cplx.py:212-278).
#include "common.h"

namespace aps {

// ------------------------------------------------------------------------------------------
// covariance
// ------------------------------------------------------------------------------------------
struct CovArgs {
  const float* store;
  const float* mask_s;
  const float* mask_n;
  const int64_t* x_len;
  float* cov_s;
  float* cov_n;
  float* pmask_s;
  float* pmask_n;
  int64_t T, F;
  int64_t stride_n, stride_c, stride_t;
  int32_t mask_norm;
};

constexpr int kCovBins = 32;    // bins per workgroup
constexpr int kCovPhases = 8;   // frame phases per workgroup (256 threads)

template <int C>
__global__ __launch_bounds__(256) void covariance_kernel(CovArgs a) {
  constexpr int NV = 4 * C * C + 2;  // [speech | noise] x (C x C x 2, upper triangle used) + 2 mask sums
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_red = reinterpret_cast<float*>(smem);  // [4 waves][NV][32]
  __shared__ float s_max[kCovPhases][2][kCovBins];

  const int tid = threadIdx.x;
  const int fl = tid & 31;
  const int tp = tid >> 5;
  const int64_t n = blockIdx.y;
  const int64_t f = (int64_t)blockIdx.x * kCovBins + fl;
  const bool valid = f < a.F;
  const int64_t T = a.T, F = a.F;
  int64_t len = T;
  if (a.x_len) {
    len = a.x_len[n];
    if (len > T) len = T;
    if (len < 0) len = 0;
  }
  const float* ms_p = a.mask_s + n * T * F + f;
  const float* mn_p = a.mask_n ? a.mask_n + n * T * F + f : nullptr;

  // ---- _process_mask: max_t |mask| after zeroing padded frames (mvdr.py:109-114) ----
  float div_s = 1.f, div_n = 1.f;
  if (a.mask_norm) {
    float mx_s = 0.f, mx_n = 0.f;
    if (valid) {
      for (int64_t t = tp; t < len; t += kCovPhases) {
        mx_s = fmaxf(mx_s, fabsf(ms_p[t * F]));
        if (mn_p) mx_n = fmaxf(mx_n, fabsf(mn_p[t * F]));
      }
    }
    s_max[tp][0][fl] = mx_s;
    s_max[tp][1][fl] = mx_n;
    __syncthreads();
    mx_s = 0.f;
    mx_n = 0.f;
#pragma unroll
    for (int q = 0; q < kCovPhases; ++q) {
      mx_s = fmaxf(mx_s, s_max[q][0][fl]);
      mx_n = fmaxf(mx_n, s_max[q][1][fl]);
    }
    div_s = mx_s + APS_EPSILON;
    div_n = mx_n + APS_EPSILON;
  }

  // ---- accumulate sum_t m x x^H (upper triangle) for both masks ----
  float acc_s[C][C][2], acc_n[C][C][2];
#pragma unroll
  for (int i = 0; i < C; ++i)
#pragma unroll
    for (int j = 0; j < C; ++j) {
      acc_s[i][j][0] = acc_s[i][j][1] = 0.f;
      acc_n[i][j][0] = acc_n[i][j][1] = 0.f;
    }
  float sum_s = 0.f, sum_n = 0.f;
  if (valid) {
    const float* xb = a.store + n * a.stride_n + 2 * f;
    for (int64_t t = tp; t < T; t += kCovPhases) {
      float ms = (t < len) ? ms_p[t * F] : 0.f;
      if (a.mask_norm) ms = ms / div_s;
      float mn;
      if (mn_p) {
        mn = (t < len) ? mn_p[t * F] : 0.f;
        if (a.mask_norm) mn = mn / div_n;
      } else {
        mn = 1.0f - ms;  // mvdr.py:136
      }
      if (a.pmask_s) a.pmask_s[(n * F + f) * T + t] = ms;
      if (a.pmask_n) a.pmask_n[(n * F + f) * T + t] = mn;
      cf x[C];
#pragma unroll
      for (int c = 0; c < C; ++c) x[c] = ld_cf(xb + c * a.stride_c + t * a.stride_t);
      sum_s += ms;
      sum_n += mn;
#pragma unroll
      for (int i = 0; i < C; ++i) {
#pragma unroll
        for (int j = i; j < C; ++j) {
          const float pr = x[i].re * x[j].re + x[i].im * x[j].im;
          acc_s[i][j][0] += ms * pr;
          acc_n[i][j][0] += mn * pr;
          if (i != j) {  // the diagonal of x x^H is real: keep it exactly so
            const float pi = x[i].im * x[j].re - x[i].re * x[j].im;
            acc_s[i][j][1] += ms * pi;
            acc_n[i][j][1] += mn * pi;
          }
        }
      }
    }
  }

  // ---- reduce the 8 frame phases: pairs inside a wave by shuffle, waves through LDS ----
  const int wv = tid >> 6;
  auto put = [&](int v, float val) {
    val += __shfl_xor(val, 32, 64);
    if ((tid & 32) == 0) s_red[(wv * NV + v) * kCovBins + fl] = val;
  };
#pragma unroll
  for (int i = 0; i < C; ++i)
#pragma unroll
    for (int j = i; j < C; ++j) {
      put((i * C + j) * 2 + 0, acc_s[i][j][0]);
      put((i * C + j) * 2 + 1, acc_s[i][j][1]);
      put(C * C * 2 + (i * C + j) * 2 + 0, acc_n[i][j][0]);
      put(C * C * 2 + (i * C + j) * 2 + 1, acc_n[i][j][1]);
    }
  put(NV - 2, sum_s);
  put(NV - 1, sum_n);
  __syncthreads();
  if (!valid) return;
  auto total = [&](int v) {
    return s_red[(0 * NV + v) * kCovBins + fl] + s_red[(1 * NV + v) * kCovBins + fl] +
           s_red[(2 * NV + v) * kCovBins + fl] + s_red[(3 * NV + v) * kCovBins + fl];
  };
  const float den_s = fmaxf(total(NV - 2), APS_EPSILON);  // clamp(min=EPSILON), mvdr.py:59
  const float den_n = fmaxf(total(NV - 1), APS_EPSILON);
  // the 8 phases of a bin share the C*(C+1)/2 upper entries round-robin
  int e = 0;
#pragma unroll
  for (int i = 0; i < C; ++i)
#pragma unroll
    for (int j = i; j < C; ++j, ++e) {
      if ((e & (kCovPhases - 1)) != tp) continue;
      const int v = (i * C + j) * 2;
      const float sr = total(v) / den_s, si = total(v + 1) / den_s;
      const float nr = total(C * C * 2 + v) / den_n, ni = total(C * C * 2 + v + 1) / den_n;
      float* os = a.cov_s + (n * F + f) * (C * C * 2);
      float* on = a.cov_n + (n * F + f) * (C * C * 2);
      st_cf(os + (i * C + j) * 2, {sr, si});
      st_cf(on + (i * C + j) * 2, {nr, ni});
      if (i != j) {
        st_cf(os + (j * C + i) * 2, {sr, -si});
        st_cf(on + (j * C + i) * 2, {nr, -ni});
      }
    }
}

// ------------------------------------------------------------------------------------------
// channel attention
// ------------------------------------------------------------------------------------------
constexpr int kAttChunk = 64;  // hidden units per workgroup

template <int C>
__global__ __launch_bounds__(256) void attention_partial_kernel(
    const float* __restrict__ cov_s, int64_t F, int64_t A, const float* __restrict__ proj_w,
    const float* __restrict__ proj_b, const float* __restrict__ gvec_w,
    float* __restrict__ scratch) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_v = reinterpret_cast<float*>(smem);  // [C][F]
  __shared__ float s_part[4][C];
  const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
  const int64_t n = blockIdx.y;
  const int nchunk = gridDim.x;
  // |sum_{j != c} Rs[f, c, j]| / (C - 1)     (mvdr.py:165-170)
  const float* rs = cov_s + n * F * (C * C * 2);
  for (int64_t idx = tid; idx < F * C; idx += 256) {
    const int64_t f = idx / C;
    const int c = (int)(idx % C);
    const float* r = rs + idx * (C * 2);
    float re = 0.f, im = 0.f;
#pragma unroll
    for (int j = 0; j < C; ++j) {
      if (j != c) {
        re += r[2 * j];
        im += r[2 * j + 1];
      }
    }
    re = re / (float)(C - 1);
    im = im / (float)(C - 1);
    s_v[c * F + f] = sqrtf(re * re + im * im);
  }
  __syncthreads();
  float score[C];
#pragma unroll
  for (int c = 0; c < C; ++c) score[c] = 0.f;
  const int64_t a0 = (int64_t)blockIdx.x * kAttChunk + wv * (kAttChunk / 4);
  for (int q = 0; q < kAttChunk / 4; ++q) {
    const int64_t aa = a0 + q;
    if (aa >= A) break;
    const float* pw = proj_w + aa * F;
    float dot[C];
#pragma unroll
    for (int c = 0; c < C; ++c) dot[c] = 0.f;
    for (int64_t f = ln; f < F; f += 64) {
      const float w = pw[f];
#pragma unroll
      for (int c = 0; c < C; ++c) dot[c] += w * s_v[c * F + f];
    }
    const float b = proj_b[aa], g = gvec_w[aa];
#pragma unroll
    for (int c = 0; c < C; ++c) score[c] += g * tanhf(wave_sum(dot[c]) + b);
  }
  if (ln == 0) {
#pragma unroll
    for (int c = 0; c < C; ++c) s_part[wv][c] = score[c];
  }
  __syncthreads();
  if (tid < C) {
    scratch[(n * C + tid) * nchunk + blockIdx.x] =
        s_part[0][tid] + s_part[1][tid] + s_part[2][tid] + s_part[3][tid];
  }
}

__global__ void attention_finalize_kernel(const float* __restrict__ scratch, int64_t N, int C,
                                          int nchunk, const float* __restrict__ gvec_b,
                                          float* __restrict__ u) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float s[8];
  float mx = -INFINITY;
  for (int c = 0; c < C; ++c) {
    float v = gvec_b[0];
    for (int q = 0; q < nchunk; ++q) v += scratch[(n * C + c) * nchunk + q];
    s[c] = v;
    mx = fmaxf(mx, v);
  }
  float den = 0.f;
  for (int c = 0; c < C; ++c) {
    s[c] = expf(s[c] - mx);
    den += s[c];
  }
  for (int c = 0; c < C; ++c) u[n * C + c] = s[c] / den;
}

// ------------------------------------------------------------------------------------------
// MVDR weight
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ cf crecip(cf a) {
  const float s = a.re * a.re + a.im * a.im;
  return {a.re / s, -a.im / s};
}

template <int C>
__global__ __launch_bounds__(64) void weight_kernel(const float* __restrict__ cov_s,
                                                    const float* __restrict__ cov_n,
                                                    const float* __restrict__ u, int64_t NF,
                                                    int64_t F, float eps,
                                                    float* __restrict__ weight) {
  const int64_t idx = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (idx >= NF) return;
  const int64_t n = idx / F;
  cf A[C][C], B[C][C];
  const float* pn = cov_n + idx * (C * C * 2);
  const float* ps = cov_s + idx * (C * C * 2);
#pragma unroll
  for (int i = 0; i < C; ++i)
#pragma unroll
    for (int j = 0; j < C; ++j) {
      A[i][j] = ld_cf(pn + (i * C + j) * 2);
      B[i][j] = ld_cf(ps + (i * C + j) * 2);
    }
#pragma unroll
  for (int i = 0; i < C; ++i) A[i][i].re += eps;  // Rn + eps I   (mvdr.py:89-90)

  // forward elimination with partial pivoting on [A | B]
#pragma unroll
  for (int k = 0; k < C; ++k) {
    float best = A[k][k].re * A[k][k].re + A[k][k].im * A[k][k].im;
#pragma unroll
    for (int r = k + 1; r < C; ++r) {
      const float mag = A[r][k].re * A[r][k].re + A[r][k].im * A[r][k].im;
      const bool sw = mag > best;
      best = sw ? mag : best;
#pragma unroll
      for (int j = 0; j < C; ++j) {
        if (j >= k) {
          const cf t0 = A[k][j], t1 = A[r][j];
          A[k][j] = sw ? t1 : t0;
          A[r][j] = sw ? t0 : t1;
        }
        const cf b0 = B[k][j], b1 = B[r][j];
        B[k][j] = sw ? b1 : b0;
        B[r][j] = sw ? b0 : b1;
      }
    }
    const cf inv = crecip(A[k][k]);
#pragma unroll
    for (int i = k + 1; i < C; ++i) {
      const cf fct = cmul(A[i][k], inv);
#pragma unroll
      for (int j = k + 1; j < C; ++j) A[i][j] = A[i][j] - cmul(fct, A[k][j]);
#pragma unroll
      for (int j = 0; j < C; ++j) B[i][j] = B[i][j] - cmul(fct, B[k][j]);
    }
    A[k][k] = inv;  // keep the reciprocal pivot for the back substitution
  }
  // back substitution: Y overwrites B
#pragma unroll
  for (int k = C - 1; k >= 0; --k) {
#pragma unroll
    for (int j = 0; j < C; ++j) {
      cf s = B[k][j];
#pragma unroll
      for (int m = k + 1; m < C; ++m) s = s - cmul(A[k][m], B[m][j]);
      B[k][j] = cmul(s, A[k][k]);
    }
  }
  // trace(Y) + eps, Y u, complex division  (mvdr.py:96-100, cplx.py:221-229)
  cf tr = {eps, 0.f};
#pragma unroll
  for (int k = 0; k < C; ++k) tr = tr + B[k][k];
  const float scale = tr.re * tr.re + tr.im * tr.im;
  float uu[C];
#pragma unroll
  for (int c = 0; c < C; ++c) uu[c] = u[n * C + c];
#pragma unroll
  for (int i = 0; i < C; ++i) {
    cf v = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < C; ++j) v = v + cscale(B[i][j], uu[j]);
    cf w = {(v.re * tr.re + v.im * tr.im) / scale, (v.im * tr.re - v.re * tr.im) / scale};
    st_cf(weight + (idx * C + i) * 2, w);
  }
}

// ------------------------------------------------------------------------------------------
// beamform
// ------------------------------------------------------------------------------------------
constexpr int kBfFramesPerWave = 4;

template <int C>
__global__ __launch_bounds__(256) void beamform_kernel(const float* __restrict__ store,
                                                       const float* __restrict__ weight, int64_t T,
                                                       int64_t F, int64_t stride_n,
                                                       int64_t stride_c, int64_t stride_t,
                                                       float* __restrict__ y) {
  const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
  const int64_t n = blockIdx.y;
  const int64_t t0 = ((int64_t)blockIdx.x * 4 + wv) * kBfFramesPerWave;
  if (t0 >= T) return;
  const int64_t t1 = (t0 + kBfFramesPerWave < T) ? t0 + kBfFramesPerWave : T;
  for (int64_t f = ln; f < F; f += 64) {
    cf w[C];
#pragma unroll
    for (int c = 0; c < C; ++c) w[c] = ld_cf(weight + ((n * F + f) * C + c) * 2);
    for (int64_t t = t0; t < t1; ++t) {
      const float* xb = store + n * stride_n + t * stride_t + 2 * f;
      cf acc = {0.f, 0.f};
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const cf x = ld_cf(xb + c * stride_c);
        acc.re += w[c].re * x.re + w[c].im * x.im;  // conj(w) * x
        acc.im += w[c].re * x.im - w[c].im * x.re;
      }
      st_cf(y + ((n * T + t) * F + f) * 2, acc);
    }
  }
}

}  // namespace aps

using namespace aps;

#define APS_DISPATCH_C(C, ...)      \
  switch (C) {                       \
    case 2: { constexpr int kC = 2; __VA_ARGS__; } break; \
    case 3: { constexpr int kC = 3; __VA_ARGS__; } break; \
    case 4: { constexpr int kC = 4; __VA_ARGS__; } break; \
    case 5: { constexpr int kC = 5; __VA_ARGS__; } break; \
    case 6: { constexpr int kC = 6; __VA_ARGS__; } break; \
    case 7: { constexpr int kC = 7; __VA_ARGS__; } break; \
    case 8: { constexpr int kC = 8; __VA_ARGS__; } break; \
    default: return APS_ERR_UNSUPPORTED; \
  }

extern "C" int aps_mvdr_covariance(const float* store, int64_t N, int64_t C, int64_t T, int64_t F,
                                   int64_t stride_n, int64_t stride_c, int64_t stride_t,
                                   const float* mask_s, const float* mask_n, const int64_t* x_len,
                                   int32_t mask_norm, float* cov_s, float* cov_n, float* pmask_s,
                                   float* pmask_n, void* stream) {
  APS_CHECK_ARG(store && mask_s && cov_s && cov_n);
  APS_CHECK_ARG(N > 0 && N <= 65535 && T > 0 && F > 0);
  hipStream_t st = static_cast<hipStream_t>(stream);
  CovArgs a{store, mask_s, mask_n, x_len, cov_s, cov_n, pmask_s, pmask_n, T, F,
            stride_n, stride_c, stride_t, mask_norm};
  dim3 grid((unsigned)((F + kCovBins - 1) / kCovBins), (unsigned)N);
  APS_DISPATCH_C(C, {
    size_t lds = (size_t)4 * (4 * kC * kC + 2) * kCovBins * sizeof(float);
    if (lds > 48 * 1024)
      hipFuncSetAttribute(reinterpret_cast<const void*>(&covariance_kernel<kC>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((covariance_kernel<kC>), grid, dim3(256), lds, st, a);
  });
  return aps_launch_status();
}

extern "C" int aps_mvdr_channel_attention(const float* cov_s, int64_t N, int64_t C, int64_t F,
                                          int64_t A, const float* proj_w, const float* proj_b,
                                          const float* gvec_w, const float* gvec_b, float* scratch,
                                          float* u_out, void* stream) {
  APS_CHECK_ARG(cov_s && proj_w && proj_b && gvec_w && gvec_b && scratch && u_out);
  APS_CHECK_ARG(N > 0 && N <= 65535 && F > 0 && A > 0);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nchunk = (int)((A + kAttChunk - 1) / kAttChunk);
  dim3 grid((unsigned)nchunk, (unsigned)N);
  APS_DISPATCH_C(C, {
    size_t lds = (size_t)kC * F * sizeof(float);
    if (lds > 150 * 1024) return APS_ERR_UNSUPPORTED;
    if (lds > 48 * 1024)
      hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_partial_kernel<kC>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((attention_partial_kernel<kC>), grid, dim3(256), lds, st, cov_s, F, A,
                       proj_w, proj_b, gvec_w, scratch);
  });
  hipLaunchKernelGGL(attention_finalize_kernel, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, st,
                     scratch, N, (int)C, nchunk, gvec_b, u_out);
  return aps_launch_status();
}

extern "C" int aps_mvdr_weight(const float* cov_s, const float* cov_n, const float* u, int64_t N,
                               int64_t C, int64_t F, float eps, float* weight_out, void* stream) {
  APS_CHECK_ARG(cov_s && cov_n && u && weight_out && N > 0 && F > 0);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t NF = N * F;
  dim3 grid((unsigned)((NF + 63) / 64));
  APS_DISPATCH_C(C, {
    hipLaunchKernelGGL((weight_kernel<kC>), grid, dim3(64), 0, st, cov_s, cov_n, u, NF, F, eps,
                       weight_out);
  });
  return aps_launch_status();
}

extern "C" int aps_mvdr_beamform(const float* store, const float* weight, int64_t N, int64_t C,
                                 int64_t T, int64_t F, int64_t stride_n, int64_t stride_c,
                                 int64_t stride_t, float* y_out, void* stream) {
  APS_CHECK_ARG(store && weight && y_out && N > 0 && N <= 65535 && T > 0 && F > 0);
  hipStream_t st = static_cast<hipStream_t>(stream);
  dim3 grid((unsigned)((T + 4 * kBfFramesPerWave - 1) / (4 * kBfFramesPerWave)), (unsigned)N);
  APS_DISPATCH_C(C, {
    hipLaunchKernelGGL((beamform_kernel<kC>), grid, dim3(256), 0, st, store, weight, T, F,
                       stride_n, stride_c, stride_t, y_out);
  });
  return aps_launch_status();
}
