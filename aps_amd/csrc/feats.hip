// Spectral (|X| -> power -> mel -> log -> cmvn) and spatial (cos/sin IPD) features from the
// bin-fastest spectrogram store: one wavefront per (utterance, frame) row, so every global access
// is a contiguous run of bins and the per-frame CMVN statistics are a single wave reduction.
//
// Replaces the nn.Sequential chain of aps/transform/asr.py:280-618 (MagnitudeTransform,
// TFTransposeTransform, PowerTransform, MelTransform, LogTransform, CmvnTransform per "band",
// AbsTransform) and aps/transform/enh.py:21-143 (RefChannel, Phase, Ipd) + the concat of
// EnhTransform.forward (enh.py:595-613).
#include <stdlib.h>

#include "common.h"

namespace aps {

struct FeatArgs {
  const float* src;  // spectrogram store or complex rows
  float* out;
  const int32_t* mel_start;
  const int32_t* mel_len;
  const int32_t* mel_off;
  const float* mel_w;
  const int32_t* pair_l;
  const int32_t* pair_r;
  int64_t num_rows;  // N * T
  int64_t T;
  int64_t stride_n, stride_c, stride_t;
  int32_t F, C, ref_channel, power, num_mels, apply_log, norm_mean, norm_var, num_pairs, ipd_sin;
  float log_eps, log_lower_bound, cmvn_eps, abs_eps;
  int32_t D;  // output row width
  int32_t* nan_count;  // optional: += 1 per wave that wrote a NaN (check_valid, asr.py:33-45)
};

constexpr int kRowsPerBlock = 4;

// MODE 0: spectrogram store (magnitude of the reference channel + IPD of channel pairs)
// MODE 1: rows are complex vectors, magnitude = |(re + abs_eps) + i im| (asr.py:330-332)
// MODE 2: rows are real vectors (stand-alone PowerTransform / MelTransform / LogTransform / Cmvn)
template <int MODE>
__global__ __launch_bounds__(256) void features_kernel(FeatArgs a) {
  constexpr bool ABS_MODE = MODE != 0;  // row addressed, no channel axis
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
  const int F = a.F;
  const int D0 = (a.ref_channel >= 0) ? (a.num_mels > 0 ? a.num_mels : F) : 0;
  const int per_wave = F + (D0 > F ? D0 : F) + (a.num_pairs > 0 ? 2 * a.C * F : 0) + 1;
  float* s_mag = reinterpret_cast<float*>(smem) + (size_t)wv * per_wave;  // [F]
  float* s_val = s_mag + F;                                               // [max(D0, F)]
  float* s_pha = s_val + (D0 > F ? D0 : F);                               // [C][F] float2
  s_pha += (reinterpret_cast<uintptr_t>(s_pha) & 4) ? 1 : 0;              // 8-byte align

  const int64_t row = (int64_t)blockIdx.x * kRowsPerBlock + wv;
  const bool active = row < a.num_rows;
  const int64_t n = active ? row / a.T : 0;
  const int64_t t = active ? row % a.T : 0;
  const float* base = ABS_MODE ? a.src + row * a.stride_t
                               : a.src + n * a.stride_n + t * a.stride_t;
  float* orow = a.out + row * (int64_t)a.D;
  bool bad = false;

  // ---- magnitude branch ----------------------------------------------------------------
  if (D0 > 0) {
    if (active) {
      const float* ref = ABS_MODE ? base : base + (int64_t)a.ref_channel * a.stride_c;
      for (int f = ln; f < F; f += 64) {
        float v;
        if (MODE == 2) {
          v = ref[f];
        } else {
          cf x = ld_cf(ref + 2 * f);
          if (MODE == 1) x.re += a.abs_eps;
          v = sqrtf(x.re * x.re + x.im * x.im);
        }
        if (a.power == 2) v = v * v;
        s_mag[f] = v;
      }
    }
    __syncthreads();
    float part = 0.f;
    if (active) {
      for (int d = ln; d < D0; d += 64) {
        float v;
        if (a.num_mels > 0) {
          const int st = a.mel_start[d], len = a.mel_len[d];
          const float* w = a.mel_w + a.mel_off[d];
          v = 0.f;
          for (int q = 0; q < len; ++q) v += w[q] * s_mag[st + q];
        } else {
          v = s_mag[d];
        }
        if (a.apply_log) {
          // clamp(min=eps) must propagate NaN like th.clamp does (fmaxf would swallow it)
          v = (a.log_lower_bound > 0.f) ? logf(a.log_lower_bound + v)
                                        : logf(v < a.log_eps ? a.log_eps : v);
        }
        s_val[d] = v;
        part += v;
      }
    }
    if (a.norm_mean || a.norm_var) {
      // CmvnTransform._cmvn_per_band: statistics over the LAST dim, i.e. this row (asr.py:576-585)
      const float mean = wave_sum(part) / (float)D0;
      float sq = 0.f;
      if (active) {
        for (int d = ln; d < D0; d += 64) {
          const float c = s_val[d] - mean;
          sq += c * c;
          if (a.norm_mean) s_val[d] = c;
        }
      }
      const float var = wave_sum(sq) / (float)D0;
      if (active) {
        for (int d = ln; d < D0; d += 64) {
          const float o = a.norm_var ? s_val[d] / sqrtf(var + a.cmvn_eps) : s_val[d];
          bad |= (o != o);
          orow[d] = o;
        }
      }
    } else if (active) {
      for (int d = ln; d < D0; d += 64) {
        bad |= (s_val[d] != s_val[d]);
        orow[d] = s_val[d];
      }
    }
  }

  // ---- spatial branch ------------------------------------------------------------------
  // cos(arg a - arg b) = Re(ua * conj(ub)), sin(..) = Im(ua * conj(ub)) with u = x / |x|
  // (u = (+-1, 0) for x = 0, matching atan2(0, +-0)); no atan2 / cos / sin evaluations.
  if (MODE == 0 && a.num_pairs > 0) {
    float2* s_unit = reinterpret_cast<float2*>(s_pha);  // [C][F]
    if (active) {
      for (int c = 0; c < a.C; ++c) {
        const float* ch = base + (int64_t)c * a.stride_c;
        for (int f = ln; f < F; f += 64) {
          const cf x = ld_cf(ch + 2 * f);
          s_unit[c * F + f] = unit_vector(x);
        }
      }
    }
    __syncthreads();
    if (active) {
      for (int p = 0; p < a.num_pairs; ++p) {
        const float2* ul = s_unit + a.pair_l[p] * F;
        const float2* ur = s_unit + a.pair_r[p] * F;
        float* oc = orow + D0 + (int64_t)p * F;
        float* os = orow + D0 + (int64_t)(a.num_pairs + p) * F;
        for (int f = ln; f < F; f += 64) {
          const float2 l = ul[f], r = ur[f];
          const float cd = l.x * r.x + l.y * r.y;
          bad |= (cd != cd);
          oc[f] = cd;
          if (a.ipd_sin) os[f] = l.y * r.x - l.x * r.y;
        }
      }
    }
  }
  if (a.nan_count != nullptr && __any(bad) && ln == 0) atomicAdd(a.nan_count, 1);
}

// ------------------------------------------------------------------------------------------
// Register-resident row kernel for the spectrogram-store case (MODE 0) with C <= 4 channels and
// rows of at most 64 * NITER values: one wavefront per (n, t) row, no LDS for the spatial branch,
// no workgroup barrier.
//  * all C * NITER 8-byte loads of a row are issued before anything is computed (nothing in
//    between orders memory), so a wavefront has the whole 8 KB row in flight;
//  * unit vectors x / |x| stay in registers; the run-time IPD pair indices select among them
//    with wave-uniform branches;
//  * the reference channel's spectral values stay in NITER registers, so CMVN is two wave
//    reductions and one store pass -- the row is never re-read.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int NITER>
__global__ __launch_bounds__(256) void features_rows_kernel(FeatArgs a) {
  constexpr int CMAX = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
  const int F = a.F;
  const bool has_mag = a.ref_channel >= 0;
  const int D0 = has_mag ? (a.num_mels > 0 ? a.num_mels : F) : 0;
  float* s_mag = reinterpret_cast<float*>(smem) + (size_t)wv * F;  // mel only: magnitudes [F]

  const int64_t row = (int64_t)blockIdx.x * kRowsPerBlock + wv;
  if (row >= a.num_rows) return;  // no workgroup barrier below
  const int64_t n = row / a.T, t = row % a.T;
  const float* base = a.src + n * a.stride_n + t * a.stride_t;
  float* orow = a.out + row * (int64_t)a.D;
  bool bad = false;

  // ---- every load of the row up front ----
  cf x[CMAX][NITER];
#pragma unroll
  for (int c = 0; c < CMAX; ++c) {
    const bool use = (c < a.C) && (a.num_pairs > 0 || c == a.ref_channel);
#pragma unroll
    for (int i = 0; i < NITER; ++i) {
      const int f = ln + 64 * i;
      x[c][i] = (use && f < F) ? ld_cf(base + (int64_t)c * a.stride_c + 2 * f) : cf{1.f, 0.f};
    }
  }

  // ---- spectral branch on the reference channel ----
  float val[NITER];
  if (has_mag) {
#pragma unroll
    for (int i = 0; i < NITER; ++i) {
      const int f = ln + 64 * i;
      cf xr = x[0][i];
#pragma unroll
      for (int c = 1; c < CMAX; ++c) xr = (a.ref_channel == c) ? x[c][i] : xr;
      float v = cabs_fast(xr);
      if (a.power == 2) v = v * v;
      if (a.num_mels > 0) {
        if (f < F) s_mag[f] = v;
      } else {
        if (a.apply_log) v = log_feature(v, a.log_eps, a.log_lower_bound);
        val[i] = (f < F) ? v : 0.f;
      }
    }
  }

  // ---- spatial branch: cos / sin IPD from unit vectors held in registers ----
  if (a.num_pairs > 0) {
    float2 u[CMAX][NITER];
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
#pragma unroll
      for (int i = 0; i < NITER; ++i) u[c][i] = unit_vector(x[c][i]);
    for (int p = 0; p < a.num_pairs; ++p) {
      // wave-uniform channel indices select among the register-resident unit vectors (a scalar
      // jump table over the (l, r) cases measured 33 % slower than the select chains)
      const int il = a.pair_l[p], ir = a.pair_r[p];
      float* oc = orow + D0 + (int64_t)p * F;
      float* os = orow + D0 + (int64_t)(a.num_pairs + p) * F;
      auto emit = [&](const float2 (&l)[NITER], const float2 (&r)[NITER]) {
#pragma unroll
        for (int i = 0; i < NITER; ++i) {
          const int f = ln + 64 * i;
          const float cd = l[i].x * r[i].x + l[i].y * r[i].y;
          bad |= (cd != cd);
          if (f < F) {
            oc[f] = cd;
            if (a.ipd_sin) os[f] = l[i].y * r[i].x - l[i].x * r[i].y;
          }
        }
      };
      {
        float2 l[NITER], r[NITER];
#pragma unroll
        for (int i = 0; i < NITER; ++i) {
          l[i] = u[0][i];
          r[i] = u[0][i];
#pragma unroll
          for (int c = 1; c < CMAX; ++c) {
            l[i] = (il == c) ? u[c][i] : l[i];
            r[i] = (ir == c) ? u[c][i] : r[i];
          }
        }
        emit(l, r);
      }
    }
  }

  if (has_mag) {
    if (a.num_mels > 0) {
      wave_fence();
#pragma unroll
      for (int i = 0; i < NITER; ++i) {
        const int d = ln + 64 * i;
        float v = 0.f;
        if (d < D0) {
          const int st = a.mel_start[d], len = a.mel_len[d];
          const float* w = a.mel_w + a.mel_off[d];
          for (int q = 0; q < len; ++q) v += w[q] * s_mag[st + q];
          if (a.apply_log) v = log_feature(v, a.log_eps, a.log_lower_bound);
        }
        val[i] = v;
      }
    }
    if (a.norm_mean || a.norm_var) {
      float part = 0.f;
#pragma unroll
      for (int i = 0; i < NITER; ++i) part += val[i];  // out-of-range entries are 0
      const float mean = wave_sum(part) / (float)D0;
      float sq = 0.f;
#pragma unroll
      for (int i = 0; i < NITER; ++i) {
        const float c = val[i] - mean;
        if (ln + 64 * i < D0) sq += c * c;
        if (a.norm_mean) val[i] = c;
      }
      const float var = wave_sum(sq) / (float)D0;
      if (a.norm_var) {
        const float isd = fast_rsq(var + a.cmvn_eps);
#pragma unroll
        for (int i = 0; i < NITER; ++i) val[i] = val[i] * isd;
      }
    }
#pragma unroll
    for (int i = 0; i < NITER; ++i) {
      const int d = ln + 64 * i;
      if (d < D0) {
        bad |= (val[i] != val[i]);
        orow[d] = val[i];
      }
    }
  }
  if (a.nan_count != nullptr && __any(bad) && ln == 0) atomicAdd(a.nan_count, 1);
}

__global__ __launch_bounds__(256) void tf_mask_kernel(const float* __restrict__ store, int64_t T,
                                                      int64_t F, int64_t stride_n, int64_t stride_t,
                                                      const float* __restrict__ mask, int64_t ms_n,
                                                      int64_t ms_t, int64_t ms_f, int cplx,
                                                      float* __restrict__ out, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * 256) {
    const int64_t f = i % F;
    const int64_t r = i / F;
    const int64_t t = r % T, n = r / T;
    cf x = ld_cf(store + n * stride_n + t * stride_t + 2 * f);
    const float* m = mask + n * ms_n + t * ms_t + f * ms_f;
    cf y;
    if (cplx) {
      y = cmul(x, cf{m[0], m[1]});
    } else {
      y = cscale(x, m[0]);
    }
    st_cf(out + 2 * i, y);
  }
}

// backward of tf_mask_kernel: out = X m (real m) or X M (complex M)
//   real:    g_m = g_re X_re + g_im X_im;            g_X = g m
//   complex: g_M = conj(X) g (as a pair of reals);   g_X = conj(M) g
__global__ __launch_bounds__(256) void tf_mask_backward_kernel(
    const float* __restrict__ store, int64_t T, int64_t F, int64_t stride_n, int64_t stride_t,
    const float* __restrict__ mask, int64_t ms_n, int64_t ms_t, int64_t ms_f, int cplx,
    const float* __restrict__ grad_out, float* __restrict__ grad_mask, int64_t gs_n, int64_t gs_t,
    int64_t gs_f, float* __restrict__ grad_store, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * 256) {
    const int64_t f = i % F;
    const int64_t r = i / F;
    const int64_t t = r % T, n = r / T;
    const cf x = ld_cf(store + n * stride_n + t * stride_t + 2 * f);
    const cf g = ld_cf(grad_out + 2 * i);
    const float* m = mask + n * ms_n + t * ms_t + f * ms_f;
    if (grad_mask != nullptr) {
      float* gm = grad_mask + n * gs_n + t * gs_t + f * gs_f;
      gm[0] = g.re * x.re + g.im * x.im;
      if (cplx) gm[1] = g.im * x.re - g.re * x.im;
    }
    if (grad_store != nullptr) {
      const cf gx = cplx ? cmul(g, cf{m[0], -m[1]}) : cscale(g, m[0]);
      st_cf(grad_store + 2 * i, gx);
    }
  }
}

}  // namespace aps

using namespace aps;

static int check_feat_params(const aps_feat_params* p) {
  if (!p || p->num_bins < 1) return 0;
  if (p->power != 1 && p->power != 2) return 0;
  if (p->num_mels < 0 || p->num_pairs < 0) return 0;
  return 1;
}

static size_t feat_lds_bytes(const aps_feat_params* p, int abs_mode) {
  const int F = p->num_bins;
  const int D0 = (abs_mode || p->ref_channel >= 0) ? (p->num_mels > 0 ? p->num_mels : F) : 0;
  size_t per_wave = (size_t)F + (D0 > F ? D0 : F) +
                    ((!abs_mode && p->num_pairs > 0) ? (size_t)2 * p->num_channels * F : 0) + 1;
  return per_wave * kRowsPerBlock * sizeof(float);
}

extern "C" int aps_enh_features(const float* store, int64_t N, int64_t T, int64_t stride_n,
                                int64_t stride_c, int64_t stride_t, const aps_feat_params* p,
                                const int32_t* mel_start, const int32_t* mel_len,
                                const int32_t* mel_off, const float* mel_w, const int32_t* pair_l,
                                const int32_t* pair_r, float* out, int32_t* nan_count,
                                void* stream) {
  APS_CHECK_ARG(store && out && N > 0 && T > 0 && check_feat_params(p));
  APS_CHECK_ARG(p->num_channels >= 1 && p->ref_channel < p->num_channels);
  APS_CHECK_ARG(p->ref_channel >= 0 || p->num_pairs > 0);
  if (p->num_mels > 0) APS_CHECK_ARG(mel_start && mel_len && mel_off && mel_w);
  if (p->num_pairs > 0) APS_CHECK_ARG(pair_l && pair_r && p->num_channels >= 2);
  const int F = p->num_bins;
  const int D0 = (p->ref_channel >= 0) ? (p->num_mels > 0 ? p->num_mels : F) : 0;
  FeatArgs a{};
  a.src = store; a.out = out;
  a.mel_start = mel_start; a.mel_len = mel_len; a.mel_off = mel_off; a.mel_w = mel_w;
  a.pair_l = pair_l; a.pair_r = pair_r;
  a.num_rows = N * T; a.T = T;
  a.stride_n = stride_n; a.stride_c = stride_c; a.stride_t = stride_t;
  a.F = F; a.C = p->num_channels; a.ref_channel = p->ref_channel; a.power = p->power;
  a.num_mels = p->num_mels; a.apply_log = p->apply_log; a.norm_mean = p->norm_mean;
  a.norm_var = p->norm_var; a.num_pairs = p->num_pairs; a.ipd_sin = p->ipd_sin;
  a.log_eps = p->log_eps; a.log_lower_bound = p->log_lower_bound; a.cmvn_eps = p->cmvn_eps;
  a.abs_eps = 0.f;
  a.nan_count = nan_count;
  a.D = D0 + p->num_pairs * (p->ipd_sin ? 2 : 1) * F;
  dim3 grid((unsigned)((a.num_rows + kRowsPerBlock - 1) / kRowsPerBlock));
  hipStream_t st = static_cast<hipStream_t>(stream);
  // register-resident fast path when a row (F bins and D0 outputs) fits 64 * NITER values
  const int need = (F > D0 ? F : D0);
  const int niter = (need + 63) / 64;
  const size_t fast_lds = (size_t)(p->num_mels > 0 ? F : 0) * kRowsPerBlock * sizeof(float);
  if (niter <= 9 && p->num_channels <= 4) {
    if (niter <= 3)
      hipLaunchKernelGGL((features_rows_kernel<3>), grid, dim3(256), fast_lds, st, a);
    else if (niter <= 5)
      hipLaunchKernelGGL((features_rows_kernel<5>), grid, dim3(256), fast_lds, st, a);
    else
      hipLaunchKernelGGL((features_rows_kernel<9>), grid, dim3(256), fast_lds, st, a);
    return aps_launch_status();
  }
  size_t lds = feat_lds_bytes(p, 0);
  if (lds > 150 * 1024) return APS_ERR_UNSUPPORTED;
  if (lds > 48 * 1024)
    hipFuncSetAttribute(reinterpret_cast<const void*>(&features_kernel<0>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((features_kernel<0>), grid, dim3(256), lds, st, a);
  return aps_launch_status();
}

static int rows_features(int mode, const float* y, int64_t num_rows, int64_t stride_row,
                         float abs_eps, const aps_feat_params* p, const int32_t* mel_start,
                         const int32_t* mel_len, const int32_t* mel_off, const float* mel_w,
                         float* out, int32_t* nan_count, void* stream) {
  APS_CHECK_ARG(y && out && num_rows > 0 && check_feat_params(p));
  if (p->num_mels > 0) APS_CHECK_ARG(mel_start && mel_len && mel_off && mel_w);
  const int F = p->num_bins;
  FeatArgs a{};
  a.src = y; a.out = out;
  a.mel_start = mel_start; a.mel_len = mel_len; a.mel_off = mel_off; a.mel_w = mel_w;
  a.num_rows = num_rows; a.T = 1;
  a.stride_t = stride_row;
  a.F = F; a.C = 1; a.ref_channel = 0; a.power = p->power;
  a.num_mels = p->num_mels; a.apply_log = p->apply_log; a.norm_mean = p->norm_mean;
  a.norm_var = p->norm_var; a.num_pairs = 0; a.ipd_sin = 0;
  a.log_eps = p->log_eps; a.log_lower_bound = p->log_lower_bound; a.cmvn_eps = p->cmvn_eps;
  a.abs_eps = abs_eps;
  a.nan_count = nan_count;
  a.D = p->num_mels > 0 ? p->num_mels : F;
  size_t lds = feat_lds_bytes(p, 1);
  if (lds > 150 * 1024) return APS_ERR_UNSUPPORTED;
  dim3 grid((unsigned)((num_rows + kRowsPerBlock - 1) / kRowsPerBlock));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (mode == 1) {
    if (lds > 48 * 1024)
      hipFuncSetAttribute(reinterpret_cast<const void*>(&features_kernel<1>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((features_kernel<1>), grid, dim3(256), lds, st, a);
  } else {
    if (lds > 48 * 1024)
      hipFuncSetAttribute(reinterpret_cast<const void*>(&features_kernel<2>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((features_kernel<2>), grid, dim3(256), lds, st, a);
  }
  return aps_launch_status();
}

// ------------------------------------------------------------------------------------------
// The enhancement chain's layers called on their own (aps/transform/enh.py:52-143): inside
// EnhTransform.forward the phase never exists (features_rows_kernel<5> forms cos / sin of the phase
// differences from unit vectors), but the reference lets a caller run PhaseTransform / IpdTransform
// as modules -- two element-wise kernels.
// ------------------------------------------------------------------------------------------
// x [outer, 2, inner] (the (re, im) axis anywhere) -> [outer, inner]: op 0 atan2(im, re)
// (PhaseTransform), op 1 sqrt(re^2 + im^2 + eps) (MagnitudeTransform with any dim / eps)
__global__ __launch_bounds__(256) void reim_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                   int64_t outer, int64_t inner, int op, float eps) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= outer * inner) return;
  const int64_t o = i / inner, r = i - o * inner;
  const float re = x[(o * 2) * inner + r], im = x[(o * 2 + 1) * inner + r];
  out[i] = op == 0 ? atan2f(im, re) : sqrtf(re * re + im * im + eps);
}

// phase p [N, C, T, F] -> ipd [N, T, M, F]: cos(p_l - p_r) for the P pairs, then (sin: M = 2 P) the sines
__global__ __launch_bounds__(256) void ipd_phase_kernel(const float* __restrict__ p,
                                                        const int32_t* __restrict__ il,
                                                        const int32_t* __restrict__ ir, int64_t N,
                                                        int64_t C, int64_t T, int64_t F, int P,
                                                        int with_sin, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int M = with_sin ? 2 * P : P;
  if (i >= N * T * M * F) return;
  const int64_t f = i % F, m = (i / F) % M, t = (i / (F * M)) % T, n = i / (F * M * T);
  const int pair = (int)(m % P);
  const float d = p[((n * C + il[pair]) * T + t) * F + f] - p[((n * C + ir[pair]) * T + t) * F + f];
  out[i] = m < P ? cosf(d) : sinf(d);
}

extern "C" int aps_reim_axis(const float* x, float* out, int64_t outer, int64_t inner, int32_t op,
                             float eps, void* stream) {
  APS_CHECK_ARG(x && out && outer > 0 && inner > 0 && (op == 0 || op == 1));
  const int64_t n = outer * inner;
  hipLaunchKernelGGL(reim_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, out, outer, inner, (int)op, eps);
  return aps_launch_status();
}

extern "C" int aps_ipd_from_phase(const float* phase, const int32_t* pair_l, const int32_t* pair_r,
                                  int64_t N, int64_t C, int64_t T, int64_t F, int32_t num_pairs,
                                  int32_t with_sin, float* out, void* stream) {
  APS_CHECK_ARG(phase && pair_l && pair_r && out && N > 0 && C > 1 && T > 0 && F > 0 && num_pairs > 0);
  const int64_t n = N * T * (with_sin ? 2 : 1) * num_pairs * F;
  hipLaunchKernelGGL(ipd_phase_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), phase, pair_l, pair_r, N, C, T, F, (int)num_pairs,
                     (int)with_sin, out);
  return aps_launch_status();
}

extern "C" int aps_abs_features(const float* y, int64_t num_rows, int64_t stride_row, float abs_eps,
                                const aps_feat_params* p, const int32_t* mel_start,
                                const int32_t* mel_len, const int32_t* mel_off, const float* mel_w,
                                float* out, int32_t* nan_count, void* stream) {
  return rows_features(1, y, num_rows, stride_row, abs_eps, p, mel_start, mel_len, mel_off, mel_w,
                       out, nan_count, stream);
}

extern "C" int aps_row_features(const float* x, int64_t num_rows, int64_t stride_row,
                                const aps_feat_params* p, const int32_t* mel_start,
                                const int32_t* mel_len, const int32_t* mel_off, const float* mel_w,
                                float* out, int32_t* nan_count, void* stream) {
  return rows_features(2, x, num_rows, stride_row, 0.f, p, mel_start, mel_len, mel_off, mel_w, out,
                       nan_count, stream);
}

extern "C" int aps_tf_mask_backward(const float* store, int64_t N, int64_t T, int64_t F,
                                    int64_t stride_n, int64_t stride_t, const float* mask,
                                    int64_t mask_stride_n, int64_t mask_stride_t,
                                    int64_t mask_stride_f, int32_t mask_complex,
                                    const float* grad_out, float* grad_mask, int64_t gm_stride_n,
                                    int64_t gm_stride_t, int64_t gm_stride_f, float* grad_store,
                                    void* stream) {
  APS_CHECK_ARG(store && mask && grad_out && (grad_mask || grad_store) && N > 0 && T > 0 && F > 0);
  const int64_t total = N * T * F;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(tf_mask_backward_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     static_cast<hipStream_t>(stream), store, T, F, stride_n, stride_t, mask,
                     mask_stride_n, mask_stride_t, mask_stride_f, (int)mask_complex, grad_out,
                     grad_mask, gm_stride_n, gm_stride_t, gm_stride_f, grad_store, total);
  return aps_launch_status();
}

extern "C" int aps_tf_mask(const float* store, int64_t N, int64_t T, int64_t F, int64_t stride_n,
                           int64_t stride_t, const float* mask, int64_t mask_stride_n,
                           int64_t mask_stride_t, int64_t mask_stride_f, int32_t mask_complex,
                           float* out, void* stream) {
  APS_CHECK_ARG(store && mask && out && N > 0 && T > 0 && F > 0);
  const int64_t total = N * T * F;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(tf_mask_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     static_cast<hipStream_t>(stream), store, T, F, stride_n, stride_t, mask,
                     mask_stride_n, mask_stride_t, mask_stride_f, (int)mask_complex, out, total);
  return aps_launch_status();
}
