// Spectral (|X| -> power -> mel -> log -> cmvn) and spatial (cos/sin IPD) features from the
// bin-fastest spectrogram store: one wavefront per (utterance, frame) row, so every global access
// is a contiguous run of bins and the per-frame CMVN statistics are a single wave reduction.
//
// Replaces the nn.Sequential chain of aps/transform/asr.py:280-618 (MagnitudeTransform,
// TFTransposeTransform, PowerTransform, MelTransform, LogTransform, CmvnTransform per "band",
// AbsTransform) and aps/transform/enh.py:21-143 (RefChannel, Phase, Ipd) + the concat of
// EnhTransform.forward (enh.py:595-613).
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
  const int per_wave = F + (D0 > F ? D0 : F) + (a.num_pairs > 0 ? a.C * F : 0);
  float* s_mag = reinterpret_cast<float*>(smem) + (size_t)wv * per_wave;  // [F]
  float* s_val = s_mag + F;                                               // [max(D0, F)]
  float* s_pha = s_val + (D0 > F ? D0 : F);                               // [C][F]

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
  if (!ABS_MODE && a.num_pairs > 0) {
    if (active) {
      for (int c = 0; c < a.C; ++c) {
        const float* ch = base + (int64_t)c * a.stride_c;
        for (int f = ln; f < F; f += 64) {
          cf x = ld_cf(ch + 2 * f);
          s_pha[c * F + f] = atan2f(x.im, x.re);
        }
      }
    }
    __syncthreads();
    if (active) {
      for (int p = 0; p < a.num_pairs; ++p) {
        const float* pl = s_pha + a.pair_l[p] * F;
        const float* pr = s_pha + a.pair_r[p] * F;
        float* oc = orow + D0 + (int64_t)p * F;
        float* os = orow + D0 + (int64_t)(a.num_pairs + p) * F;
        for (int f = ln; f < F; f += 64) {
          const float d = pl[f] - pr[f];
          const float cd = cosf(d);
          bad |= (cd != cd);
          oc[f] = cd;
          if (a.ipd_sin) os[f] = sinf(d);
        }
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
                    ((!abs_mode && p->num_pairs > 0) ? (size_t)p->num_channels * F : 0);
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
  size_t lds = feat_lds_bytes(p, 0);
  if (lds > 150 * 1024) return APS_ERR_UNSUPPORTED;
  if (lds > 48 * 1024)
    hipFuncSetAttribute(reinterpret_cast<const void*>(&features_kernel<0>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  dim3 grid((unsigned)((a.num_rows + kRowsPerBlock - 1) / kRowsPerBlock));
  hipLaunchKernelGGL((features_kernel<0>), grid, dim3(256), lds,
                     static_cast<hipStream_t>(stream), a);
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
