// Post-spectral feature layers of AsrTransform that work across frames / whole utterances:
//   SpliceTransform   (aps/transform/asr.py:687-728, splice_feature utils.py:193-224)
//   DeltaTransform    (asr.py:731-782)
//   CmvnTransform with all-band (utterance) statistics or global statistics (asr.py:576-618)
// All are HBM-bound gathers / two-pass reductions over [rows, F] feature matrices; rows of one
// utterance are contiguous (T frames x F features, frame pitch and utterance pitch given).
#include "common.h"

namespace aps {

// out[u, to, c * F + f] = in[u, clamp(to * sub + c - lctx, 0, T - 1), f],  c = 0 .. lctx + rctx
__global__ __launch_bounds__(256) void splice_kernel(const float* __restrict__ in,
                                                     float* __restrict__ out, int64_t total, int T,
                                                     int To, int F, int lctx, int D, int sub,
                                                     int64_t in_utt, int64_t in_row) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * 256) {
    const int f = (int)(i % F);
    const int c = (int)((i / F) % D);
    const int to = (int)((i / ((int64_t)F * D)) % To);
    const int64_t u = i / ((int64_t)F * D * To);
    const int t = min(max(to * sub + c - lctx, 0), T - 1);
    out[i] = in[u * in_utt + (int64_t)t * in_row + f];
  }
}

// out[u, t, f] = sum_c scale[c] * in[u, clamp(t + c - ctx, 0, T - 1), f],  c = 0 .. 2 ctx
// (one delta order; in / out may be different column blocks or channel planes of one buffer)
__global__ __launch_bounds__(256) void delta_kernel(const float* __restrict__ in,
                                                    float* __restrict__ out,
                                                    const float* __restrict__ scale, int64_t total,
                                                    int T, int F, int ctx, int64_t in_utt,
                                                    int64_t in_row, int64_t out_utt, int64_t out_row) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * 256) {
    const int f = (int)(i % F);
    const int t = (int)((i / F) % T);
    const int64_t u = i / ((int64_t)F * T);
    const float* base = in + u * in_utt + f;
    // the reference sums the 2 ctx + 1 products left to right (th.sum over the stacked context)
    float acc = 0.f;
    for (int c = 0; c <= 2 * ctx; ++c)
      acc += base[(int64_t)min(max(t + c - ctx, 0), T - 1) * in_row] * scale[c];
    out[u * out_utt + (int64_t)t * out_row + f] = acc;
  }
}

// utterance ("all band") CMVN: statistics over the T x F matrix of one utterance-channel.
// One workgroup per utterance: mean, centred second moment, normalise (3 sweeps, L2 resident).
__global__ __launch_bounds__(256) void cmvn_utt_kernel(const float* __restrict__ x,
                                                       float* __restrict__ out, int64_t count,
                                                       int norm_mean, int norm_var, float eps) {
  __shared__ float s_part[4];
  const float* xu = x + (int64_t)blockIdx.x * count;
  float* ou = out + (int64_t)blockIdx.x * count;
  const int tid = threadIdx.x;
  auto block_sum = [&](float v) {
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) s_part[tid >> 6] = v;
    __syncthreads();
    return s_part[0] + s_part[1] + s_part[2] + s_part[3];
  };
  float s = 0.f;
  for (int64_t i = tid; i < count; i += 256) s += xu[i];
  const float mean = block_sum(s) / (float)count;
  float inv = 1.f;
  if (norm_var) {
    // both reference branches (mean(x_c^2) after centring / var(x, unbiased=False)) are the
    // centred second moment
    float q = 0.f;
    for (int64_t i = tid; i < count; i += 256) {
      const float c = xu[i] - mean;
      q += c * c;
    }
    inv = 1.0f / sqrtf(block_sum(q) / (float)count + eps);
  }
  const float sub = norm_mean ? mean : 0.f;
  for (int64_t i = tid; i < count; i += 256) ou[i] = (xu[i] - sub) * inv;
}

// global CMVN: (x - gmean[f]) / gstd[f]
__global__ __launch_bounds__(256) void cmvn_global_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ gmean,
                                                          const float* __restrict__ gstd,
                                                          float* __restrict__ out, int64_t total,
                                                          int F, int norm_mean, int norm_var) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * 256) {
    const int f = (int)(i % F);
    float v = x[i];
    if (norm_mean) v -= gmean[f];
    if (norm_var) v /= gstd[f];
    out[i] = v;
  }
}

static unsigned grid_for(int64_t total) {
  int64_t b = (total + 255) / 256;
  return (unsigned)(b > 8192 ? 8192 : b);
}

}  // namespace aps

using namespace aps;

extern "C" int aps_splice(const float* in, float* out, int64_t U, int64_t T, int64_t F,
                          int64_t in_utt, int64_t in_row, int32_t lctx, int32_t rctx,
                          int32_t subsampling, void* stream) {
  APS_CHECK_ARG(in && out && U > 0 && T > 0 && F > 0 && lctx >= 0 && rctx >= 0 && subsampling >= 1);
  APS_CHECK_ARG(T < (1 << 30) && F < (1 << 30));
  const int D = lctx + rctx + 1;
  const int64_t To = T / subsampling;  // the reference keeps (T // s) * s frames, then strides
  if (To == 0) return APS_OK;
  const int64_t total = U * To * D * F;
  hipLaunchKernelGGL(splice_kernel, dim3(grid_for(total)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), in, out, total, (int)T, (int)To, (int)F,
                     (int)lctx, D, (int)subsampling, in_utt, in_row);
  return aps_launch_status();
}

extern "C" int aps_delta(const float* in, float* out, const float* scale, int64_t U, int64_t T,
                         int64_t F, int32_t ctx, int64_t in_utt, int64_t in_row, int64_t out_utt,
                         int64_t out_row, void* stream) {
  APS_CHECK_ARG(in && out && scale && U > 0 && T > 0 && F > 0 && ctx >= 0);
  APS_CHECK_ARG(T < (1 << 30) && F < (1 << 30));
  const int64_t total = U * T * F;
  hipLaunchKernelGGL(delta_kernel, dim3(grid_for(total)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), in, out, scale, total, (int)T, (int)F,
                     (int)ctx, in_utt, in_row, out_utt, out_row);
  return aps_launch_status();
}

extern "C" int aps_cmvn_utterance(const float* x, float* out, int64_t U, int64_t count,
                                  int32_t norm_mean, int32_t norm_var, float eps, void* stream) {
  APS_CHECK_ARG(x && out && U > 0 && U < ((int64_t)1 << 31) && count > 0);
  hipLaunchKernelGGL(cmvn_utt_kernel, dim3((unsigned)U), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, out, count, (int)norm_mean,
                     (int)norm_var, eps);
  return aps_launch_status();
}

extern "C" int aps_cmvn_global(const float* x, const float* gmean, const float* gstd, float* out,
                               int64_t rows, int64_t F, int32_t norm_mean, int32_t norm_var,
                               void* stream) {
  APS_CHECK_ARG(x && gmean && gstd && out && rows > 0 && F > 0 && F < (1 << 30));
  const int64_t total = rows * F;
  hipLaunchKernelGGL(cmvn_global_kernel, dim3(grid_for(total)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, gmean, gstd, out, total, (int)F,
                     (int)norm_mean, (int)norm_var);
  return aps_launch_status();
}

// ------------------------------------------------------------------------------------------------
// DCCRN complex ratio masks (aps/sse/bss/dccrn.py:217-242):
//   m = (mr, mi) channels s and S + s of the decoder output [rows, 2S];  |m| = sqrt(mr^2 + mi^2 + eps)
//   m' = nl(|m|) * m / |m|;   out_s = m' (mode 0: the mask)  or  m' * X (mode 1: masked spectrogram)
// X: store [rows, 2] (re, im); out: [S, rows, 2].  nl: 0 none, 1 relu, 2 tanh, 3 softplus, 4 sigmoid
// ------------------------------------------------------------------------------------------------
namespace aps {
__device__ __forceinline__ float mask_non_linear(float v, int nl) {
  if (nl == 1) return fmaxf(v, 0.f);
  if (nl == 2) return tanhf(v);
  if (nl == 3) return v > 20.f ? v : log1pf(expf(v));  // torch softplus (threshold 20)
  if (nl == 4) return 1.0f / (1.0f + expf(-v));
  return v;
}

__global__ __launch_bounds__(256) void dccrn_mask_kernel(const float* __restrict__ dec,
                                                         const float* __restrict__ store,
                                                         float* __restrict__ out, int64_t rows,
                                                         int S, int nl, int apply, float eps) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows * S;
       i += (int64_t)gridDim.x * 256) {
    const int s = (int)(i % S);
    const int64_t r = i / S;
    const float mr = dec[r * 2 * S + s], mi = dec[r * 2 * S + S + s];
    const float mabs = sqrtf(mr * mr + mi * mi + eps);
    const float g = mask_non_linear(mabs, nl);
    const float a = g * mr / mabs, b = g * mi / mabs;
    float2 o = make_float2(a, b);
    if (apply) {
      const float2 x = *reinterpret_cast<const float2*>(store + r * 2);
      o = make_float2(x.x * a - x.y * b, x.x * b + x.y * a);
    }
    *reinterpret_cast<float2*>(out + ((int64_t)s * rows + r) * 2) = o;
  }
}

// real-valued variant (cplx = False, dccrn.py:234-241): m = nl(dec); out = m or (sr m, si m)
__global__ __launch_bounds__(256) void dccrn_real_mask_kernel(const float* __restrict__ dec,
                                                              const float* __restrict__ store,
                                                              float* __restrict__ out,
                                                              int64_t rows, int S, int nl,
                                                              int apply) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows * S;
       i += (int64_t)gridDim.x * 256) {
    const int s = (int)(i % S);
    const int64_t r = i / S;
    const float m = mask_non_linear(dec[r * S + s], nl);
    if (apply) {
      const float2 x = *reinterpret_cast<const float2*>(store + r * 2);
      *reinterpret_cast<float2*>(out + ((int64_t)s * rows + r) * 2) = make_float2(x.x * m, x.y * m);
    } else {
      out[(int64_t)s * rows + r] = m;
    }
  }
}

__global__ __launch_bounds__(256) void store_magnitude_kernel(const float* __restrict__ store,
                                                              float* __restrict__ out,
                                                              int64_t rows, float eps) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows;
       i += (int64_t)gridDim.x * 256) {
    const float2 x = *reinterpret_cast<const float2*>(store + i * 2);
    out[i] = sqrtf(x.x * x.x + x.y * x.y + eps);
  }
}
}  // namespace aps

extern "C" int aps_dccrn_mask(const float* dec, const float* store, float* out, int64_t rows,
                              int64_t S, int32_t non_linear, int32_t apply, int32_t cplx, float eps,
                              void* stream) {
  APS_CHECK_ARG(dec && out && rows > 0 && S > 0 && S < 1024 && non_linear >= 0 && non_linear <= 4);
  APS_CHECK_ARG(!apply || store);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (cplx)
    hipLaunchKernelGGL(aps::dccrn_mask_kernel, dim3(aps::grid_for(rows * S)), dim3(256), 0, st, dec,
                       store, out, rows, (int)S, (int)non_linear, (int)apply, eps);
  else
    hipLaunchKernelGGL(aps::dccrn_real_mask_kernel, dim3(aps::grid_for(rows * S)), dim3(256), 0, st,
                       dec, store, out, rows, (int)S, (int)non_linear, (int)apply);
  return aps_launch_status();
}

// frame / length arithmetic of the reference on device-resident int64 lengths in ONE launch:
// out[i] = trunc((in[i] + add) / div) + post  (STFT frame counts utils.py:653-662, Conv1d / Conv2d
// output lengths component.py:187-190, 290-297, subsampling asr.py:1017-1019); the torch form is
// three or four 5 us elementwise launches per formula
namespace aps {
__global__ void length_map_kernel(const int64_t* __restrict__ in, int64_t* __restrict__ out,
                                  int64_t n, int64_t add, int64_t div, int64_t post) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (in[i] + add) / div + post;
}
}  // namespace aps

extern "C" int aps_length_map(const int64_t* in, int64_t* out, int64_t n, int64_t add, int64_t div,
                              int64_t post, void* stream) {
  APS_CHECK_ARG(in && out && n > 0 && div != 0);
  hipLaunchKernelGGL(aps::length_map_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), in, out, n, add, div, post);
  return aps_launch_status();
}

extern "C" int aps_store_magnitude(const float* store, float* out, int64_t rows, float eps,
                                   void* stream) {
  APS_CHECK_ARG(store && out && rows > 0);
  hipLaunchKernelGGL(aps::store_magnitude_kernel, dim3(aps::grid_for(rows)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), store, out, rows, eps);
  return aps_launch_status();
}

// ------------------------------------------------------------------------------------------
// MaskNonLinear.forward (aps/sse/base.py:141-156): out = clamp(f(x) * scale, vmin, vmax) with
// f in {identity, relu, tanh, softplus, sigmoid} elementwise, or softmax over the LEADING axis
// (the sources: x [S, inner]), one thread per inner position.
// ------------------------------------------------------------------------------------------
namespace aps {

__global__ __launch_bounds__(256) void mask_nonlinear_kernel(const float* __restrict__ x,
                                                             float* __restrict__ out, int64_t inner,
                                                             int S, int code, float scale,
                                                             float vmin, float vmax) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < inner;
       i += (int64_t)gridDim.x * 256) {
    if (code == 5) {  // softmax over the S sources
      float m = -INFINITY;
      for (int s = 0; s < S; ++s) m = fmaxf(m, x[s * inner + i]);
      float sum = 0.f;
      for (int s = 0; s < S; ++s) sum += expf(x[s * inner + i] - m);
      for (int s = 0; s < S; ++s) {
        float v = expf(x[s * inner + i] - m) / sum * scale;
        out[s * inner + i] = fmaxf(fminf(v, vmax), vmin);
      }
      continue;
    }
    for (int s = 0; s < S; ++s) {
      const float v = x[s * inner + i];
      float y;
      switch (code) {
        case 1: y = fmaxf(v, 0.f); break;
        case 2: y = tanhf(v); break;
        case 3: y = v > 20.f ? v : log1pf(expf(v)); break;  // torch softplus (beta 1, threshold 20)
        case 4: y = 1.0f / (1.0f + expf(-v)); break;
        default: y = v;
      }
      out[s * inner + i] = fmaxf(fminf(y * scale, vmax), vmin);
    }
  }
}

}  // namespace aps

namespace aps {
// adjoint of mask_nonlinear_kernel: g_x = scale f'(x) g_out where the clamp let the value through (torch's
// clamp_min / clamp_max pass the gradient at the bound itself: y >= vmin and y <= vmax), the softmax's
// Jacobian over the sources.  One thread per inner position, the forward recomputed.
__global__ __launch_bounds__(256) void mask_nonlinear_backward_kernel(const float* __restrict__ x,
                                                                      const float* __restrict__ g_out,
                                                                      float* __restrict__ g_x, int64_t inner,
                                                                      int S, int code, float scale,
                                                                      float vmin, float vmax) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < inner; i += (int64_t)gridDim.x * 256) {
    if (code == 5) {
      float m = -INFINITY;
      for (int s = 0; s < S; ++s) m = fmaxf(m, x[s * inner + i]);
      float sum = 0.f;
      for (int s = 0; s < S; ++s) sum += expf(x[s * inner + i] - m);
      float dot = 0.f;  // sum_s g'_s y_s with g' = the gradient that passed the clamp, times scale
      for (int s = 0; s < S; ++s) {
        const float y = expf(x[s * inner + i] - m) / sum;
        const float v = y * scale;
        const float g = (v >= vmin && v <= vmax) ? g_out[s * inner + i] * scale : 0.f;
        dot += g * y;
      }
      for (int s = 0; s < S; ++s) {
        const float y = expf(x[s * inner + i] - m) / sum;
        const float v = y * scale;
        const float g = (v >= vmin && v <= vmax) ? g_out[s * inner + i] * scale : 0.f;
        g_x[s * inner + i] = y * (g - dot);
      }
      continue;
    }
    for (int s = 0; s < S; ++s) {
      const float v = x[s * inner + i];
      float y, d;
      switch (code) {
        case 1: y = fmaxf(v, 0.f); d = v > 0.f ? 1.f : 0.f; break;
        case 2: y = tanhf(v); d = 1.f - y * y; break;
        case 3: y = v > 20.f ? v : log1pf(expf(v)); d = v > 20.f ? 1.f : 1.0f / (1.0f + expf(-v)); break;
        case 4: y = 1.0f / (1.0f + expf(-v)); d = y * (1.f - y); break;
        default: y = v; d = 1.f;
      }
      const float o = y * scale;
      g_x[s * inner + i] = (o >= vmin && o <= vmax) ? g_out[s * inner + i] * scale * d : 0.f;
    }
  }
}
}  // namespace aps

extern "C" int aps_mask_nonlinear_backward(const float* x, const float* g_out, float* g_x, int64_t sources,
                                           int64_t inner, int32_t code, float scale, float vmin,
                                           float vmax, void* stream) {
  APS_CHECK_ARG(x && g_out && g_x && sources > 0 && inner > 0 && code >= 0 && code <= 5 &&
                sources <= INT32_MAX);
  const int64_t S = code == 5 ? sources : 1, I = code == 5 ? inner : sources * inner;
  hipLaunchKernelGGL(aps::mask_nonlinear_backward_kernel, dim3(aps::grid_for(I)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, g_out, g_x, I, (int)S, (int)code, scale, vmin,
                     vmax);
  return aps_launch_status();
}

extern "C" int aps_mask_nonlinear(const float* x, float* out, int64_t sources, int64_t inner,
                                  int32_t code, float scale, float vmin, float vmax, void* stream) {
  APS_CHECK_ARG(x && out && sources > 0 && inner > 0 && code >= 0 && code <= 5 &&
                sources <= INT32_MAX);
  // elementwise codes do not care how the tensor is split: use the widest grid
  const int64_t S = code == 5 ? sources : 1, I = code == 5 ? inner : sources * inner;
  hipLaunchKernelGGL(aps::mask_nonlinear_kernel, dim3(aps::grid_for(I)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, out, I, (int)S, (int)code, scale, vmin,
                     vmax);
  return aps_launch_status();
}
