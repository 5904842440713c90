// Training-time augmentation layers of the ASR feature transform (aps/transform/asr.py:116-195,
// 621-684; aps/transform/augment.py).  The random draws stay on the host, in the reference's own
// order (th.randint for the speed choice; th.rand + random.randint for the SpecAugment bands), so
// a seeded run reproduces the reference's masks; these kernels apply them.
#include "common.h"

namespace aps {

// ------------------------------------------------------------------------------------------
// Speed perturbation = polyphase resampling (augment.py:86-109): the signal is cut into blocks
// of `src` samples, a Conv1d with weight [dst, src, K] (padding (K - 1) / 2 blocks) turns every
// block into `dst` samples:
//   out[n, b dst + j] = sum_{k < K} sum_{i < src} w[j, i, k] x[n, (b + k - pad) src + i]
// Utterance n uses filter choice[n]; choice[n] = num_filters keeps the signal as it is.  Samples
// past an utterance's new length are zero (the reference pads the batch to its longest member).
// A workgroup owns 256 consecutive output samples of one utterance; the filter bank of its choice
// (dst src K <= 12 K floats for the 0.9 / 1.1 factors) is read through the L1.
// ------------------------------------------------------------------------------------------
constexpr int kMaxFilters = 8;

struct PerturbArgs {
  const float* wav;       // [N, S]
  const int64_t* choice;  // [N]
  float* out;             // [N, S_out]
  int64_t S, S_out;
  int num_filters;
  const float* w[kMaxFilters];
  int src[kMaxFilters], dst[kMaxFilters], K[kMaxFilters];
};

__global__ __launch_bounds__(256) void speed_perturb_kernel(PerturbArgs a) {
  const int64_t n = blockIdx.y;
  const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= a.S_out) return;
  const int c = (int)a.choice[n];
  const float* x = a.wav + n * a.S;
  float* y = a.out + n * a.S_out;
  if (c < 0 || c >= a.num_filters) {  // factor 1.0 (or an out-of-range draw): copy
    y[o] = o < a.S ? x[o] : 0.f;
    return;
  }
  const int src = a.src[c], dst = a.dst[c], K = a.K[c], pad = (K - 1) / 2;
  const int64_t blocks = a.S / src;
  if (o >= blocks * dst) {
    y[o] = 0.f;
    return;
  }
  const int64_t b = o / dst;
  const int j = (int)(o - b * dst);
  const float* w = a.w[c] + (int64_t)j * src * K;
  float acc = 0.f;
  const int k0 = (int)max((int64_t)0, (int64_t)pad - b);
  const int k1 = (int)min((int64_t)K, blocks + pad - b);
  for (int i = 0; i < src; ++i) {
    const float* wi = w + i * K;
    const float* xi = x + (b - pad) * src + i;
    for (int k = k0; k < k1; ++k) acc += wi[k] * xi[(int64_t)k * src];
  }
  y[o] = acc;
}

// ------------------------------------------------------------------------------------------
// SpecAugment masking (asr.py:660-684 + augment.py:13-83): x [N, C, T, F]; utterance n has
// `num_f` frequency bands and `num_t` time bands, each a (begin, length) pair (length 0: the draw
// was skipped); a value inside any band becomes 0 (mask_zero) or the mean of the whole input.
// ------------------------------------------------------------------------------------------
constexpr int kMaxBands = 16;

__global__ __launch_bounds__(256) void total_sum_kernel(const float* __restrict__ x, int64_t total,
                                                        double* __restrict__ sum) {
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256)
    acc += (double)x[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  __shared__ double s_part[4];
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(sum, s_part[0] + s_part[1] + s_part[2] + s_part[3]);
}

__global__ __launch_bounds__(256) void spec_augment_kernel(const float* __restrict__ x,
                                                           const int32_t* __restrict__ bands,
                                                           const double* __restrict__ sum,
                                                           float* __restrict__ out, int64_t rows,
                                                           int C, int T, int F, int num_f, int num_t,
                                                           double count) {
  __shared__ int s_band[2 * kMaxBands];
  const int64_t row = blockIdx.x;  // (n, c, t)
  const int t = (int)(row % T);
  const int64_t n = row / ((int64_t)T * C);
  const int nb = num_f + num_t;
  if (threadIdx.x < 2 * nb) s_band[threadIdx.x] = bands[n * 2 * nb + threadIdx.x];
  __syncthreads();
  bool row_masked = false;
  for (int q = num_f; q < nb; ++q)
    row_masked |= t >= s_band[2 * q] && t < s_band[2 * q] + s_band[2 * q + 1];
  const float fill = sum ? (float)(*sum / count) : 0.f;
  for (int f = threadIdx.x; f < F; f += 256) {
    bool masked = row_masked;
    for (int q = 0; q < num_f; ++q)
      masked |= f >= s_band[2 * q] && f < s_band[2 * q] + s_band[2 * q + 1];
    const float v = x[row * F + f];
    // mask_zero multiplies by the 0 / 1 mask (asr.py:679): keeps the sign of zero and NaN / inf
    // of the input exactly as x * 0 does
    out[row * F + f] = masked ? (sum ? fill : v * 0.f) : v;
  }
}

}  // namespace aps

using namespace aps;

extern "C" int aps_speed_perturb(const float* wav, const int64_t* choice,
                                 const float* const* filters, const int32_t* src,
                                 const int32_t* dst, const int32_t* taps, int32_t num_filters,
                                 float* out, int64_t N, int64_t S, int64_t S_out, void* stream) {
  APS_CHECK_ARG(wav && choice && out && N > 0 && S > 0 && S_out > 0 && N <= 65535);
  APS_CHECK_ARG(num_filters >= 0 && (num_filters == 0 || (filters && src && dst && taps)));
  if (num_filters > kMaxFilters) return APS_ERR_UNSUPPORTED;
  PerturbArgs a{};
  a.wav = wav, a.choice = choice, a.out = out, a.S = S, a.S_out = S_out, a.num_filters = num_filters;
  for (int c = 0; c < num_filters; ++c) {
    APS_CHECK_ARG(filters[c] && src[c] > 0 && dst[c] > 0 && taps[c] > 0 && taps[c] % 2 == 1);
    a.w[c] = filters[c], a.src[c] = src[c], a.dst[c] = dst[c], a.K[c] = taps[c];
  }
  dim3 grid((unsigned)((S_out + 255) / 256), (unsigned)N);
  hipLaunchKernelGGL(speed_perturb_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return aps_launch_status();
}

extern "C" int aps_spec_augment(const float* x, const int32_t* bands, float* out, int64_t N,
                                int64_t C, int64_t T, int64_t F, int32_t num_freq, int32_t num_time,
                                int32_t mask_zero, void* workspace, void* stream) {
  APS_CHECK_ARG(x && bands && out && N > 0 && C > 0 && T > 0 && F > 0 && num_freq >= 0 &&
                num_time >= 0 && T <= INT32_MAX && F <= INT32_MAX && C <= INT32_MAX);
  if (num_freq + num_time > kMaxBands) return APS_ERR_UNSUPPORTED;
  APS_CHECK_ARG(mask_zero || workspace);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t rows = N * C * T, total = rows * F;
  APS_CHECK_ARG(rows <= 0x7fffffff);
  double* sum = nullptr;
  if (!mask_zero) {  // x.mean() of the whole input (asr.py:681)
    sum = static_cast<double*>(workspace);
    if (aps_fill_u32(sum, 0u, 2, st) != APS_OK) return APS_ERR_LAUNCH;  // (not a memset node: common.h)
    int64_t blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(total_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, total, sum);
  }
  hipLaunchKernelGGL(spec_augment_kernel, dim3((unsigned)rows), dim3(256), 0, st, x, bands, sum, out,
                     rows, (int)C, (int)T, (int)F, (int)num_freq, (int)num_time, (double)total);
  return aps_launch_status();
}
