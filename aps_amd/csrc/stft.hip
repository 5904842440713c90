// Framed STFT / iSTFT for gfx950.
//
// Forward hot path (W = 512): one workgroup = 16 consecutive frames of one (utterance, channel)
// sequence.  The raw samples the 16 frames cover are read once, coalesced, into LDS (frames
// overlap, so HBM sees every sample ~once); each frame is owned by a 16-lane slot that runs a
// 256-point complex FFT as 16 x 16 Cooley-Tukey with both radix-16 passes held in registers and a
// single LDS transpose in between; the real-FFT split then runs wave-wide so a wave stores 512
// contiguous bytes of one frame row per instruction (bin-fastest spectrogram store).
//
// Replaces aps/transform/utils.py:227-360 (_forward_stft / _inverse_stft, dense DFT by conv1d)
// and is the device counterpart of csrc/utils/{fft,stft}.cc (radix-2 RealFFT per frame).
#include "common.h"

namespace aps {

struct StftArgs {
  const float* wav;
  const float* window;
  float* out;
  int64_t num_samples;
  int64_t stride_seq;
  int64_t stride_frame;
  int64_t num_frames;
  int32_t fft_size;
  int32_t frame_len;
  int32_t frame_hop;
  int32_t num_bins;
  int32_t center;
  float pre_emphasis;
  float eps;
  float scale;
};

// padded coordinate -> sample value (reflect padding by index math, utils.py:257-260)
__device__ __forceinline__ float fetch_sample(const float* __restrict__ seq, int64_t pos,
                                              int64_t pad, int64_t S) {
  int64_t i = pos - pad;
  if (i < 0) i = -i;
  if (i >= S) {
    if (pos >= S + 2 * pad) return 0.f;
    i = 2 * (S - 1) - i;
  }
  return (i >= 0 && i < S) ? seq[i] : 0.f;
}

// ------------------------------------------------------------------------------------------
// W = 512 kernel
// ------------------------------------------------------------------------------------------
constexpr int kSlots = 16;           // frames per workgroup (16 lanes each)
constexpr int kPitch = 17;           // transpose pitch in complex words (conflict free, see DESIGN)
constexpr int kSlotWords = 16 * kPitch;  // 272 complex per slot

__host__ __device__ inline size_t stft512_lds_bytes(int hop) {
  size_t span = (size_t)(kSlots - 1) * hop + 512;
  return (256 + 260) * sizeof(cf) + 512 * sizeof(float) + (size_t)kSlots * kSlotWords * sizeof(cf) +
         span * sizeof(float);
}

template <bool PREEMPH, bool POLAR>
__global__ __launch_bounds__(256) void stft512_kernel(StftArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf* s_tw = reinterpret_cast<cf*>(smem);             // exp(-2 pi i m / 256), m < 256
  cf* s_sp = s_tw + 256;                              // exp(-2 pi i k / 512), k <= 256
  float* s_w = reinterpret_cast<float*>(s_sp + 260);  // window * scale, zero beyond frame_len
  cf* s_scr = reinterpret_cast<cf*>(s_w + 512);       // per-slot transpose / spectrum scratch
  float* s_x = reinterpret_cast<float*>(s_scr + kSlots * kSlotWords);

  const int tid = threadIdx.x;
  const int64_t seq = blockIdx.y;
  const int64_t t0 = (int64_t)blockIdx.x * kSlots;
  const int hop = a.frame_hop;
  const int L = a.frame_len;
  const int span = (kSlots - 1) * hop + 512;
  const float* __restrict__ wav = a.wav + seq * a.num_samples;
  const int64_t pad = a.center ? (L / 2) : 0;

  {  // tables
    float s, c;
    sincospif((float)tid * (1.0f / 128.0f), &s, &c);
    s_tw[tid] = {c, -s};
    sincospif((float)tid * (1.0f / 256.0f), &s, &c);
    s_sp[tid] = {c, -s};
    if (tid == 0) s_sp[256] = {-1.0f, 0.0f};
    for (int e = tid; e < 512; e += 256) s_w[e] = (e < L) ? a.window[e] * a.scale : 0.f;
  }
  const int64_t p0 = t0 * hop;
  for (int i = tid; i < span; i += 256) s_x[i] = fetch_sample(wav, p0 + i, pad, a.num_samples);
  __syncthreads();

  const int g = tid >> 4;   // slot
  const int j = tid & 15;   // lane in slot
  cf* scr = s_scr + g * kSlotWords;
  cf z[16];
  {  // pass 1: column FFTs over n1 for fixed n2 = j
    const float* fx = s_x + g * hop;
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
      const int e0 = 2 * (16 * n1 + j);
      float x0 = fx[e0], x1 = fx[e0 + 1];
      if (PREEMPH) {
        const float pe = a.pre_emphasis;
        const float xm = (e0 > 0) ? fx[e0 - 1] : 0.f;
        const float y1 = x1 - pe * x0;
        x0 = (e0 > 0) ? (x0 - pe * xm) : x0 * (1.0f - pe);
        x1 = y1;
      }
      z[n1].re = (e0 < L) ? x0 * s_w[e0] : 0.f;
      z[n1].im = (e0 + 1 < L) ? x1 * s_w[e0 + 1] : 0.f;
    }
    dft16<false>(z);
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) {
      cf v = (k1 == 0) ? z[0] : cmul(z[k1], s_tw[j * k1]);
      scr[k1 * kPitch + j] = v;
    }
  }
  __syncthreads();
  // pass 2: row FFTs over n2 for fixed k1 = j
#pragma unroll
  for (int n2 = 0; n2 < 16; ++n2) z[n2] = scr[j * kPitch + n2];
  dft16<false>(z);
  __syncthreads();
#pragma unroll
  for (int k2 = 0; k2 < 16; ++k2) scr[j + 16 * k2] = z[k2];  // Z[k1 + 16 k2], natural order
  __syncthreads();

  // real split + store, wave-wide: one frame row at a time, 64 consecutive bins per instruction
  const int wv = tid >> 6, ln = tid & 63;
#pragma unroll
  for (int gs = 0; gs < 4; ++gs) {
    const int slot = wv * 4 + gs;
    const int64_t t = t0 + slot;
    if (t >= a.num_frames) break;
    const cf* Z = s_scr + slot * kSlotWords;
    float* row = a.out + seq * a.stride_seq + t * a.stride_frame;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = ln + 64 * i;
      cf x = r2c_split(Z[k], Z[(256 - k) & 255], s_sp[k]);
      if (POLAR) x = {sqrtf(x.re * x.re + x.im * x.im + a.eps), atan2f(x.im, x.re)};
      st_cf(row + 2 * k, x);
    }
    if (ln == 0) {
      cf x = r2c_split(Z[0], Z[0], s_sp[256]);
      if (POLAR) x = {sqrtf(x.re * x.re + x.im * x.im + a.eps), atan2f(x.im, x.re)};
      st_cf(row + 512, x);
    }
  }
}

// ------------------------------------------------------------------------------------------
// General kernel: any DFT size (radix-2 Stockham in LDS for powers of two, direct DFT otherwise),
// any hop, one or two sided.  One workgroup per frame.  Correctness path for configurations off
// the benchmark (W != 512, odd hops, two-sided output).
// ------------------------------------------------------------------------------------------
template <bool POLAR>
__global__ __launch_bounds__(256) void stft_any_kernel(StftArgs a, int is_pow2) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int W = a.fft_size;
  cf* buf0 = reinterpret_cast<cf*>(smem);
  cf* buf1 = buf0 + W;
  cf* tw = buf1 + W;  // exp(-2 pi i m / W), m < W

  const int tid = threadIdx.x;
  const int64_t t = blockIdx.x, seq = blockIdx.y;
  const int L = a.frame_len;
  const float* __restrict__ wav = a.wav + seq * a.num_samples;
  const int64_t pad = a.center ? (L / 2) : 0;
  const int64_t p0 = t * a.frame_hop;
  const float pe = a.pre_emphasis;

  for (int e = tid; e < W; e += 256) {
    float s, c;
    sincospif(2.0f * (float)e / (float)W, &s, &c);
    tw[e] = {c, -s};
    float v = 0.f;
    if (e < L) {
      v = fetch_sample(wav, p0 + e, pad, a.num_samples);
      if (pe > 0.f) {
        v = (e > 0) ? v - pe * fetch_sample(wav, p0 + e - 1, pad, a.num_samples) : v * (1.0f - pe);
      }
      v *= a.window[e] * a.scale;
    }
    buf0[e] = {v, 0.f};
  }
  __syncthreads();

  cf* src = buf0;
  if (is_pow2) {
    cf* dst = buf1;
    const int half = W >> 1;
    for (int ns = 1; ns < W; ns <<= 1) {
      const int tstep = half / ns;  // twiddle stride: exp(-i pi k / ns) = tw[k * W / (2 ns)]
      for (int i = tid; i < half; i += 256) {
        const int k = i & (ns - 1);
        const int jj = ((i - k) << 1) + k;
        cf u0 = src[i];
        cf u1 = cmul(src[i + half], tw[k * tstep]);
        dst[jj] = u0 + u1;
        dst[jj + ns] = u0 - u1;
      }
      __syncthreads();
      cf* tmp = src;
      src = dst;
      dst = tmp;
    }
  }
  float* row = a.out + seq * a.stride_seq + t * a.stride_frame;
  for (int k = tid; k < a.num_bins; k += 256) {
    cf x;
    if (is_pow2) {
      x = src[k];
    } else {
      float re = 0.f, im = 0.f;
      int idx = 0;  // (k * e) mod W, incrementally
      for (int e = 0; e < L; ++e) {
        const float v = buf0[e].re;
        const cf w = tw[idx];
        re += v * w.re;
        im += v * w.im;
        idx += k;
        if (idx >= W) idx -= W;
      }
      x = {re, im};
    }
    if (POLAR) x = {sqrtf(x.re * x.re + x.im * x.im + a.eps), atan2f(x.im, x.re)};
    st_cf(row + 2 * k, x);
  }
}

// ------------------------------------------------------------------------------------------
// Inverse: (1) per-frame inverse DFT + synthesis window into a frame buffer,
//          (2) deterministic overlap-add gather with the window^2 normaliser.
// ------------------------------------------------------------------------------------------
struct IstftArgs {
  const float* spec;
  const float* window;
  float* frames;  // [seq, T, L]
  int64_t stride_seq;
  int64_t stride_frame;
  int64_t num_frames;
  int32_t fft_size;
  int32_t frame_len;
  int32_t num_bins;
  int32_t polar;
  float scale;
};

__global__ __launch_bounds__(256) void istft_frames_kernel(IstftArgs a, int is_pow2) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int W = a.fft_size;
  cf* buf0 = reinterpret_cast<cf*>(smem);
  cf* buf1 = buf0 + W;
  cf* tw = buf1 + W;  // exp(+2 pi i m / W)
  const int tid = threadIdx.x;
  const int64_t t = blockIdx.x, seq = blockIdx.y;
  const float* row = a.spec + seq * a.stride_seq + t * a.stride_frame;
  const int F = a.num_bins;
  const bool onesided = (F != W);
  for (int k = tid; k < W; k += 256) {
    float s, c;
    sincospif(2.0f * (float)k / (float)W, &s, &c);
    tw[k] = {c, s};
    // Hermitian extension (utils.py:327-332): bins F..W-1 mirror bins W-k with negated imag
    const int kk = (onesided && k >= F) ? (W - k) : k;
    cf v = ld_cf(row + 2 * kk);
    if (a.polar) {
      float sn, cs;
      sincosf(v.im, &sn, &cs);
      v = {v.re * cs, v.re * sn};
    }
    if (onesided && k >= F) v.im = -v.im;
    buf0[k] = v;
  }
  __syncthreads();
  cf* src = buf0;
  if (is_pow2) {
    cf* dst = buf1;
    const int half = W >> 1;
    for (int ns = 1; ns < W; ns <<= 1) {
      const int tstep = half / ns;
      for (int i = tid; i < half; i += 256) {
        const int k = i & (ns - 1);
        const int jj = ((i - k) << 1) + k;
        cf u0 = src[i];
        cf u1 = cmul(src[i + half], tw[k * tstep]);
        dst[jj] = u0 + u1;
        dst[jj + ns] = u0 - u1;
      }
      __syncthreads();
      cf* tmp = src;
      src = dst;
      dst = tmp;
    }
  }
  float* fr = a.frames + (seq * a.num_frames + t) * a.frame_len;
  for (int e = tid; e < a.frame_len; e += 256) {
    float v;
    if (is_pow2) {
      v = src[e].re;
    } else {
      float re = 0.f;
      int idx = 0;
      for (int k = 0; k < W; ++k) {
        const cf x = buf0[k];
        const cf w = tw[idx];
        re += x.re * w.re - x.im * w.im;
        idx += e;
        if (idx >= W) idx -= W;
      }
      v = re;
    }
    fr[e] = v * a.scale * a.window[e];
  }
}

__global__ __launch_bounds__(256) void istft_ola_kernel(const float* __restrict__ frames,
                                                        const float* __restrict__ window,
                                                        float* __restrict__ wav, int64_t T, int L,
                                                        int hop, int64_t crop, int64_t S_out,
                                                        float eps) {
  const int64_t seq = blockIdx.y;
  const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= S_out) return;
  const int64_t p = s + crop;  // position in the un-cropped overlap-add signal
  // frames t with 0 <= p - t*hop < L
  int64_t t_hi = p / hop;
  if (t_hi > T - 1) t_hi = T - 1;
  int64_t t_lo = (p - L + hop) / hop;  // ceil((p - L + 1) / hop)
  if (p - L + 1 <= 0) t_lo = 0;
  float acc = 0.f, den = 0.f;
  const float* base = frames + seq * T * L;
  for (int64_t t = t_lo; t <= t_hi; ++t) {
    const int e = (int)(p - t * hop);
    if (e >= 0 && e < L) {
      acc += base[t * L + e];
      const float w = window[e];
      den += w * w;
    }
  }
  wav[seq * S_out + s] = acc / (den + eps);
}

}  // namespace aps

// ------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------
using namespace aps;

extern "C" int64_t aps_stft_num_frames(int64_t num_samples, const aps_stft_params* p) {
  if (!p || p->frame_hop <= 0) return -1;
  int64_t s = num_samples;
  if (p->center) s += 2 * (int64_t)(p->frame_len / 2);
  if (s < p->frame_len) return 0;
  return (s - p->frame_len) / p->frame_hop + 1;
}

static bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

extern "C" int aps_stft_forward(const float* wav, int64_t num_seq, int64_t num_samples,
                                const float* window, const aps_stft_params* p, float* out,
                                int64_t stride_seq, int64_t stride_frame, int64_t num_frames,
                                void* stream) {
  APS_CHECK_ARG(wav && window && p && out);
  APS_CHECK_ARG(num_seq > 0 && num_samples > 0 && num_frames > 0);
  APS_CHECK_ARG(p->fft_size >= 2 && p->frame_len >= 1 && p->frame_len <= p->fft_size);
  APS_CHECK_ARG(p->frame_hop >= 1);
  APS_CHECK_ARG(p->num_bins == p->fft_size || p->num_bins == p->fft_size / 2 + 1);
  APS_CHECK_ARG(stride_frame >= 2 * (int64_t)p->num_bins && stride_seq >= stride_frame);
  APS_CHECK_ARG(num_frames <= aps_stft_num_frames(num_samples, p));
  if (p->center) APS_CHECK_ARG(p->frame_len / 2 < num_samples);
  if (p->fft_size > 4096) return APS_ERR_UNSUPPORTED;
  APS_CHECK_ARG(num_seq <= 65535);
  hipStream_t st = static_cast<hipStream_t>(stream);
  StftArgs a{wav,           window,       out,           num_samples,   stride_seq,
             stride_frame,  num_frames,   p->fft_size,   p->frame_len,  p->frame_hop,
             p->num_bins,   p->center,    p->pre_emphasis, p->eps,      p->scale};
  const bool fast = p->fft_size == 512 && p->num_bins == 257 && (p->frame_hop % 2 == 0) &&
                    p->frame_hop <= 512;
  if (fast) {
    dim3 grid((unsigned)((num_frames + kSlots - 1) / kSlots), (unsigned)num_seq);
    size_t lds = stft512_lds_bytes(p->frame_hop);
    const bool pe = p->pre_emphasis > 0.f;
#define APS_LAUNCH512(PE, PO)                                                                  \
  do {                                                                                         \
    if (lds > 48 * 1024)                                                                       \
      hipFuncSetAttribute(reinterpret_cast<const void*>(&stft512_kernel<PE, PO>),              \
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);               \
    hipLaunchKernelGGL((stft512_kernel<PE, PO>), grid, dim3(256), lds, st, a);                 \
  } while (0)
    if (pe && p->polar) APS_LAUNCH512(true, true);
    else if (pe) APS_LAUNCH512(true, false);
    else if (p->polar) APS_LAUNCH512(false, true);
    else APS_LAUNCH512(false, false);
#undef APS_LAUNCH512
    return aps_launch_status();
  }
  dim3 grid((unsigned)num_frames, (unsigned)num_seq);
  size_t lds = (size_t)p->fft_size * 3 * sizeof(cf);
  const int p2 = is_pow2(p->fft_size) ? 1 : 0;
  if (p->polar) {
    if (lds > 48 * 1024)
      hipFuncSetAttribute(reinterpret_cast<const void*>(&stft_any_kernel<true>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((stft_any_kernel<true>), grid, dim3(256), lds, st, a, p2);
  } else {
    if (lds > 48 * 1024)
      hipFuncSetAttribute(reinterpret_cast<const void*>(&stft_any_kernel<false>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((stft_any_kernel<false>), grid, dim3(256), lds, st, a, p2);
  }
  return aps_launch_status();
}

extern "C" int aps_stft_inverse(const float* spec, int64_t num_seq, int64_t num_frames,
                                int64_t stride_seq, int64_t stride_frame, const float* window,
                                const aps_stft_params* p, float* wav_out, int64_t num_samples_out,
                                float* workspace, void* stream) {
  APS_CHECK_ARG(spec && window && p && wav_out && workspace);
  APS_CHECK_ARG(num_seq > 0 && num_frames > 0 && num_seq <= 65535);
  APS_CHECK_ARG(p->fft_size >= 2 && p->frame_len >= 1 && p->frame_len <= p->fft_size);
  APS_CHECK_ARG(p->num_bins == p->fft_size || p->num_bins == p->fft_size / 2 + 1);
  if (p->fft_size > 4096) return APS_ERR_UNSUPPORTED;
  const int64_t crop = p->center ? (p->frame_len / 2) : 0;
  const int64_t full = (num_frames - 1) * p->frame_hop + p->frame_len;
  APS_CHECK_ARG(num_samples_out == full - 2 * crop && num_samples_out > 0);
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* frames = workspace;
  IstftArgs a{spec,        window,       frames,       stride_seq, stride_frame, num_frames,
              p->fft_size, p->frame_len, p->num_bins,  p->polar,   p->scale};
  size_t lds = (size_t)p->fft_size * 3 * sizeof(cf);
  if (lds > 48 * 1024)
    hipFuncSetAttribute(reinterpret_cast<const void*>(&istft_frames_kernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(istft_frames_kernel, dim3((unsigned)num_frames, (unsigned)num_seq), dim3(256),
                     lds, st, a, is_pow2(p->fft_size) ? 1 : 0);
  hipLaunchKernelGGL(istft_ola_kernel, dim3((unsigned)((num_samples_out + 255) / 256), (unsigned)num_seq),
                     dim3(256), 0, st, frames, window, wav_out, num_frames, p->frame_len,
                     p->frame_hop, crop, num_samples_out, p->eps);
  return aps_launch_status();
}
