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
#include <stdlib.h>

#include "common.h"
#include "twiddles.h"

namespace aps {

struct StftArgs {
  const float* wav;
  const float* window;
  float* out;
  int64_t num_samples;
  int64_t stride_seq;
  int64_t stride_frame;
  int64_t num_frames;
  int64_t num_seq;
  int32_t fft_size;
  int32_t frame_len;
  int32_t frame_hop;
  int32_t num_bins;
  int32_t center;
  float pre_emphasis;
  float eps;
  float scale;
  // 1: `wav` points at int16 PCM samples and a sample is value / 32768 (exact in fp32) -- what the reference's reader
  // does on the host before the waveform ever reaches a module (aps/io/audio.py:41-44, norm=True); half the bytes
  // over PCIe and half the sample traffic of the kernel (aps_stft_forward_pcm16 / aps_stft_features_pcm16)
  int32_t pcm16;
};

constexpr float kPcm16Scale = 1.0f / 32768.0f;

// first sample of sequence `seq` (a float pointer whatever the sample type: the loaders below re-type it)
__device__ __forceinline__ const float* seq_base(const StftArgs& a, int64_t seq) {
  return a.pcm16 ? reinterpret_cast<const float*>(reinterpret_cast<const int16_t*>(a.wav) + seq * a.num_samples)
                 : a.wav + seq * a.num_samples;
}
__device__ __forceinline__ float sample_at(const float* __restrict__ seq, int64_t i, int pcm16) {
  return pcm16 ? (float)reinterpret_cast<const int16_t*>(seq)[i] * kPcm16Scale : seq[i];
}

// padded coordinate -> sample value (reflect padding by index math, utils.py:257-260)
__device__ __forceinline__ float fetch_sample(const float* __restrict__ seq, int64_t pos,
                                              int64_t pad, int64_t S, int pcm16 = 0) {
  int64_t i = pos - pad;
  if (i < 0) i = -i;
  if (i >= S) {
    if (pos >= S + 2 * pad) return 0.f;
    i = 2 * (S - 1) - i;
  }
  return (i >= 0 && i < S) ? sample_at(seq, i, pcm16) : 0.f;
}

// ------------------------------------------------------------------------------------------
// W = 512 kernel.  A workgroup is 4 INDEPENDENT wavefronts that only share read-only LDS tables;
// each wavefront owns 4 frames per iteration (16 lanes per frame) of one (utterance, channel)
// sequence and never waits for another wavefront.
//
//  * samples: lane j of a slot loads z[16 n1 + j] = (x[2m], x[2m+1]) straight from global as 16
//    independent 8-byte loads (128 contiguous bytes per slot per instruction); the 50 % frame
//    overlap is served by L1/L2.  The next iteration's samples are issued before the current
//    iteration's butterflies (register prefetch), so the FFT covers the load latency.
//  * 256-point complex FFT = 16 x 16 Cooley-Tukey, both radix-16 passes in registers, one LDS
//    transpose in between (pitch 17 -> conflict free), then the natural-order spectrum goes back
//    to LDS for the wave-wide real split: a wave stores 512 contiguous bytes of one frame row per
//    instruction into the bin-fastest store.
//  * twiddles W256^(j k1) and the window pairs are LDS tables (4 KB per workgroup), so the
//    kernel needs no scratch at 149 VGPRs -> 3 workgroups = 12 wavefronts per CU (LDS: 3 x 38.9 KB).
//  * a wavefront's LDS traffic is private to it: the ordering points are wave-level fences
//    (LDS operations of one wavefront execute in issue order), not workgroup barriers.
// ------------------------------------------------------------------------------------------
constexpr int kPitch = 17;               // transpose pitch in complex words (conflict free)
constexpr int kSlotWords = 16 * kPitch;  // 272 complex per slot
constexpr int kWaveFrames = 4;           // frames per wavefront iteration
constexpr int kWavesPerBlock = 4;


template <typename T>
__device__ __forceinline__ T lds_fetch(const T* p) {
  return *p;
}

__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct Samples {
  float2 v[16];
};

// Interior frames: 16 independent 8-byte global loads per lane (`load_frame_direct`).  Frames that
// touch the reflect padding, the end of the signal or an unaligned base are staged through the
// slot's LDS scratch with a rolled loop (`load_frame_staged`; keeps the register footprint of the
// hot path small); the whole wavefront takes that path if any of its 4 frames needs it.
// PCM (the loaders below): 0 float samples | 1 int16 PCM | 2 what a.pcm16 says (a uniform branch; the kernels whose
// register budget has room for both forms)
template <int PCM = 2>
__device__ __forceinline__ bool frame_is_inside(const StftArgs& a, const float* __restrict__ wav,
                                                int64_t t, int64_t pad) {
  const int64_t s0 = t * a.frame_hop - pad;
  const bool pcm16 = PCM == 2 ? a.pcm16 != 0 : PCM == 1;
  const bool aligned = pcm16 ? ((((uintptr_t)(reinterpret_cast<const int16_t*>(wav) + s0)) & 3) == 0)
                               : ((((uintptr_t)(wav + s0)) & 7) == 0);
  const bool inside = (t < a.num_frames) && (s0 >= 0) && (s0 + 512 <= a.num_samples) && aligned;
  return __all(inside);
}

template <int PCM = 2>
__device__ __forceinline__ void load_frame_direct(const StftArgs& a, const float* __restrict__ wav,
                                                  int64_t t, int j, int64_t pad, Samples& x) {
  if (PCM == 2 ? a.pcm16 != 0 : PCM == 1) {   // two int16 samples per lane and load: 4-byte loads, the same lanes and frame positions
    const int16_t* fx = reinterpret_cast<const int16_t*>(wav) + (t * a.frame_hop - pad) + 2 * j;
    // (the pair lands in the .x slot it is converted into: no more registers than the float path's 16 x float2)
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) x.v[n1].x = __int_as_float(*reinterpret_cast<const int32_t*>(fx + 32 * n1));
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
      const int32_t r = __float_as_int(x.v[n1].x);
      x.v[n1] = make_float2((float)(int16_t)(r & 0xffff) * kPcm16Scale, (float)(r >> 16) * kPcm16Scale);
    }
    return;
  }
  const float* fx = wav + (t * a.frame_hop - pad) + 2 * j;
#pragma unroll
  for (int n1 = 0; n1 < 16; ++n1) x.v[n1] = *reinterpret_cast<const float2*>(fx + 32 * n1);
}

// must only be called while the slot scratch holds nothing live
template <int PCM = 2>
__device__ __forceinline__ void load_frame_staged(const StftArgs& a, const float* __restrict__ wav,
                                                  int64_t t, int j, int64_t pad,
                                                  float* slot_scratch, Samples& x) {
  const int64_t p = t * a.frame_hop;  // padded coordinate of the frame start
  const bool live = t < a.num_frames;
  wave_lds_fence();
#pragma unroll 1
  for (int e = j; e < 512; e += 16)
    slot_scratch[e] = live ? fetch_sample(wav, p + e, pad, a.num_samples, PCM == 2 ? a.pcm16 : PCM) : 0.f;
  wave_lds_fence();
#pragma unroll
  for (int n1 = 0; n1 < 16; ++n1) {
    const int e0 = 2 * (16 * n1 + j);
    x.v[n1].x = slot_scratch[e0];
    x.v[n1].y = slot_scratch[e0 + 1];
  }
  wave_lds_fence();
}

template <int PCM = 2>
__device__ __forceinline__ void load_frame(const StftArgs& a, const float* __restrict__ wav,
                                           int64_t t, int j, int64_t pad, float* slot_scratch,
                                           Samples& x) {
  if (frame_is_inside<PCM>(a, wav, t, pad))
    load_frame_direct<PCM>(a, wav, t, j, pad, x);
  else
    load_frame_staged<PCM>(a, wav, t, j, pad, slot_scratch, x);
}


// (three workgroups per CU: held to the 128 VGPRs of four, the kernel spilled 18 dwords per lane and
// re-read them every tile -- scratch traffic of the order of the tile's own; at 149 VGPRs and no
// scratch a launch of 32 utterances takes 36.4 us instead of 39.5, profiles/r03_frontend_stft.txt)
template <bool PREEMPH, bool POLAR, bool PCM16 = false>
__global__ __launch_bounds__(256, 3) void stft512_wave_kernel(StftArgs a, int iters,
                                                              int64_t tiles_per_seq) {
  constexpr int PCM = PCM16 ? 1 : 0;   // (a compile-time sample type: this kernel sits at its register budget)
  __shared__ __attribute__((aligned(16))) cf s_scr[kWavesPerBlock * kWaveFrames * kSlotWords];
  __shared__ __attribute__((aligned(16))) cf s_tw[256];       // [k1][j] = W256^(j k1)
  __shared__ __attribute__((aligned(16))) float2 s_win[256];  // window pairs * scale, 0 beyond L
  __shared__ __attribute__((aligned(16))) cf s_w512[258];     // W512^k, k = 0 .. 256 (the real split)
  const int tid = threadIdx.x;
  const int wv = tid >> 6, ln = tid & 63;
  const int g = ln >> 4;  // slot = frame within the iteration
  const int j = ln & 15;  // lane in slot
  const int L = a.frame_len;
  {
    const int k1 = tid >> 4, jj = tid & 15;
    const float2 v = kW256[(jj * k1) & 255];
    s_tw[tid] = {v.x, v.y};
    const int e0 = 2 * tid;
    s_win[tid] = make_float2(e0 < L ? a.window[e0] * a.scale : 0.f,
                             e0 + 1 < L ? a.window[e0 + 1] * a.scale : 0.f);
    const float2 w = kW512[tid];
    s_w512[tid] = {w.x, w.y};
    if (tid == 0) s_w512[256] = cf{-1.f, 0.f};
  }
  __syncthreads();  // the only workgroup barrier: tables are read-only from here on

  // work item of this wavefront: `iters` consecutive 4-frame tiles of one sequence
  const int64_t groups_per_seq = (tiles_per_seq + iters - 1) / iters;
  const int64_t item = (int64_t)blockIdx.x * kWavesPerBlock + wv;
  const int64_t seq = item / groups_per_seq;
  if (seq >= (int64_t)a.num_seq) return;
  const int64_t tile0 = (item % groups_per_seq) * iters;
  const float* __restrict__ wav = seq_base(a, seq);
  const int64_t pad = a.center ? (L / 2) : 0;
  const float pe = a.pre_emphasis;

  cf* wscr = s_scr + wv * (kWaveFrames * kSlotWords);
  cf* scr = wscr + g * kSlotWords;
  Samples cur;
  load_frame<PCM>(a, wav, tile0 * kWaveFrames + g, j, pad, reinterpret_cast<float*>(scr), cur);
  // frame rows of 257 bins are 2056 bytes: stored row by row, every row straddles the 128-byte lines
  // at both of its ends (PMC: 83.6 MB written for a 65.5 MB spectrogram).  When the rows of a tile
  // follow each other in memory the wave stores the tile's 4 x 257 bins as ONE run, 512 contiguous
  // bytes per instruction: only the two ends of the 8 224-byte run share a line with a neighbour.
  const bool rows_contiguous = a.stride_frame == 2 * 257;

#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    const int64_t tbase = (tile0 + it) * kWaveFrames;
    if (tbase >= a.num_frames) break;
    cf z[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
      const int e0 = 2 * (16 * n1 + j);
      float x0 = cur.v[n1].x, x1 = cur.v[n1].y;
      if (PREEMPH) {
        // previous sample x[e0 - 1]: the .y of lane j-1 (same n1) or of lane 15 (n1 - 1)
        const float left = __shfl_up(cur.v[n1].y, 1, 16);
        const float wrap = __shfl(cur.v[n1 > 0 ? n1 - 1 : 0].y, (ln & 48) | 15, 64);
        const float xm = (j > 0) ? left : wrap;
        const float y1 = x1 - pe * x0;
        x0 = (e0 > 0) ? (x0 - pe * xm) : x0 * (1.0f - pe);
        x1 = y1;
      }
      const float2 w = lds_fetch(&s_win[16 * n1 + j]);
      // select (not multiply by 0): samples beyond the frame are never read by the reference
      z[n1].re = (e0 < L) ? x0 * w.x : 0.f;
      z[n1].im = (e0 + 1 < L) ? x1 * w.y : 0.f;
    }
    // `cur` is consumed: issue the next tile's sample loads into the same registers now, so
    // they are in flight under the butterflies / split / stores of this tile
    const bool more = it + 1 < iters;
    const bool ahead = more && frame_is_inside<PCM>(a, wav, tbase + kWaveFrames + g, pad);
    if (ahead) load_frame_direct<PCM>(a, wav, tbase + kWaveFrames + g, j, pad, cur);
    dft16<false>(z);
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) {
      const cf v = (k1 == 0) ? z[0] : cmul(z[k1], lds_fetch(&s_tw[k1 * 16 + j]));
      scr[k1 * kPitch + j] = v;
    }
    wave_lds_fence();
#pragma unroll
    for (int n2 = 0; n2 < 16; ++n2) z[n2] = lds_fetch(&scr[j * kPitch + n2]);
    dft16<false>(z);
    wave_lds_fence();
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) scr[j + 16 * k2] = z[k2];  // Z[k1 + 16 k2], natural order
    wave_lds_fence();

    // real split + store, wave-wide, 64 consecutive bins per instruction (bin k of a row needs
    // Z[k], Z[256 - k] and W512^k; k = 256 reads Z[0] twice with W = -1: the table's last entry)
    if (rows_contiguous && tbase + kWaveFrames <= a.num_frames) {
      float* run = a.out + seq * a.stride_seq + tbase * a.stride_frame;
#pragma unroll
      for (int i = 0; i < (kWaveFrames * 257 + 63) / 64; ++i) {
        const int e = ln + 64 * i;
        if (e < kWaveFrames * 257) {
          const int gs = (e >= 257) + (e >= 514) + (e >= 771);
          const int k = e - 257 * gs;
          const cf* Z = wscr + gs * kSlotWords;
          cf x = r2c_split(lds_fetch(&Z[k & 255]), lds_fetch(&Z[(256 - k) & 255]), lds_fetch(&s_w512[k]));
          if (POLAR) x = {sqrtf(x.re * x.re + x.im * x.im + a.eps), atan2f(x.im, x.re)};
          st_cf(run + 2 * e, x);
        }
      }
    } else {
#pragma unroll
      for (int gs = 0; gs < kWaveFrames; ++gs) {
        const int64_t t = tbase + gs;
        if (t >= a.num_frames) break;
        const cf* Z = wscr + gs * kSlotWords;
        float* row = a.out + seq * a.stride_seq + t * a.stride_frame;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int k = ln + 64 * i;
          cf x = r2c_split(lds_fetch(&Z[k]), lds_fetch(&Z[(256 - k) & 255]), lds_fetch(&s_w512[k]));
          if (POLAR) x = {sqrtf(x.re * x.re + x.im * x.im + a.eps), atan2f(x.im, x.re)};
          st_cf(row + 2 * k, x);
        }
        if (ln == 0) {
          cf x = r2c_split(lds_fetch(&Z[0]), lds_fetch(&Z[0]), cf{-1.f, 0.f});
          if (POLAR) x = {sqrtf(x.re * x.re + x.im * x.im + a.eps), atan2f(x.im, x.re)};
          st_cf(row + 512, x);
        }
      }
    }
    wave_lds_fence();  // the next iteration overwrites the scratch
    if (more && !ahead)
      load_frame_staged<PCM>(a, wav, tbase + kWaveFrames + g, j, pad, reinterpret_cast<float*>(scr), cur);
  }
}

// ------------------------------------------------------------------------------------------
// W = 512 kernel fused with the feature chain: a workgroup owns `iters` 4-frame tiles of ONE
// utterance, wavefront w runs the FFT of channel w (same machinery as stft512_wave_kernel).
// After the real split every wavefront leaves the unit vectors x / |x| of its channel in its own
// LDS scratch (in place of the spectrum rows it just consumed) and the reference channel's
// wavefront keeps |X| -> power -> log in registers; one workgroup barrier later the IPD pairs are
// formed from the channels' LDS rows and the reference wavefront normalises (CMVN) and stores its
// rows.  The spectrogram is written (WRITE_X) but never read back: the feature pass over X
// (2 MB / utterance) and one launch disappear.  With C = 1 and WRITE_X = false this is the
// AsrTransform chain STFT -> |X| -> [mel] -> log -> cmvn without materialising X at all.
// ------------------------------------------------------------------------------------------
struct FusedArgs {
  StftArgs st;
  float* feats;
  const int32_t* mel_start;
  const int32_t* mel_len;
  const int32_t* mel_off;
  const float* mel_w;
  const int32_t* pair_l;
  const int32_t* pair_r;
  int32_t* nan_count;
  int32_t C, D, ref_channel, power, num_mels, apply_log, norm_mean, norm_var, num_pairs, ipd_sin;
  float log_eps, log_lower_bound, cmvn_eps;
};

template <bool PREEMPH, bool WRITE_X, int CW>
__global__ __launch_bounds__(64 * CW, (CW <= 4) ? 3 : (CW == 8 ? 2 : 1)) void stft512_feat_kernel(
    FusedArgs fa, int iters, int64_t tiles_per_seq) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf* s_scr = reinterpret_cast<cf*>(smem);                 // [CW][4][272]
  cf* s_tw = s_scr + CW * kWaveFrames * kSlotWords;        // [k1][j] = W256^(j k1)
  float2* s_win = reinterpret_cast<float2*>(s_tw + 256);   // window pairs * scale
  float* s_val = reinterpret_cast<float*>(s_win + 256);    // [4][260] reference channel values
  const StftArgs& a = fa.st;
  const int tid = threadIdx.x;
  const int wv = tid >> 6, ln = tid & 63;
  const int g = ln >> 4, j = ln & 15;
  const int L = a.frame_len;
  for (int e = tid; e < 256; e += 64 * CW) {
    const int k1 = e >> 4, jj = e & 15;
    const float2 v = kW256[(jj * k1) & 255];
    s_tw[e] = {v.x, v.y};
    const int e0 = 2 * e;
    s_win[e] = make_float2(e0 < L ? a.window[e0] * a.scale : 0.f,
                           e0 + 1 < L ? a.window[e0 + 1] * a.scale : 0.f);
  }
  __syncthreads();

  const int C = fa.C;
  const bool chan = wv < C;  // wavefronts beyond C only keep the barriers company
  const int64_t groups_per_seq = (tiles_per_seq + iters - 1) / iters;
  const int64_t n = blockIdx.x / groups_per_seq;
  const int64_t tile0 = (blockIdx.x % groups_per_seq) * iters;
  const int64_t seq = n * C + (chan ? wv : 0);
  const float* __restrict__ wav = seq_base(a, seq);
  const int64_t pad = a.center ? (L / 2) : 0;
  const float pe = a.pre_emphasis;
  const int F = 257;
  const bool has_mag = fa.ref_channel >= 0;
  const bool is_ref = has_mag && wv == fa.ref_channel;
  const int D0 = has_mag ? (fa.num_mels > 0 ? fa.num_mels : F) : 0;

  cf sp[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float2 v = kW512[ln + 64 * i];
    sp[i] = {v.x, v.y};
  }
  cf* wscr = s_scr + wv * (kWaveFrames * kSlotWords);
  cf* scr = wscr + g * kSlotWords;
  bool bad = false;
  Samples cur;
  if (chan) load_frame(a, wav, tile0 * kWaveFrames + g, j, pad, reinterpret_cast<float*>(scr), cur);

#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    const int64_t tbase = (tile0 + it) * kWaveFrames;
    if (tbase >= a.num_frames) break;  // uniform over the workgroup
    const bool more = it + 1 < iters;
    bool ahead = false;
    if (chan) {
      cf z[16];
#pragma unroll
      for (int n1 = 0; n1 < 16; ++n1) {
        const int e0 = 2 * (16 * n1 + j);
        float x0 = cur.v[n1].x, x1 = cur.v[n1].y;
        if (PREEMPH) {
          const float left = __shfl_up(cur.v[n1].y, 1, 16);
          const float wrap = __shfl(cur.v[n1 > 0 ? n1 - 1 : 0].y, (ln & 48) | 15, 64);
          const float xm = (j > 0) ? left : wrap;
          const float y1 = x1 - pe * x0;
          x0 = (e0 > 0) ? (x0 - pe * xm) : x0 * (1.0f - pe);
          x1 = y1;
        }
        const float2 w = s_win[16 * n1 + j];
        z[n1].re = (e0 < L) ? x0 * w.x : 0.f;
        z[n1].im = (e0 + 1 < L) ? x1 * w.y : 0.f;
      }
      dft16<false>(z);
#pragma unroll
      for (int k1 = 0; k1 < 16; ++k1)
        scr[k1 * kPitch + j] = (k1 == 0) ? z[0] : cmul(z[k1], s_tw[k1 * 16 + j]);
      wave_lds_fence();
#pragma unroll
      for (int n2 = 0; n2 < 16; ++n2) z[n2] = scr[j * kPitch + n2];
      dft16<false>(z);
      wave_lds_fence();
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) scr[j + 16 * k2] = z[k2];
      wave_lds_fence();

#pragma unroll
      for (int gs = 0; gs < kWaveFrames; ++gs) {
        const int64_t t = tbase + gs;
        const bool live = t < a.num_frames;
        cf* Z = wscr + gs * kSlotWords;
        cf zk[4], zc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int k = ln + 64 * i;
          zk[i] = Z[k];
          zc[i] = Z[(256 - k) & 255];
        }
        const cf z0 = Z[0];
        wave_lds_fence();  // the row is fully in registers: it may now be overwritten in place
        float* row = WRITE_X ? a.out + seq * a.stride_seq + t * a.stride_frame : nullptr;
        float* mrow = reinterpret_cast<float*>(Z);  // mel: magnitudes of this frame
#pragma unroll
        for (int i = 0; i <= 4; ++i) {
          const int k = (i < 4) ? ln + 64 * i : 256;
          const bool own = (i < 4) || (ln == 0);
          const cf x = (i < 4) ? r2c_split(zk[i], zc[i], sp[i]) : r2c_split(z0, z0, cf{-1.f, 0.f});
          if (WRITE_X && live && own) st_cf(row + 2 * k, x);
          if (fa.num_pairs > 0 && own) {
            const float2 u = unit_vector(x);
            Z[k] = {u.x, u.y};
          }
          float v = 0.f;
          if (is_ref && own) {
            v = sqrtf(x.re * x.re + x.im * x.im);
            if (fa.power == 2) v = v * v;
            if (fa.num_mels > 0) {
              mrow[k] = v;
            } else if (fa.apply_log) {
              v = log_feature(v, fa.log_eps, fa.log_lower_bound);
            }
          }
          if (is_ref && own) s_val[gs * 260 + k] = v;
        }
      }
      if (is_ref && fa.num_mels > 0) {  // banded mel product per frame, then log
        wave_lds_fence();
#pragma unroll
        for (int gs = 0; gs < kWaveFrames; ++gs) {
          const float* mrow = reinterpret_cast<const float*>(wscr + gs * kSlotWords);
#pragma unroll
          for (int i = 0; i <= 4; ++i) {
            const int d = ln + 64 * i;
            float v = 0.f;
            if (i < 4 && d < D0) {
              const int st = fa.mel_start[d], len = fa.mel_len[d];
              const float* w = fa.mel_w + fa.mel_off[d];
              for (int q = 0; q < len; ++q) v += w[q] * mrow[st + q];
              if (fa.apply_log) v = log_feature(v, fa.log_eps, fa.log_lower_bound);
            }
            if (i < 4 && d < D0) s_val[gs * 260 + d] = v;
          }
        }
      }
    }
    __syncthreads();  // every channel's unit vectors of this tile are in LDS

    if (chan) {
      // ---- IPD: (pair, frame) items round-robin over the channel wavefronts ----
      for (int item = wv; item < fa.num_pairs * kWaveFrames; item += C) {
        const int gs = item & (kWaveFrames - 1), p = item / kWaveFrames;
        const int64_t t = tbase + gs;
        if (t >= a.num_frames) continue;
        const cf* ul = s_scr + (fa.pair_l[p] * kWaveFrames + gs) * kSlotWords;
        const cf* ur = s_scr + (fa.pair_r[p] * kWaveFrames + gs) * kSlotWords;
        float* oc = fa.feats + (n * a.num_frames + t) * (int64_t)fa.D + D0 + (int64_t)p * F;
        float* os = oc + (int64_t)fa.num_pairs * F;
#pragma unroll
        for (int i = 0; i <= 4; ++i) {
          const int k = (i < 4) ? ln + 64 * i : 256;
          if (i == 4 && ln != 0) continue;
          const cf l = ul[k], r = ur[k];
          const float cd = l.re * r.re + l.im * r.im;
          bad |= (cd != cd);
          oc[k] = cd;
          if (fa.ipd_sin) os[k] = l.im * r.re - l.re * r.im;
        }
      }
      // ---- spectral branch: per-frame CMVN (two wave reductions) + store; frame gs is handled
      // by wavefront gs mod C so the four frames proceed in parallel ----
      if (has_mag) {
        for (int gs = wv; gs < kWaveFrames; gs += C) {
          const int64_t t = tbase + gs;
          if (t >= a.num_frames) break;
          float* orow = fa.feats + (n * a.num_frames + t) * (int64_t)fa.D;
          float o[5];
#pragma unroll
          for (int i = 0; i <= 4; ++i) {
            const int d = (i < 4) ? ln + 64 * i : 256;
            const bool in = (i < 4) ? d < D0 : (ln == 0 && d < D0);
            o[i] = in ? s_val[gs * 260 + d] : 0.f;
          }
          if (fa.norm_mean || fa.norm_var) {
            const float mean = wave_sum(o[0] + o[1] + o[2] + o[3] + o[4]) / (float)D0;
            float sq = 0.f;
#pragma unroll
            for (int i = 0; i <= 4; ++i) {
              const int d = (i < 4) ? ln + 64 * i : 256;
              const bool in = (i < 4) ? d < D0 : (ln == 0 && d < D0);
              const float c = o[i] - mean;
              if (in) sq += c * c;
              if (fa.norm_mean) o[i] = c;
            }
            const float var = wave_sum(sq) / (float)D0;
            if (fa.norm_var) {
              const float sd = sqrtf(var + fa.cmvn_eps);
#pragma unroll
              for (int i = 0; i <= 4; ++i) o[i] = o[i] / sd;
            }
          }
#pragma unroll
          for (int i = 0; i <= 4; ++i) {
            const int d = (i < 4) ? ln + 64 * i : 256;
            const bool in = (i < 4) ? d < D0 : (ln == 0 && d < D0);
            if (in) {
              bad |= (o[i] != o[i]);
              orow[d] = o[i];
            }
          }
        }
      }
    }
    __syncthreads();  // the next tile's FFT overwrites the scratch
    if (chan && more && !ahead)
      load_frame_staged(a, wav, tbase + kWaveFrames + g, j, pad, reinterpret_cast<float*>(scr), cur);
  }
  if (fa.nan_count != nullptr && __any(bad) && ln == 0) atomicAdd(fa.nan_count, 1);
}

// ------------------------------------------------------------------------------------------
// W = 512, C = 2 .. 4 channels, STFT + the enhancement front end's features in ONE pass with NO
// exchange between wavefronts (round 4; SURVEY.md 8(d) P1: X is written once and never re-read):
// a wavefront owns ONE frame of ALL channels per iteration -- slot g of stft512_wave_kernel's four
// 16-lane slots runs channel g instead of frame g -- so after the wave-wide real split lane `ln`
// holds bins ln + 64 i of every channel of the frame: the unit vectors x / |x| go back into the
// wave's own scratch rows (in place of the spectra just consumed), the IPD of a pair is the dot
// product of two of them read back by the same lanes, the reference channel's |X| -> power -> log
// stays in registers for the per-frame CMVN (two wave reductions).  Same tables, same FFT, same
// sample prefetch and wave-level fences as stft512_wave_kernel; stft512_feat_kernel's form (a
// wavefront per channel, four frames per tile) needs two workgroup barriers per tile and measured
// 57 - 72 us against 27 + 25 for the two stand-alone launches in round 1.
// A wavefront walks `iters` consecutive frames, so the 2056-byte frame rows of a channel leave one
// after the other and the 50 % frame overlap of its samples is served by L1.
// Domain: num_mels == 0 (the enhancement chain: log-magnitude + IPD), no pre-emphasis.
// ------------------------------------------------------------------------------------------
template <bool WRITE_X, bool PCM16 = false>
__global__ __launch_bounds__(256, 3) void stft512_frame_feat_kernel(FusedArgs fa, int iters,
                                                                    int64_t items_per_utt) {
  constexpr int PCM = PCM16 ? 1 : 0;   // (a compile-time sample type: the float path's code is what it was)
  __shared__ __attribute__((aligned(16))) cf s_scr[kWavesPerBlock * kWaveFrames * kSlotWords];
  __shared__ __attribute__((aligned(16))) cf s_tw[256];
  __shared__ __attribute__((aligned(16))) float2 s_win[256];
  __shared__ __attribute__((aligned(16))) cf s_w512[258];
  const StftArgs& a = fa.st;
  const int tid = threadIdx.x;
  const int wv = tid >> 6, ln = tid & 63;
  const int g = ln >> 4, j = ln & 15;  // slot = channel, lane in slot
  const int L = a.frame_len;
  {
    const int k1 = tid >> 4, jj = tid & 15;
    const float2 v = kW256[(jj * k1) & 255];
    s_tw[tid] = {v.x, v.y};
    const int e0 = 2 * tid;
    s_win[tid] = make_float2(e0 < L ? a.window[e0] * a.scale : 0.f,
                             e0 + 1 < L ? a.window[e0 + 1] * a.scale : 0.f);
    const float2 w = kW512[tid];
    s_w512[tid] = {w.x, w.y};
    if (tid == 0) s_w512[256] = cf{-1.f, 0.f};
  }
  __syncthreads();  // the only workgroup barrier: tables are read-only from here on

  const int C = fa.C;
  const int64_t item = (int64_t)blockIdx.x * kWavesPerBlock + wv;
  const int64_t n = item / items_per_utt;
  if (n * C >= (int64_t)a.num_seq) return;
  const int64_t t0 = (item % items_per_utt) * iters;
  // (a slot beyond C runs channel C - 1 once more and drops what it computes)
  const int ch = g < C ? g : C - 1;
  const float* __restrict__ wav = seq_base(a, n * C + ch);
  const int64_t pad = a.center ? (L / 2) : 0;
  const int F = 257;
  const bool has_mag = fa.ref_channel >= 0;
  const int D0 = has_mag ? F : 0;

  cf* wscr = s_scr + wv * (kWaveFrames * kSlotWords);
  cf* scr = wscr + g * kSlotWords;
  bool bad = false;
  Samples cur;
  load_frame<PCM>(a, wav, t0, j, pad, reinterpret_cast<float*>(scr), cur);

#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    const int64_t t = t0 + it;
    if (t >= a.num_frames) break;
    cf z[16];
    if (L == 512) {   // (uniform; the usual case: every sample of the 512-point block is inside the frame, no select)
#pragma unroll
      for (int n1 = 0; n1 < 16; ++n1) {
        const float2 w = lds_fetch(&s_win[16 * n1 + j]);
        z[n1].re = cur.v[n1].x * w.x;
        z[n1].im = cur.v[n1].y * w.y;
      }
    } else {
#pragma unroll
      for (int n1 = 0; n1 < 16; ++n1) {
        const int e0 = 2 * (16 * n1 + j);
        const float2 w = lds_fetch(&s_win[16 * n1 + j]);
        z[n1].re = (e0 < L) ? cur.v[n1].x * w.x : 0.f;
        z[n1].im = (e0 + 1 < L) ? cur.v[n1].y * w.y : 0.f;
      }
    }
    // the next frame's samples are requested now: in flight under this frame's butterflies
    const bool more = it + 1 < iters && t + 1 < a.num_frames;
    const bool ahead = more && frame_is_inside<PCM>(a, wav, t + 1, pad);
    if (ahead) load_frame_direct<PCM>(a, wav, t + 1, j, pad, cur);
    dft16<false>(z);
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) {
      const cf v = (k1 == 0) ? z[0] : cmul(z[k1], lds_fetch(&s_tw[k1 * 16 + j]));
      scr[k1 * kPitch + j] = v;
    }
    wave_lds_fence();
#pragma unroll
    for (int n2 = 0; n2 < 16; ++n2) z[n2] = lds_fetch(&scr[j * kPitch + n2]);
    dft16<false>(z);
    wave_lds_fence();
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) scr[j + 16 * k2] = z[k2];  // Z[k1 + 16 k2], natural order
    wave_lds_fence();

    // ---- real split of every channel's row, wave-wide; X leaves, x / |x| goes back in place ----
    float o[5];  // the reference channel's |X| -> power -> log, bins ln + 64 i (and 256 in lane 0)
#pragma unroll
    for (int i = 0; i < 5; ++i) o[i] = 0.f;
    cf z0_mine = {0.f, 0.f};   // lane c < C: Z_c[0], whose real split is bin 256 of channel c (handled behind the loop)
#pragma unroll
    for (int c = 0; c < kWaveFrames; ++c) {
      if (c >= C) break;  // (uniform)
      cf* Z = wscr + c * kSlotWords;
      cf zk[4], zc[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = ln + 64 * i;
        zk[i] = lds_fetch(&Z[k]);
        zc[i] = lds_fetch(&Z[(256 - k) & 255]);
      }
      const cf z0 = lds_fetch(&Z[0]);
      z0_mine = (ln == c) ? z0 : z0_mine;
      wave_lds_fence();  // the row is in registers: it may be overwritten in place
      float* row = WRITE_X ? a.out + (n * C + c) * a.stride_seq + t * a.stride_frame : nullptr;
      const bool is_ref = has_mag && c == fa.ref_channel;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = ln + 64 * i;
        const cf x = r2c_split(zk[i], zc[i], lds_fetch(&s_w512[k]));
        if (WRITE_X) st_cf(row + 2 * k, x);
        if (fa.num_pairs > 0) {
          const float2 u = unit_vector(x);
          Z[k] = {u.x, u.y};
        }
        if (is_ref) {
          float v = cabs_fast(x);   // (v_sqrt_f32, 1 ulp: the IEEE expansion is 10 more instructions per value)
          if (fa.power == 2) v = v * v;
          if (fa.apply_log) v = log_feature(v, fa.log_eps, fa.log_lower_bound);
          o[i] = v;
        }
      }
    }
    // bin 256 (the Nyquist bin: the real split of Z[0] against -1) of EVERY channel in one go, lane c = channel c --
    // inside the channel loop it was a fifth wave-wide split + unit vector + magnitude per channel for one live lane
    {
      const cf x = r2c_split(z0_mine, z0_mine, cf{-1.f, 0.f});
      const int cc = ln < C ? ln : 0;
      if (ln < C) {
        if (WRITE_X) st_cf(a.out + (n * C + cc) * a.stride_seq + t * a.stride_frame + 2 * 256, x);
        if (fa.num_pairs > 0) {
          const float2 u = unit_vector(x);
          wscr[cc * kSlotWords + 256] = {u.x, u.y};
        }
      }
      if (has_mag) {
        float v = cabs_fast(x);
        if (fa.power == 2) v = v * v;
        if (fa.apply_log) v = log_feature(v, fa.log_eps, fa.log_lower_bound);
        v = __shfl(v, fa.ref_channel, 64);   // the reference channel's lane -> every lane; lane 0 keeps it
        o[4] = (ln == 0) ? v : 0.f;
      }
    }
    wave_lds_fence();  // every channel's unit vectors of this frame are in the wave's scratch

    float* orow = fa.feats + (n * a.num_frames + t) * (int64_t)fa.D;
    // ---- IPD: cos (and sin) of the phase difference = real (imaginary) part of u_l conj(u_r) ----
    for (int p = 0; p < fa.num_pairs; ++p) {
      const cf* ul = wscr + fa.pair_l[p] * kSlotWords;
      const cf* ur = wscr + fa.pair_r[p] * kSlotWords;
      float* oc = orow + D0 + (int64_t)p * F;
      float* os = oc + (int64_t)fa.num_pairs * F;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = ln + 64 * i;
        const cf l = lds_fetch(&ul[k]), r = lds_fetch(&ur[k]);
        const float cd = l.re * r.re + l.im * r.im;
        bad |= (cd != cd);
        oc[k] = cd;
        if (fa.ipd_sin) os[k] = l.im * r.re - l.re * r.im;
      }
    }
    if (ln < fa.num_pairs) {   // bin 256 of pair `ln` (one pass for all pairs instead of a one-lane pass per pair)
      const cf l = lds_fetch(&wscr[fa.pair_l[ln] * kSlotWords + 256]), r = lds_fetch(&wscr[fa.pair_r[ln] * kSlotWords + 256]);
      float* oc = orow + D0 + (int64_t)ln * F;
      const float cd = l.re * r.re + l.im * r.im;
      bad |= (cd != cd);
      oc[256] = cd;
      if (fa.ipd_sin) oc[(int64_t)fa.num_pairs * F + 256] = l.im * r.re - l.re * r.im;
    }
    // ---- spectral branch: per-frame CMVN over the 257 bins (two wave reductions), then the row ----
    if (has_mag) {
      if (fa.norm_mean || fa.norm_var) {
        const float mean = wave_sum(o[0] + o[1] + o[2] + o[3] + o[4]) / (float)F;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i <= 4; ++i) {
          const bool own = (i < 4) || (ln == 0);
          const float c = o[i] - mean;
          if (own) sq += c * c;
          if (fa.norm_mean) o[i] = c;
        }
        const float var = wave_sum(sq) / (float)F;
        if (fa.norm_var) {
          const float isd = 1.0f / sqrtf(var + fa.cmvn_eps);   // (one division per frame, not one per value)
#pragma unroll
          for (int i = 0; i <= 4; ++i) o[i] = o[i] * isd;
        }
      }
#pragma unroll
      for (int i = 0; i <= 4; ++i) {
        const int k = (i < 4) ? ln + 64 * i : 256;
        if (i == 4 && ln != 0) continue;
        bad |= (o[i] != o[i]);
        orow[k] = o[i];
      }
    }
    wave_lds_fence();  // the next frame's FFT overwrites the scratch
    if (more && !ahead)
      load_frame_staged<PCM>(a, wav, t + 1, j, pad, reinterpret_cast<float*>(scr), cur);
  }
  if (fa.nan_count != nullptr && __any(bad) && ln == 0) atomicAdd(fa.nan_count, 1);
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
  const float* __restrict__ wav = seq_base(a, seq);
  const int64_t pad = a.center ? (L / 2) : 0;
  const int64_t p0 = t * a.frame_hop;
  const float pe = a.pre_emphasis;

  for (int e = tid; e < W; e += 256) {
    float s, c;
    sincospif(2.0f * (float)e / (float)W, &s, &c);
    tw[e] = {c, -s};
    float v = 0.f;
    if (e < L) {
      v = fetch_sample(wav, p0 + e, pad, a.num_samples, a.pcm16);
      if (pe > 0.f) {
        v = (e > 0) ? v - pe * fetch_sample(wav, p0 + e - 1, pad, a.num_samples, a.pcm16) : v * (1.0f - pe);
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
  float edge_weight;  // onesided input: factor on bins 0 and W/2 (1; 2 for the STFT adjoint)
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
    if (onesided && (k == 0 || k == F - 1)) v = cscale(v, a.edge_weight);
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

// ------------------------------------------------------------------------------------------
// Adjoints (backward of the two transforms; both are linear maps, so the backward of one is the
// other's machinery with different weights -- see aps_stft_backward / aps_stft_inverse_backward)
// ------------------------------------------------------------------------------------------
// overlap-add of the adjoint frames WITHOUT normaliser into grad_wav [seq, S]; with `pad` > 0 the
// forward reflect-padded the signal (utils.py:257-260), so a sample also collects the padded
// positions that mirrored it
__global__ __launch_bounds__(256) void stft_adjoint_ola_kernel(const float* __restrict__ frames,
                                                               float* __restrict__ grad_wav,
                                                               int64_t T, int L, int hop,
                                                               int64_t pad, int64_t S) {
  const int64_t seq = blockIdx.y;
  const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= S) return;
  const float* base = frames + seq * T * L;
  auto at = [&](int64_t p) {  // sum over frames covering padded position p
    int64_t t_hi = p / hop;
    if (t_hi > T - 1) t_hi = T - 1;
    int64_t t_lo = (p - L + hop) / hop;
    if (p - L + 1 <= 0) t_lo = 0;
    float acc = 0.f;
    for (int64_t t = t_lo; t <= t_hi; ++t) {
      const int64_t e = p - t * hop;
      if (e >= 0 && e < L) acc += base[t * L + e];
    }
    return acc;
  };
  float g = at(s + pad);
  if (pad > 0) {
    if (s >= 1 && s <= pad) g += at(pad - s);                          // left mirror
    const int64_t j = S - 1 - s;
    if (j >= 1 && j <= pad) g += at(pad + S - 1 + j);                  // right mirror
  }
  grad_wav[seq * S + s] = g;
}

// u [seq, full] = grad_wav / (OLA(window^2) + eps) placed back at its un-cropped position (zeros
// in the margins the centre crop removed): the input of the forward machinery for the iSTFT adjoint
__global__ __launch_bounds__(256) void istft_adjoint_normalize_kernel(
    const float* __restrict__ grad_wav, const float* __restrict__ window, float* __restrict__ u,
    int64_t T, int L, int hop, int64_t crop, int64_t S_out, int64_t full, float eps) {
  const int64_t seq = blockIdx.y;
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= full) return;
  const int64_t s = p - crop;
  float v = 0.f;
  if (s >= 0 && s < S_out) {
    int64_t t_hi = p / hop;
    if (t_hi > T - 1) t_hi = T - 1;
    int64_t t_lo = (p - L + hop) / hop;
    if (p - L + 1 <= 0) t_lo = 0;
    float den = 0.f;
    for (int64_t t = t_lo; t <= t_hi; ++t) {
      const int64_t e = p - t * hop;
      if (e >= 0 && e < L) den += window[e] * window[e];
    }
    v = grad_wav[seq * S_out + s] / (den + eps);
  }
  u[seq * full + p] = v;
}

// bins 0 and W/2 of every frame of a onesided store times `w`
__global__ __launch_bounds__(256) void store_edge_scale_kernel(float* __restrict__ store,
                                                               int64_t stride_seq,
                                                               int64_t stride_frame, int64_t T,
                                                               int F, int64_t rows, float w) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * 2) return;
  const int64_t r = i >> 1;
  float* row = store + (r / T) * stride_seq + (r % T) * stride_frame + ((i & 1) ? 2 * (F - 1) : 0);
  row[0] *= w;
  row[1] *= w;
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

static int stft_forward_impl(const float* wav, int pcm16, int64_t num_seq, int64_t num_samples,
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
             stride_frame,  num_frames,   num_seq,       p->fft_size,   p->frame_len,  p->frame_hop,
             p->num_bins,   p->center,    p->pre_emphasis, p->eps,      p->scale,      pcm16};
  const bool fast = p->fft_size == 512 && p->num_bins == 257 && (p->frame_hop % 2 == 0) &&
                    (p->frame_len % 2 == 0);
  if (fast) {
    const int64_t tiles = (num_frames + kWaveFrames - 1) / kWaveFrames;
    // iterations per wavefront: amortise the per-wave set-up but keep the grid >= ~8 waves / CU.  An ODD
    // count: neighbouring wavefronts then write runs 3 x 8 224 bytes apart -- with 2 or 4 tiles per wave
    // the spacing is 16 448 / 32 896 bytes, just over a power of two, and the waves of a round pile up on
    // the same memory channels (us per launch at 32 / 64 / 128 utterances of 4 channels, 249 frames:
    // 1 tile 35.7 / 57.8 / 102.6, 2 tiles 46.3 / - / 131.5, 3 tiles 36.3 / 56.7 / 97.4, 4 tiles
    // 35.5 / 67.5 / 114.2 -- scripts/gpu_stft_iters.sh)
    int iters = 3;
    if (((tiles + iters - 1) / iters) * num_seq < 2048) iters = 1;
    const char* variant = getenv("APS_STFT_ITERS");  // tuning only
    if (variant && variant[0] >= '1' && variant[0] <= '8') iters = variant[0] - '0';
    const int64_t items = ((tiles + iters - 1) / iters) * num_seq;
    dim3 grid((unsigned)((items + kWavesPerBlock - 1) / kWavesPerBlock));
    const bool pe = p->pre_emphasis > 0.f;
#define APS_LAUNCH_WAVE(PE, POLAR)                                                                               \
  do {                                                                                                           \
    if (pcm16)                                                                                                   \
      hipLaunchKernelGGL((stft512_wave_kernel<PE, POLAR, true>), grid, dim3(256), 0, st, a, iters, tiles);       \
    else                                                                                                         \
      hipLaunchKernelGGL((stft512_wave_kernel<PE, POLAR, false>), grid, dim3(256), 0, st, a, iters, tiles);      \
  } while (0)
    if (pe && p->polar) APS_LAUNCH_WAVE(true, true);
    else if (pe) APS_LAUNCH_WAVE(true, false);
    else if (p->polar) APS_LAUNCH_WAVE(false, true);
    else APS_LAUNCH_WAVE(false, false);
#undef APS_LAUNCH_WAVE
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

extern "C" int aps_stft_forward(const float* wav, int64_t num_seq, int64_t num_samples,
                                const float* window, const aps_stft_params* p, float* out,
                                int64_t stride_seq, int64_t stride_frame, int64_t num_frames,
                                void* stream) {
  return stft_forward_impl(wav, 0, num_seq, num_samples, window, p, out, stride_seq, stride_frame, num_frames,
                           stream);
}

// int16 PCM in, the same transform of sample / 32768 (see StftArgs::pcm16)
extern "C" int aps_stft_forward_pcm16(const int16_t* wav, int64_t num_seq, int64_t num_samples,
                                      const float* window, const aps_stft_params* p, float* out,
                                      int64_t stride_seq, int64_t stride_frame, int64_t num_frames,
                                      void* stream) {
  APS_CHECK_ARG((reinterpret_cast<uintptr_t>(wav) & 3) == 0);
  return stft_forward_impl(reinterpret_cast<const float*>(wav), 1, num_seq, num_samples, window, p, out,
                           stride_seq, stride_frame, num_frames, stream);
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
              p->fft_size, p->frame_len, p->num_bins,  p->polar,   p->scale,     1.0f};
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

// Adjoint of aps_stft_forward (a linear map for polar = 0, pre_emphasis = 0):
//   grad_wav[n] = scale sum_t w[n - tH] sum_{f < F} (gRe[t,f] cos(2 pi f (n - tH) / W) - gIm[t,f] sin(...)).
// For a onesided gradient that is the inverse machinery (Hermitian extension doubles the interior
// bins) run at scale / 2 with bins 0 and W/2 doubled, overlap-added WITHOUT the window^2
// normaliser, and folded back through the reflect padding when center = 1.
extern "C" int aps_stft_backward(const float* grad_store, int64_t num_seq, int64_t num_frames,
                                 int64_t stride_seq, int64_t stride_frame, const float* window,
                                 const aps_stft_params* p, float* grad_wav, int64_t num_samples,
                                 float* workspace, void* stream) {
  APS_CHECK_ARG(grad_store && window && p && grad_wav && workspace);
  APS_CHECK_ARG(num_seq > 0 && num_frames > 0 && num_seq <= 65535 && num_samples > 0);
  APS_CHECK_ARG(p->fft_size >= 2 && p->frame_len >= 1 && p->frame_len <= p->fft_size);
  APS_CHECK_ARG(p->num_bins == p->fft_size || p->num_bins == p->fft_size / 2 + 1);
  APS_CHECK_ARG(num_frames <= aps_stft_num_frames(num_samples, p));
  if (p->polar || p->pre_emphasis > 0.f || p->fft_size > 4096) return APS_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool onesided = p->num_bins != p->fft_size;
  IstftArgs a{grad_store,  window,       workspace,    stride_seq, stride_frame, num_frames,
              p->fft_size, p->frame_len, p->num_bins,  0,          onesided ? 0.5f * p->scale : p->scale,
              2.0f};
  size_t lds = (size_t)p->fft_size * 3 * sizeof(cf);
  if (lds > 48 * 1024)
    hipFuncSetAttribute(reinterpret_cast<const void*>(&istft_frames_kernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(istft_frames_kernel, dim3((unsigned)num_frames, (unsigned)num_seq), dim3(256),
                     lds, st, a, is_pow2(p->fft_size) ? 1 : 0);
  const int64_t pad = p->center ? (p->frame_len / 2) : 0;
  hipLaunchKernelGGL(stft_adjoint_ola_kernel,
                     dim3((unsigned)((num_samples + 255) / 256), (unsigned)num_seq), dim3(256), 0,
                     st, workspace, grad_wav, num_frames, p->frame_len, p->frame_hop, pad,
                     num_samples);
  return aps_launch_status();
}

// Adjoint of aps_stft_inverse (polar = 0): with u = grad_wav / (OLA(w^2) + eps) put back at the
// un-cropped positions, grad_spec[t,k] = m_k scale sum_n w[n] u[tH + n] e^{-2 pi i k n / W}, m_k = 2
// for the interior bins of a onesided spectrum (they entered twice through the Hermitian
// extension), 1 otherwise: the forward machinery at 2 scale with bins 0 and W/2 halved.
// workspace: float [num_seq * ((T-1) H + L)].
extern "C" int aps_stft_inverse_backward(const float* grad_wav, int64_t num_seq,
                                         int64_t num_samples_out, const float* window,
                                         const aps_stft_params* p, float* grad_store,
                                         int64_t stride_seq, int64_t stride_frame,
                                         int64_t num_frames, float* workspace, void* stream) {
  APS_CHECK_ARG(grad_wav && window && p && grad_store && workspace);
  APS_CHECK_ARG(num_seq > 0 && num_frames > 0 && num_seq <= 65535);
  APS_CHECK_ARG(p->num_bins == p->fft_size || p->num_bins == p->fft_size / 2 + 1);
  if (p->polar) return APS_ERR_UNSUPPORTED;
  const int64_t crop = p->center ? (p->frame_len / 2) : 0;
  const int64_t full = (num_frames - 1) * p->frame_hop + p->frame_len;
  APS_CHECK_ARG(num_samples_out == full - 2 * crop && num_samples_out > 0);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(istft_adjoint_normalize_kernel,
                     dim3((unsigned)((full + 255) / 256), (unsigned)num_seq), dim3(256), 0, st,
                     grad_wav, window, workspace, num_frames, p->frame_len, p->frame_hop, crop,
                     num_samples_out, full, p->eps);
  const bool onesided = p->num_bins != p->fft_size;
  aps_stft_params q = *p;
  q.center = 0, q.polar = 0, q.pre_emphasis = 0.f;
  q.scale = onesided ? 2.0f * p->scale : p->scale;
  const int rc = aps_stft_forward(workspace, num_seq, full, window, &q, grad_store, stride_seq,
                                  stride_frame, num_frames, stream);
  if (rc != APS_OK) return rc;
  if (onesided) {
    const int64_t rows = num_seq * num_frames;
    hipLaunchKernelGGL(store_edge_scale_kernel, dim3((unsigned)((2 * rows + 255) / 256)), dim3(256),
                       0, st, grad_store, stride_seq, stride_frame, num_frames, (int)p->num_bins,
                       rows, 0.5f);
  }
  return aps_launch_status();
}

static int stft_features_impl(const float* wav, int pcm16, int64_t N, int64_t C, int64_t num_samples,
                              const float* window, const aps_stft_params* p,
                              const aps_feat_params* q, const int32_t* mel_start,
                              const int32_t* mel_len, const int32_t* mel_off,
                              const float* mel_w, const int32_t* pair_l, const int32_t* pair_r,
                              float* store_out, int64_t stride_seq, int64_t stride_frame,
                              int64_t num_frames, float* feats_out, int32_t* nan_count,
                              void* stream) {
  APS_CHECK_ARG(wav && window && p && q && feats_out);
  APS_CHECK_ARG(N > 0 && C >= 1 && num_samples > 0 && num_frames > 0);
  APS_CHECK_ARG(num_frames <= aps_stft_num_frames(num_samples, p));
  if (p->center) APS_CHECK_ARG(p->frame_len / 2 < num_samples);
  // the fused kernel exists for the 512-point fast path only; callers fall back to the two
  // stand-alone kernels otherwise
  if (!(p->fft_size == 512 && p->num_bins == 257 && p->frame_hop % 2 == 0 &&
        p->frame_len % 2 == 0 && p->frame_len <= 512 && !p->polar))
    return APS_ERR_UNSUPPORTED;
  if (C > 8 || q->num_bins != 257 || q->num_channels != C) return APS_ERR_UNSUPPORTED;
  if (q->num_mels > 0 && (q->num_pairs > 0 || q->num_mels > 256)) return APS_ERR_UNSUPPORTED;
  if (q->power != 1 && q->power != 2) return APS_ERR_INVALID;
  APS_CHECK_ARG(q->ref_channel < C && (q->ref_channel >= 0 || q->num_pairs > 0));
  if (q->num_mels > 0) APS_CHECK_ARG(mel_start && mel_len && mel_off && mel_w);
  if (q->num_pairs > 0) APS_CHECK_ARG(pair_l && pair_r && C >= 2);
  if (store_out) APS_CHECK_ARG(stride_frame >= 2 * 257 && stride_seq >= stride_frame);
  if (p->pre_emphasis > 0.f && (C != 1 || store_out)) return APS_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int D0 = (q->ref_channel >= 0) ? (q->num_mels > 0 ? q->num_mels : 257) : 0;
  FusedArgs fa{};
  fa.st = StftArgs{wav,          window,     store_out,      num_samples,  stride_seq,
                   stride_frame, num_frames, N * C,          p->fft_size,  p->frame_len,
                   p->frame_hop, p->num_bins, p->center,     p->pre_emphasis, p->eps, p->scale, pcm16};
  fa.feats = feats_out;
  fa.mel_start = mel_start; fa.mel_len = mel_len; fa.mel_off = mel_off; fa.mel_w = mel_w;
  fa.pair_l = pair_l; fa.pair_r = pair_r; fa.nan_count = nan_count;
  fa.C = (int)C; fa.D = D0 + q->num_pairs * (q->ipd_sin ? 2 : 1) * 257;
  fa.ref_channel = q->ref_channel; fa.power = q->power; fa.num_mels = q->num_mels;
  fa.apply_log = q->apply_log; fa.norm_mean = q->norm_mean; fa.norm_var = q->norm_var;
  fa.num_pairs = q->num_pairs; fa.ipd_sin = q->ipd_sin;
  fa.log_eps = q->log_eps; fa.log_lower_bound = q->log_lower_bound; fa.cmvn_eps = q->cmvn_eps;
  // the enhancement front end (2 .. 4 channels, log-magnitude + IPD, spectrogram kept): one frame of
  // every channel per wavefront, no barrier (stft512_frame_feat_kernel)
  static const bool frame_form = [] {
    const char* e = getenv("APS_STFT_FRAME_FORM");  // "0": the per-channel-wavefront form (A/B runs)
    return !(e && e[0] == '0');
  }();
  if (frame_form && C >= 2 && C <= kWaveFrames && q->num_mels == 0 && p->pre_emphasis == 0.f) {
    // frames per wavefront: what keeps the grid inside one resident round (3 workgroups of 4 waves per CU)
    int iters = (int)((N * num_frames + 256 * 12 - 1) / (256 * 12));
    if (iters < 1) iters = 1;
    if (iters > 16) iters = 16;
    const char* tune = getenv("APS_STFT_ITERS");  // tuning only
    if (tune && tune[0] >= '1' && tune[0] <= '9') iters = atoi(tune);
    const int64_t items = (num_frames + iters - 1) / iters;
    const int64_t blocks = (N * items + kWavesPerBlock - 1) / kWavesPerBlock;
    if (store_out && pcm16)
      hipLaunchKernelGGL((stft512_frame_feat_kernel<true, true>), dim3((unsigned)blocks), dim3(256), 0, st, fa,
                         iters, items);
    else if (store_out)
      hipLaunchKernelGGL((stft512_frame_feat_kernel<true, false>), dim3((unsigned)blocks), dim3(256), 0, st, fa,
                         iters, items);
    else if (pcm16)
      hipLaunchKernelGGL((stft512_frame_feat_kernel<false, true>), dim3((unsigned)blocks), dim3(256), 0, st, fa,
                         iters, items);
    else
      hipLaunchKernelGGL((stft512_frame_feat_kernel<false, false>), dim3((unsigned)blocks), dim3(256), 0, st, fa,
                         iters, items);
    return aps_launch_status();
  }
  const int64_t tiles = (num_frames + kWaveFrames - 1) / kWaveFrames;
  const int cw = C <= 1 ? 1 : (C <= 2 ? 2 : (C <= 4 ? 4 : 8));
  // tiles per workgroup: the smallest count for which the whole grid is resident in one round
  // (3 workgroups of 4 wavefronts per CU, 256 CUs), so no second partially filled round trails
  const int64_t resident = 256 * (cw <= 4 ? 12 / cw : 2);
  int iters = 1;
  while (iters < 8 && ((tiles + iters - 1) / iters) * N > resident) ++iters;
  const char* tune = getenv("APS_STFT_ITERS");  // tuning only
  if (tune && tune[0] >= '1' && tune[0] <= '8') iters = tune[0] - '0';
  const int64_t blocks = ((tiles + iters - 1) / iters) * N;
  const size_t lds = ((size_t)cw * kWaveFrames * kSlotWords + 256) * sizeof(cf) + 256 * sizeof(float2) +
                     (size_t)kWaveFrames * 260 * sizeof(float);
#define APS_LAUNCH_FUSED(PE, WX, CWV)                                                            \
  do {                                                                                           \
    if (lds > 48 * 1024)                                                                         \
      hipFuncSetAttribute(reinterpret_cast<const void*>(&stft512_feat_kernel<PE, WX, CWV>),      \
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                 \
    hipLaunchKernelGGL((stft512_feat_kernel<PE, WX, CWV>), dim3((unsigned)blocks),               \
                       dim3(64 * CWV), lds, st, fa, iters, tiles);                               \
  } while (0)
  if (cw == 1) {
    if (store_out) APS_LAUNCH_FUSED(false, true, 1);
    else if (p->pre_emphasis > 0.f) APS_LAUNCH_FUSED(true, false, 1);
    else APS_LAUNCH_FUSED(false, false, 1);
  } else {
    if (!store_out) return APS_ERR_UNSUPPORTED;
    if (cw == 2) APS_LAUNCH_FUSED(false, true, 2);
    else if (cw == 4) APS_LAUNCH_FUSED(false, true, 4);
    else APS_LAUNCH_FUSED(false, true, 8);
  }
#undef APS_LAUNCH_FUSED
  return aps_launch_status();
}

extern "C" int aps_stft_features(const float* wav, int64_t N, int64_t C, int64_t num_samples,
                                 const float* window, const aps_stft_params* p,
                                 const aps_feat_params* q, const int32_t* mel_start,
                                 const int32_t* mel_len, const int32_t* mel_off,
                                 const float* mel_w, const int32_t* pair_l, const int32_t* pair_r,
                                 float* store_out, int64_t stride_seq, int64_t stride_frame,
                                 int64_t num_frames, float* feats_out, int32_t* nan_count,
                                 void* stream) {
  return stft_features_impl(wav, 0, N, C, num_samples, window, p, q, mel_start, mel_len, mel_off, mel_w, pair_l,
                            pair_r, store_out, stride_seq, stride_frame, num_frames, feats_out, nan_count, stream);
}

// int16 PCM in, the same chain on sample / 32768 (see StftArgs::pcm16)
extern "C" int aps_stft_features_pcm16(const int16_t* wav, int64_t N, int64_t C, int64_t num_samples,
                                       const float* window, const aps_stft_params* p,
                                       const aps_feat_params* q, const int32_t* mel_start,
                                       const int32_t* mel_len, const int32_t* mel_off,
                                       const float* mel_w, const int32_t* pair_l, const int32_t* pair_r,
                                       float* store_out, int64_t stride_seq, int64_t stride_frame,
                                       int64_t num_frames, float* feats_out, int32_t* nan_count,
                                       void* stream) {
  APS_CHECK_ARG((reinterpret_cast<uintptr_t>(wav) & 3) == 0);
  return stft_features_impl(reinterpret_cast<const float*>(wav), 1, N, C, num_samples, window, p, q, mel_start,
                            mel_len, mel_off, mel_w, pair_l, pair_r, store_out, stride_seq, stride_frame,
                            num_frames, feats_out, nan_count, stream);
}
