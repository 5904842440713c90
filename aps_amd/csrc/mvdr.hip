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
#include <stdlib.h>

#include "common.h"

namespace aps {

// ------------------------------------------------------------------------------------------
// covariance
//
// grid (F / 32, N, TS): a workgroup owns 32 bins x one segment of the frame axis; its 256 threads
// are 32 bins x 8 frame phases.  Every thread keeps the upper triangle of sum_t m x x^H for both
// masks in registers (C(C+1) floats each), so the spectrogram is read exactly once, 256
// contiguous bytes per half-wave.  Phases are folded by one shuffle + one LDS pass; the TS
// segment partials go to a small scratch ([N, TS, NV, F], F fastest) that a second tiny kernel
// folds, normalises and expands to the Hermitian N x F x C x C x 2 output.
//
// _process_mask (mvdr.py:103-116) is folded in: padded frames are zeroed on the fly; the
// max-normalisation m / (max_t|m| + EPS) is a per-(n, f) constant d, applied after the
// accumulation in the common case ((sum m x x^H) / d, clamp((sum m) / d)): same value, one
// division per output instead of one per element, and no second pass over the masks.  When the
// noise mask is implicit (1 - processed speech mask) or the processed masks are requested, d is
// needed up front and comes from a small max pre-pass (exact reference order of operations).
// ------------------------------------------------------------------------------------------
struct CovArgs {
  const float* store;
  const float* mask_s;
  const float* mask_n;
  const int64_t* x_len;
  const float* pre_div;  // [N, 2, F] max+EPS from the pre-pass, or NULL (post-hoc division)
  float* partial;        // [N, TS, NV, F]
  float* pmask_s;
  float* pmask_n;
  int64_t T, F;
  int64_t stride_n, stride_c, stride_t;
  int32_t mask_norm;
  int32_t seg_len;  // frames per segment
  int64_t mask_ld;  // floats between two frames of a mask (F: dense; 2 F: one half of a [N, T, 2 F] estimate)
  int32_t gx, gy, gz;  // the logical grid (bin blocks, utterances, frame segments) behind the 1-D launch
};

#ifndef APS_COV_XCD_SWIZZLE
#define APS_COV_XCD_SWIZZLE 1  // 0: logical block = launch order (A/B builds)
#endif

constexpr int kCovMaxSegments = 8;
constexpr int kCovBins = 32;   // bins per workgroup
constexpr int kCovPhases = 8;  // frame phases per workgroup (256 threads)

template <int C>
struct CovLayout {
  static constexpr int NU = C * (C + 1);    // floats of one packed upper triangle (re, im)
  static constexpr int NV = 2 * NU + 4;     // speech, noise, sum_s, sum_n, max_s, max_n
  __host__ __device__ static constexpr int upper(int i, int j) {  // i <= j
    return (i * C - i * (i - 1) / 2 + (j - i)) * 2;
  }
};

__global__ __launch_bounds__(256) void mask_max_kernel(const float* __restrict__ mask_s,
                                                       const float* __restrict__ mask_n,
                                                       const int64_t* __restrict__ x_len,
                                                       int64_t T, int64_t F, int64_t ld,
                                                       float* __restrict__ pre_div) {
  __shared__ float s_max[kCovPhases][2][kCovBins];
  const int fl = threadIdx.x & 31, tp = threadIdx.x >> 5;
  const int64_t n = blockIdx.y, f = (int64_t)blockIdx.x * kCovBins + fl;
  int64_t len = T;
  if (x_len) len = max((int64_t)0, min(T, x_len[n]));
  float mx_s = 0.f, mx_n = 0.f;
  if (f < F) {
    for (int64_t t = tp; t < len; t += kCovPhases) {
      mx_s = fmaxf(mx_s, fabsf(mask_s[(n * T + t) * ld + f]));
      if (mask_n) mx_n = fmaxf(mx_n, fabsf(mask_n[(n * T + t) * ld + f]));
    }
  }
  s_max[tp][0][fl] = mx_s;
  s_max[tp][1][fl] = mx_n;
  __syncthreads();
  if (tp == 0 && f < F) {
#pragma unroll
    for (int q = 1; q < kCovPhases; ++q) {
      mx_s = fmaxf(mx_s, s_max[q][0][fl]);
      mx_n = fmaxf(mx_n, s_max[q][1][fl]);
    }
    pre_div[(n * 2 + 0) * F + f] = mx_s + APS_EPSILON;
    pre_div[(n * 2 + 1) * F + f] = mx_n + APS_EPSILON;
  }
}

// MvdrBeamformer._process_mask on its own (mvdr.py:103-116; `forward` folds it into the covariance
// pass): frames past x_len zeroed, m / (max_t |m| + EPS) per (n, f) when mask_norm, transposed to
// N x F x T.  A workgroup owns 32 bins of one utterance: maxima as in mask_max_kernel, then the rows.
__global__ __launch_bounds__(256) void process_mask_kernel(const float* __restrict__ mask,
                                                           const int64_t* __restrict__ x_len, int64_t T,
                                                           int64_t F, int mask_norm, int complement,
                                                           float* __restrict__ out) {
  __shared__ float s_max[kCovPhases][kCovBins];
  const int fl = threadIdx.x & 31, tp = threadIdx.x >> 5;
  const int64_t n = blockIdx.y, f = (int64_t)blockIdx.x * kCovBins + fl;
  int64_t len = T;
  if (x_len) len = max((int64_t)0, min(T, x_len[n]));
  float mx = 0.f;
  if (f < F && mask_norm)
    for (int64_t t = tp; t < len; t += kCovPhases) mx = fmaxf(mx, fabsf(mask[(n * T + t) * F + f]));
  s_max[tp][fl] = mx;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < kCovPhases; ++q) mx = fmaxf(mx, s_max[q][fl]);
  if (f >= F) return;
  const float d = mx + APS_EPSILON;
  for (int64_t t = tp; t < T; t += kCovPhases) {
    float m = t < len ? mask[(n * T + t) * F + f] : 0.f;
    if (mask_norm) m = m / d;
    if (complement)
      out[(n * T + t) * F + f] = 1.0f - m;  // (the implicit noise mask, in the mask's own layout)
    else
      out[(n * F + f) * T + t] = m;
  }
}

template <int C, int BINS>
__global__ __launch_bounds__(256) void covariance_partial_kernel(CovArgs a) {
  constexpr int PH = 256 / BINS;  // frame phases per workgroup
  using Lay = CovLayout<C>;
  constexpr int NU = Lay::NU, NV = Lay::NV;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_red = reinterpret_cast<float*>(smem);  // [4 waves][NV][BINS]

  const int tid = threadIdx.x;
  const int fl = tid % BINS;
  const int tp = tid / BINS;
  // The launch is 1-D and padded to a multiple of 8: workgroup L is dispatched to XCD L % 8, and logical blocks that
  // follow each other (the bin blocks of one (utterance, segment): rows of 257 bins are 2 056 bytes, so neighbouring
  // bin blocks share the 128-byte lines at their seams, and the masks' 256-byte runs straddle three lines) are given to
  // ONE XCD -- v = (L % 8) * (grid / 8) + L / 8 -- whose L2 then fetches a seam line once instead of once per XCD.
#if APS_COV_XCD_SWIZZLE
  const int v_blk = (int)(blockIdx.x % 8) * (int)(gridDim.x / 8) + (int)(blockIdx.x / 8);
#else
  const int v_blk = (int)blockIdx.x;
#endif
  if (v_blk >= a.gx * a.gy * a.gz) return;
  const int bx = v_blk % a.gx, by = (v_blk / a.gx) % a.gy, bz = v_blk / (a.gx * a.gy);
  const int64_t n = by;
  const int64_t f = (int64_t)bx * BINS + fl;
  const bool valid = f < a.F;
  const int64_t T = a.T, F = a.F;
  int64_t len = T;
  if (a.x_len) len = max((int64_t)0, min(T, a.x_len[n]));
  const int64_t t_beg = (int64_t)bz * a.seg_len;
  const int64_t t_end = min(T, t_beg + a.seg_len);
  const int64_t ML = a.mask_ld;
  const float* ms_p = a.mask_s + n * T * ML + f;
  const float* mn_p = a.mask_n ? a.mask_n + n * T * ML + f : nullptr;
  const bool pre = a.pre_div != nullptr;
  float div_s = 1.f, div_n = 1.f;
  if (pre && a.mask_norm && valid) {
    div_s = a.pre_div[(n * 2 + 0) * F + f];
    div_n = a.pre_div[(n * 2 + 1) * F + f];
  }

  float acc_s[NU], acc_n[NU];
#pragma unroll
  for (int v = 0; v < NU; ++v) acc_s[v] = acc_n[v] = 0.f;
  float sum_s = 0.f, sum_n = 0.f, mx_s = 0.f, mx_n = 0.f;
  if (valid) {
    const float* xb = a.store + n * a.stride_n + 2 * f;
#pragma unroll 2
    for (int64_t t = t_beg + tp; t < t_end; t += PH) {
      float ms = (t < len) ? ms_p[t * ML] : 0.f;
      float mn = (mn_p && t < len) ? mn_p[t * ML] : 0.f;
      cf x[C];
#pragma unroll
      for (int c = 0; c < C; ++c) x[c] = ld_cf(xb + c * a.stride_c + t * a.stride_t);
      mx_s = fmaxf(mx_s, fabsf(ms));
      mx_n = fmaxf(mx_n, fabsf(mn));
      if (pre) {
        if (a.mask_norm) {
          ms = ms / div_s;
          mn = mn / div_n;
        }
        if (!mn_p) mn = 1.0f - ms;  // mvdr.py:136
        if (a.pmask_s) a.pmask_s[(n * F + f) * T + t] = ms;
        if (a.pmask_n) a.pmask_n[(n * F + f) * T + t] = mn;
      }
      sum_s += ms;
      sum_n += mn;
#pragma unroll
      for (int i = 0; i < C; ++i) {
#pragma unroll
        for (int j = i; j < C; ++j) {
          const int v = Lay::upper(i, j);
          const float pr = x[i].re * x[j].re + x[i].im * x[j].im;
          acc_s[v] += ms * pr;
          acc_n[v] += mn * pr;
          if (i != j) {  // the diagonal of x x^H is real: keep it exactly so
            const float pi = x[i].im * x[j].re - x[i].re * x[j].im;
            acc_s[v + 1] += ms * pi;
            acc_n[v + 1] += mn * pi;
          }
        }
      }
    }
  }

  // fold the 8 frame phases: pairs inside a wave by shuffle, the 4 waves through LDS
  const int wv = tid >> 6;
  auto put = [&](int v, float val, bool is_max) {
    if (BINS == 32) {  // two phases share a wavefront
      const float o = __shfl_xor(val, 32, 64);
      val = is_max ? fmaxf(val, o) : val + o;
    }
    if (BINS == 64 || (tid & 32) == 0) s_red[(wv * NV + v) * BINS + fl] = val;
  };
#pragma unroll
  for (int v = 0; v < NU; ++v) {
    put(v, acc_s[v], false);
    put(NU + v, acc_n[v], false);
  }
  put(2 * NU + 0, sum_s, false);
  put(2 * NU + 1, sum_n, false);
  put(2 * NU + 2, mx_s, true);
  put(2 * NU + 3, mx_n, true);
  __syncthreads();
  if (!valid) return;
  float* out = a.partial + ((n * a.gz + bz) * NV) * F + f;
  for (int v = tp; v < NV; v += PH) {
    const float* r = s_red + v * BINS + fl;
    const float p0 = r[0 * NV * BINS], p1 = r[1 * NV * BINS], p2 = r[2 * NV * BINS],
                p3 = r[3 * NV * BINS];
    out[(int64_t)v * F] = (v >= 2 * NU + 2) ? fmaxf(fmaxf(p0, p1), fmaxf(p2, p3))
                                            : (p0 + p1) + (p2 + p3);
  }
}

// One thread per (n, f): every segment partial of the bin (NV x TS values, each a contiguous run
// of bins across the wave) is loaded in ONE batch, then folded, normalised and expanded to the
// Hermitian N x F x C x C x 2 outputs.  The kernel also emits v[n, c, f] = |mean_{j != c} Rs[c, j]|
// (mvdr.py:165-170), the only thing ChannelAttention needs from Rs, so the attention kernel does
// not re-read the covariance.
// the folded values v[NV] of one (n, f) -> normalised, expanded outputs (the body of the fold kernels)
template <int C>
__device__ __forceinline__ void covariance_outputs(const float (&v)[2 * C * (C + 1) + 4], int64_t n, int64_t f,
                                                   int64_t F, int mask_norm, int pre_divided,
                                                   float* __restrict__ cov_s, float* __restrict__ cov_n,
                                                   float* __restrict__ offdiag, float* __restrict__ packed) {
  using Lay = CovLayout<C>;
  constexpr int NU = Lay::NU;
  const int64_t idx = n * F + f;
  // post-hoc mask normalisation: m' = m / d  =>  sum m' x x^H = (sum m x x^H) / d
  const bool post = mask_norm && !pre_divided;
  const float d_s = post ? v[2 * NU + 2] + APS_EPSILON : 1.f;
  const float d_n = post ? v[2 * NU + 3] + APS_EPSILON : 1.f;
  const float den_s = fmaxf(v[2 * NU + 0] / d_s, APS_EPSILON);  // clamp(min=EPSILON), mvdr.py:59
  const float den_n = fmaxf(v[2 * NU + 1] / d_n, APS_EPSILON);
  float* os = cov_s ? cov_s + idx * (C * C * 2) : nullptr;
  float* on = cov_n ? cov_n + idx * (C * C * 2) : nullptr;
  float* pk = packed ? packed + (n * 2 * NU) * F + f : nullptr;  // [N][2 NU][F], coalesced
  float ore[C], oim[C];  // off-diagonal row sums of Rs
#pragma unroll
  for (int c = 0; c < C; ++c) ore[c] = oim[c] = 0.f;
#pragma unroll
  for (int i = 0; i < C; ++i)
#pragma unroll
    for (int j = i; j < C; ++j) {
      const int u = Lay::upper(i, j);
      const float sr = v[u] / d_s / den_s, si = (i == j) ? 0.f : v[u + 1] / d_s / den_s;
      const float nr = v[NU + u] / d_n / den_n, ni = (i == j) ? 0.f : v[NU + u + 1] / d_n / den_n;
      if (pk) {
        pk[(int64_t)(u + 0) * F] = sr;
        pk[(int64_t)(u + 1) * F] = si;
        pk[(int64_t)(NU + u + 0) * F] = nr;
        pk[(int64_t)(NU + u + 1) * F] = ni;
      }
      if (os) {
        st_cf(os + (i * C + j) * 2, {sr, si});
        st_cf(on + (i * C + j) * 2, {nr, ni});
      }
      if (i != j) {
        if (os) {
          st_cf(os + (j * C + i) * 2, {sr, -si});
          st_cf(on + (j * C + i) * 2, {nr, -ni});
        }
        ore[i] += sr;
        oim[i] += si;
        ore[j] += sr;
        oim[j] -= si;
      }
    }
  if (offdiag) {
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float re = ore[c] / (float)(C - 1), im = oim[c] / (float)(C - 1);
      offdiag[(n * C + c) * F + f] = sqrtf(re * re + im * im);
    }
  }
}

template <int C, int TSMAX>
__global__ __launch_bounds__(64) void covariance_finalize_kernel(const float* __restrict__ partial,
                                                                 int64_t NF, int64_t F, int TS,
                                                                 int mask_norm, int pre_divided,
                                                                 float* __restrict__ cov_s,
                                                                 float* __restrict__ cov_n,
                                                                 float* __restrict__ offdiag,
                                                                 float* __restrict__ packed) {
  using Lay = CovLayout<C>;
  constexpr int NU = Lay::NU, NV = Lay::NV;
  const int64_t idx = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (idx >= NF) return;
  const int64_t n = idx / F, f = idx % F;
  const float* p0 = partial + (n * TS * NV) * F + f;
  float x[TSMAX][NV];
#pragma unroll
  for (int ts = 0; ts < TSMAX; ++ts)
#pragma unroll
    for (int q = 0; q < NV; ++q) x[ts][q] = (ts < TS) ? p0[((int64_t)ts * NV + q) * F] : 0.f;
  float v[NV];
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    float r = x[0][q];
#pragma unroll
    for (int ts = 1; ts < TSMAX; ++ts) r = (q >= 2 * NU + 2) ? fmaxf(r, x[ts][q]) : r + x[ts][q];
    v[q] = r;
  }
  covariance_outputs<C>(v, n, f, F, mask_norm, pre_divided, cov_s, cov_n, offdiag, packed);
}

// ------------------------------------------------------------------------------------------
// channel attention
//
// score[n, c] = sum_a g[a] tanh(b[a] + sum_f P[a, f] v[n, c, f]),  v = |off-diagonal mean of Rs|.
// grid (A / rows_per_block, N); a wavefront owns ROWS = 64 / C rows of P.  Lanes run along f, so a
// row of P is read as contiguous 256-byte runs, and each lane keeps ROWS x C partial dot products
// in 64 registers: ROWS independent loads per 64-bin chunk are in flight and nothing is reduced
// inside the loop.  The 64 partials are then reduced across the 64 lanes with one halving
// butterfly (63 shuffles, lane L ends up owning value L = (row, channel)), followed by tanh / gvec
// and a tiny LDS fold over rows and wavefronts.  Softmax over channels is the finalise kernel.
// ------------------------------------------------------------------------------------------
// V_GIVEN: `src` is v[n, c, f] (emitted by the covariance fold); otherwise `src` is Rs and v is
// derived here.
template <int C, bool V_GIVEN>
__global__ __launch_bounds__(256) void attention_partial_kernel(
    const float* __restrict__ src, int64_t F, int64_t A, const float* __restrict__ proj_w,
    const float* __restrict__ proj_b, const float* __restrict__ gvec_w,
    float* __restrict__ scratch) {
  constexpr int ROWS = 64 / C;  // rows of P per wavefront; ROWS * C <= 64 partial sums per lane
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_v = reinterpret_cast<float*>(smem);  // [C][F]
  __shared__ float s_h[4][64];
  const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
  const int64_t n = blockIdx.y;
  const int nchunk = gridDim.x;
  // first 64-bin chunk of this wavefront's rows of P: issued before the Rs pass so its
  // latency overlaps it (P does not depend on Rs)
  const int64_t a0 = ((int64_t)blockIdx.x * 4 + wv) * ROWS;
  float pw[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r)
    pw[r] = (ln < F && a0 + r < A) ? proj_w[(a0 + r) * F + ln] : 0.f;
  if (V_GIVEN) {
    const float* vg = src + n * C * F;
    const int64_t total = F * C;
    float tmp[8];  // 2048 values with every load in flight, rolled beyond
#pragma unroll
    for (int k = 0; k < 8; ++k) tmp[k] = (tid + 256 * k < total) ? vg[tid + 256 * k] : 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (tid + 256 * k < total) s_v[tid + 256 * k] = tmp[k];
    for (int64_t idx = tid + 2048; idx < total; idx += 256) s_v[idx] = vg[idx];
  }
  // |sum_{j != c} Rs[f, c, j]| / (C - 1)     (mvdr.py:165-170)
  const float* rs = src + n * F * (C * C * 2);
  for (int64_t idx = tid; !V_GIVEN && idx < F * C; idx += 256) {
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

  float acc[64];
#pragma unroll
  for (int v = 0; v < 64; ++v) acc[v] = 0.f;
  for (int64_t f = ln; f < F; f += 64) {
    // prefetch the next 64-bin chunk of the ROWS rows before consuming the current one
    float pn[ROWS];
    const int64_t fn = f + 64;
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
      pn[r] = (fn < F && a0 + r < A) ? proj_w[(a0 + r) * F + fn] : 0.f;
    float vv[C];
#pragma unroll
    for (int c = 0; c < C; ++c) vv[c] = s_v[c * F + f];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
#pragma unroll
      for (int c = 0; c < C; ++c) acc[r * C + c] += pw[r] * vv[c];
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) pw[r] = pn[r];
  }
  // halving butterfly: after the step with offset o a lane keeps the half of its values selected
  // by its bit o; lane L finishes with the full sum of value index L
#pragma unroll
  for (int o = 32, cnt = 32; o >= 1; o >>= 1, cnt >>= 1) {
    const bool up = (ln & o) != 0;
#pragma unroll
    for (int k = 0; k < cnt; ++k) {
      const float keep = up ? acc[k + cnt] : acc[k];
      const float send = up ? acc[k] : acc[k + cnt];
      acc[k] = keep + __shfl_xor(send, o, 64);
    }
  }
  float h = 0.f;
  if (ln < ROWS * C) {
    const int64_t aa = a0 + ln / C;
    if (aa < A) h = gvec_w[aa] * tanhf(acc[0] + proj_b[aa]);
  }
  s_h[wv][ln] = h;
  __syncthreads();
  if (tid < C) {
    float part = 0.f;
    for (int w = 0; w < 4; ++w)
      for (int r = 0; r < ROWS; ++r) part += s_h[w][r * C + tid];
    scratch[(n * C + tid) * nchunk + blockIdx.x] = part;
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
// (the operand scaled by its larger part first: |a|^2 leaves the fp32 range for |a| > 1.8e19 or < 1e-19)
__device__ __forceinline__ cf crecip(cf a) {
  const float big = fmaxf(fabsf(a.re), fabsf(a.im));
  const float r = 1.0f / big, x = a.re * r, y = a.im * r, k = r / (x * x + y * y);
  return {x * k, -y * k};
}

// w = (Rn + eps I)^-1 Rs u / (tr((Rn + eps I)^-1 Rs) + eps)      (mvdr.py:75-101, cplx.py:221-278)
// A = Rn, B = Rs (both destroyed).  Complex Gaussian elimination with partial pivoting, all in
// registers (every index is a compile-time constant after unrolling).
// Returns false when a pivot of Rn + eps I is zero or not finite -- where the reference's Rn.inverse()
// (th.inverse of the real embedding) raises; the outputs are inf / NaN then.
template <int C>
__device__ __forceinline__ bool mvdr_weight_of(cf (&A)[C][C], cf (&B)[C][C], const float (&uu)[C],
                                               float eps, cf (&w)[C]) {
  bool ok = true;
#pragma unroll
  for (int i = 0; i < C; ++i) A[i][i].re += eps;  // Rn + eps I   (mvdr.py:89-90)
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
    // (the pivot itself zero or not finite -- not its square, which over- / underflows long before the pivot does)
    const float big = fmaxf(fabsf(A[k][k].re), fabsf(A[k][k].im));
    ok = ok && big > 0.f && big <= 3.4028234e38f;
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
#pragma unroll
  for (int k = C - 1; k >= 0; --k) {  // back substitution: Y overwrites B
#pragma unroll
    for (int j = 0; j < C; ++j) {
      cf acc = B[k][j];
#pragma unroll
      for (int m = k + 1; m < C; ++m) acc = acc - cmul(A[k][m], B[m][j]);
      B[k][j] = cmul(acc, A[k][k]);
    }
  }
  cf tr = {eps, 0.f};  // trace(Y) + eps
#pragma unroll
  for (int k = 0; k < C; ++k) tr = tr + B[k][k];
  const float scale = tr.re * tr.re + tr.im * tr.im;
#pragma unroll
  for (int i = 0; i < C; ++i) {
    cf v = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < C; ++j) v = v + cscale(B[i][j], uu[j]);
    w[i] = {(v.re * tr.re + v.im * tr.im) / scale, (v.im * tr.re - v.re * tr.im) / scale};
  }
  return ok;
}

// FROM_SCORES: u is not given; it is softmax_c(gvec_b + sum_chunks score[n, c, chunk]) from the
// attention partials (the finalise step of ChannelAttention folded into this kernel).
// PACKED: cov_s points at the fold kernel's packed upper triangles [N][2 NU][F] (coalesced reads).
template <int C, bool FROM_SCORES, bool PACKED>
__global__ __launch_bounds__(64) void weight_kernel(const float* __restrict__ cov_s,
                                                    const float* __restrict__ cov_n,
                                                    const float* __restrict__ u, int64_t NF,
                                                    int64_t F, float eps,
                                                    float* __restrict__ weight,
                                                    const float* __restrict__ scores, int nchunk,
                                                    const float* __restrict__ gvec_b,
                                                    float* __restrict__ u_out, int32_t* singular) {
  const int64_t idx = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (idx >= NF) return;
  const int64_t n = idx / F;
  float uu[C];
  if (FROM_SCORES) {
    float mx = -INFINITY;
    constexpr int QMAX = 16;  // chunks folded with all loads in flight (rolled tail beyond)
    float sc[C][QMAX];
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int q = 0; q < QMAX; ++q)
        sc[c][q] = (q < nchunk) ? scores[(n * C + c) * nchunk + q] : 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      float v = gvec_b[0];
#pragma unroll
      for (int q = 0; q < QMAX; ++q) v += sc[c][q];
      for (int q = QMAX; q < nchunk; ++q) v += scores[(n * C + c) * nchunk + q];
      uu[c] = v;
      mx = fmaxf(mx, v);
    }
    float den = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      uu[c] = expf(uu[c] - mx);
      den += uu[c];
    }
#pragma unroll
    for (int c = 0; c < C; ++c) uu[c] = uu[c] / den;
    if (idx % F == 0) {
#pragma unroll
      for (int c = 0; c < C; ++c) u_out[n * C + c] = uu[c];
    }
  } else {
#pragma unroll
    for (int c = 0; c < C; ++c) uu[c] = u[n * C + c];
  }
  cf A[C][C], B[C][C];
  if (PACKED) {
    using Lay = CovLayout<C>;
    constexpr int NU = Lay::NU;
    const float* pk = cov_s + (n * 2 * NU) * F + (idx - n * F);
    float raw[2 * NU];
#pragma unroll
    for (int q = 0; q < 2 * NU; ++q) raw[q] = pk[(int64_t)q * F];
#pragma unroll
    for (int i = 0; i < C; ++i)
#pragma unroll
      for (int j = i; j < C; ++j) {
        const int u = Lay::upper(i, j);
        B[i][j] = {raw[u], raw[u + 1]};
        A[i][j] = {raw[NU + u], raw[NU + u + 1]};
        if (i != j) {
          B[j][i] = cconj(B[i][j]);
          A[j][i] = cconj(A[i][j]);
        }
      }
  } else {
    const float* pn = cov_n + idx * (C * C * 2);
    const float* ps = cov_s + idx * (C * C * 2);
#pragma unroll
    for (int i = 0; i < C; ++i)
#pragma unroll
      for (int j = 0; j < C; ++j) {
        A[i][j] = ld_cf(pn + (i * C + j) * 2);
        B[i][j] = ld_cf(ps + (i * C + j) * 2);
      }
  }
  cf w[C];
  if (!mvdr_weight_of<C>(A, B, uu, eps, w) && singular) atomicAdd(singular, 1);
#pragma unroll
  for (int i = 0; i < C; ++i) st_cf(weight + (idx * C + i) * 2, w[i]);
}

// ------------------------------------------------------------------------------------------
// beamform
// ------------------------------------------------------------------------------------------

// One wavefront per frame row; lanes along bins.  NITER > 0: the row fits 64 * NITER bins and all
// C * NITER spectrogram loads plus the C * NITER weight loads are issued before the first use
// (a rolled loop over the 64-bin chunks would expose one memory round trip per chunk).
template <int C, int NITER>
__global__ __launch_bounds__(256) void beamform_kernel(const float* __restrict__ store,
                                                       const float* __restrict__ weight, int64_t T,
                                                       int64_t F, int64_t stride_n,
                                                       int64_t stride_c, int64_t stride_t,
                                                       float* __restrict__ y, int fpw) {
  const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
  const int64_t n = blockIdx.y;
  const int64_t t0 = ((int64_t)blockIdx.x * 4 + wv) * fpw;
  if (t0 >= T) return;
  const int64_t t1 = (t0 + fpw < T) ? t0 + fpw : T;
  if (NITER > 0) {
    constexpr int NI = NITER > 0 ? NITER : 1;
    cf w[NI][C];
#pragma unroll
    for (int i = 0; i < NITER; ++i) {
      const int64_t f = ln + 64 * i;
#pragma unroll
      for (int c = 0; c < C; ++c)
        w[i][c] = (f < F) ? ld_cf(weight + ((n * F + f) * C + c) * 2) : cf{0.f, 0.f};
    }
    for (int64_t t = t0; t < t1; ++t) {
      const float* xb = store + n * stride_n + t * stride_t;
      cf x[NI][C];
#pragma unroll
      for (int i = 0; i < NITER; ++i) {
        const int64_t f = ln + 64 * i;
#pragma unroll
        for (int c = 0; c < C; ++c)
          x[i][c] = (f < F) ? ld_cf(xb + c * stride_c + 2 * f) : cf{0.f, 0.f};
      }
#pragma unroll
      for (int i = 0; i < NITER; ++i) {
        const int64_t f = ln + 64 * i;
        cf acc = {0.f, 0.f};
#pragma unroll
        for (int c = 0; c < C; ++c) {
          acc.re += w[i][c].re * x[i][c].re + w[i][c].im * x[i][c].im;  // conj(w) * x
          acc.im += w[i][c].re * x[i][c].im - w[i][c].im * x[i][c].re;
        }
        if (f < F) st_cf(y + ((n * T + t) * F + f) * 2, acc);
      }
    }
    return;
  }
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

// ------------------------------------------------------------------------------------------
// SURVEY 8(d) P3 in one pass (round 4): beamform + AbsTransform + [mel] [log] [row CMVN] -- what EnhASRBase
// does with the beamformer's output (enh_att.py:86-93: x_enh = asr_transform(enh_net(...)); the complex beam
// output itself is never returned).  The loop of beamform_kernel<C, 0>, with |(Re y + eps) + i Im y| of the
// wave's frames parked in wave-private LDS instead of (or, when y is given, beside) the 2 x T x F x 4-byte
// row of Y; then, per frame, the tail of features_kernel<1> (feats.hip) on the parked row: banded mel
// product, log, the row's mean / variance as two wave reductions, one store of D floats.  Y is neither
// written (16.4 MB per 32 utterances) nor read back (another 16.4 MB); one launch instead of two.
// ------------------------------------------------------------------------------------------
struct BeamFeatArgs {
  const float* store;
  const float* weight;
  float* y;    // [N, T, F, 2] or null
  float* out;  // [N, T, D]
  const int32_t* mel_start;
  const int32_t* mel_len;
  const int32_t* mel_off;
  const float* mel_w;
  int32_t* nan_count;
  int64_t T, F, stride_n, stride_c, stride_t;
  int32_t fpw, power, num_mels, apply_log, norm_mean, norm_var, D;
  float abs_eps, log_eps, log_lower_bound, cmvn_eps;
};

constexpr int kBeamMelCap = 1024;  // floats of banded mel weights kept in LDS (80 mels over 257 bins: ~514)

template <int C>
__global__ __launch_bounds__(256) void beamform_features_kernel(BeamFeatArgs a) {
  extern __shared__ __attribute__((aligned(16))) float s_bf[];
  const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
  const int64_t n = blockIdx.y, T = a.T, F = a.F;
  const int D = a.D, per_wave = a.fpw * (int)F + (D > (int)F ? D : (int)F);
  // the banded mel weights (2 F floats for triangular filters) once per workgroup in LDS: read from global inside
  // the band loop they were a dependent L1 round trip per tap and frame (round 6)
  float* s_melw = s_bf + (size_t)4 * per_wave;  // [kBeamMelCap]
  const int mel_total = a.num_mels > 0 ? a.mel_off[D - 1] + a.mel_len[D - 1] : 0;
  const bool mel_lds = mel_total > 0 && mel_total <= kBeamMelCap;
  if (mel_lds)
    for (int i = threadIdx.x; i < mel_total; i += 256) s_melw[i] = a.mel_w[i];
  __syncthreads();   // (the only workgroup barrier; below a wave's LDS is its own)
  const int64_t t0 = ((int64_t)blockIdx.x * 4 + wv) * a.fpw;
  if (t0 >= T) return;
  const int64_t t1 = (t0 + a.fpw < T) ? t0 + a.fpw : T;
  float* s_mag = s_bf + (size_t)wv * per_wave;  // [fpw][F]
  float* s_val = s_mag + (size_t)a.fpw * F;     // [max(D, F)]
  for (int64_t f = ln; f < F; f += 64) {
    cf w[C];
#pragma unroll
    for (int c = 0; c < C; ++c) w[c] = ld_cf(a.weight + ((n * F + f) * C + c) * 2);
    for (int64_t t = t0; t < t1; ++t) {
      const float* xb = a.store + n * a.stride_n + t * a.stride_t + 2 * f;
      cf acc = {0.f, 0.f};
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const cf x = ld_cf(xb + c * a.stride_c);
        acc.re += w[c].re * x.re + w[c].im * x.im;  // conj(w) * x
        acc.im += w[c].re * x.im - w[c].im * x.re;
      }
      if (a.y) st_cf(a.y + ((n * T + t) * F + f) * 2, acc);
      const float re = acc.re + a.abs_eps;  // AbsTransform on a complex input: eps joins the real part
      float v = sqrtf(re * re + acc.im * acc.im);
      if (a.power == 2) v = v * v;
      s_mag[(t - t0) * F + f] = v;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  bool bad = false;
  for (int64_t t = t0; t < t1; ++t) {
    const float* mag = s_mag + (t - t0) * F;
    float* orow = a.out + (n * T + t) * (int64_t)D;
    float part = 0.f;
    for (int d = ln; d < D; d += 64) {
      float v;
      if (a.num_mels > 0) {
        const int st = a.mel_start[d], len = a.mel_len[d];
        const float* mw = (mel_lds ? s_melw : a.mel_w) + a.mel_off[d];
        v = 0.f;
        for (int q = 0; q < len; ++q) v += mw[q] * mag[st + q];
      } else {
        v = mag[d];
      }
      if (a.apply_log)  // clamp(min=eps) must propagate NaN like th.clamp does
        v = (a.log_lower_bound > 0.f) ? logf(a.log_lower_bound + v) : logf(v < a.log_eps ? a.log_eps : v);
      s_val[d] = v;
      part += v;
    }
    if (a.norm_mean || a.norm_var) {  // CmvnTransform per band = over the row (asr.py:576-585)
      const float mean = wave_sum(part) / (float)D;
      float sq = 0.f;
      for (int d = ln; d < D; d += 64) {  // (a lane re-reads the values it wrote itself)
        const float c = s_val[d] - mean;
        sq += c * c;
        if (a.norm_mean) s_val[d] = c;
      }
      const float var = wave_sum(sq) / (float)D;
      for (int d = ln; d < D; d += 64) {
        const float o = a.norm_var ? s_val[d] / sqrtf(var + a.cmvn_eps) : s_val[d];
        bad |= (o != o);
        orow[d] = o;
      }
    } else {
      for (int d = ln; d < D; d += 64) {
        bad |= (s_val[d] != s_val[d]);
        orow[d] = s_val[d];
      }
    }
  }
  if (a.nan_count != nullptr && __any(bad) && ln == 0) atomicAdd(a.nan_count, 1);
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

extern "C" int aps_mvdr_process_mask(const float* mask, const int64_t* x_len, int64_t N, int64_t T,
                                     int64_t F, int32_t mask_norm, int32_t complement, float* out,
                                     void* stream) {
  APS_CHECK_ARG(mask && out && N > 0 && N <= 65535 && T > 0 && F > 0);
  hipLaunchKernelGGL(process_mask_kernel, dim3((unsigned)((F + kCovBins - 1) / kCovBins), (unsigned)N),
                     dim3(256), 0, static_cast<hipStream_t>(stream), mask, x_len, T, F, (int)mask_norm,
                     (int)complement, out);
  return aps_launch_status();
}

extern "C" int aps_mvdr_beamform_features(const float* store, const float* weight, int64_t N, int64_t C,
                                          int64_t T, int64_t F, int64_t stride_n, int64_t stride_c,
                                          int64_t stride_t, float abs_eps, const aps_feat_params* p,
                                          const int32_t* mel_start, const int32_t* mel_len,
                                          const int32_t* mel_off, const float* mel_w, float* y_out,
                                          float* feats_out, int32_t* nan_count, void* stream) {
  APS_CHECK_ARG(store && weight && feats_out && p && N > 0 && N <= 65535 && T > 0 && F > 0);
  APS_CHECK_ARG(p->num_bins == F && p->num_mels >= 0 && (p->num_mels == 0 || (mel_start && mel_len && mel_off && mel_w)));
  hipStream_t st = static_cast<hipStream_t>(stream);
  int fpw = 4;  // (the rule of aps_mvdr_beamform)
  while (fpw > 1 && ((T + 4 * fpw - 1) / (4 * fpw)) * N < 1024) fpw >>= 1;
  const int D = p->num_mels > 0 ? p->num_mels : (int)F;
  BeamFeatArgs a{store, weight, y_out, feats_out, mel_start, mel_len, mel_off, mel_w, nan_count, T, F,
                 stride_n, stride_c, stride_t, fpw, p->power, p->num_mels, p->apply_log, p->norm_mean,
                 p->norm_var, D, abs_eps, p->log_eps, p->log_lower_bound, p->cmvn_eps};
  const size_t lds = ((size_t)4 * (fpw * F + (D > F ? D : F)) + kBeamMelCap) * sizeof(float);
  if (lds > 64 * 1024) return APS_ERR_UNSUPPORTED;
  dim3 grid((unsigned)((T + 4 * fpw - 1) / (4 * fpw)), (unsigned)N);
  APS_DISPATCH_C(C, { hipLaunchKernelGGL((beamform_features_kernel<kC>), grid, dim3(256), lds, st, a); });
  return aps_launch_status();
}

extern "C" int64_t aps_mvdr_covariance_workspace(int64_t N, int64_t C, int64_t T, int64_t F) {
  if (N <= 0 || C < 2 || C > 8 || T <= 0 || F <= 0) return -1;
  const int64_t nv = 2 * C * (C + 1) + 4;
  return (N * kCovMaxSegments * nv * F + N * 2 * F) * (int64_t)sizeof(float);
}

// covariance partials only (shared by aps_mvdr_covariance and aps_mvdr_weights)
static int launch_cov_partials(const float* store, int64_t N, int64_t C, int64_t T, int64_t F,
                               int64_t stride_n, int64_t stride_c, int64_t stride_t,
                               const float* mask_s, const float* mask_n, int64_t mask_ld, const int64_t* x_len,
                               int32_t mask_norm, float* pmask_s, float* pmask_n, float* workspace,
                               hipStream_t st, int* ts_out, int* pre_out) {
  if (mask_ld == 0) mask_ld = F;
  const char* tb = getenv("APS_COV_BINS");  // tuning only
  const int bins = (tb && tb[0] == '3') ? 32 : 64;
  const int64_t fblocks = (F + bins - 1) / bins;
  const int64_t fblocks32 = (F + kCovBins - 1) / kCovBins;
  // frame segments: ~2 workgroups per CU (measured optimum: 480 workgroups at N=32, F=257) and
  // at least 16 frames per segment
  int64_t TS = (512 + fblocks * N / 2) / (fblocks * N);
  if (TS > kCovMaxSegments) TS = kCovMaxSegments;
  if (TS > T / 16) TS = T / 16;
  if (TS < 1) TS = 1;
  const char* tune = getenv("APS_COV_SEGMENTS");  // tuning only
  if (tune && tune[0] >= '1' && tune[0] <= '8') TS = tune[0] - '0';
  const int seg_len = (int)((T + TS - 1) / TS);
  const int64_t nv = 2 * C * (C + 1) + 4;
  float* partial = workspace;
  float* pre_div = workspace + N * kCovMaxSegments * nv * F;
  // exact reference order of operations is needed when d enters before the accumulation
  const bool exact = mask_norm && (mask_n == nullptr || pmask_s != nullptr || pmask_n != nullptr);
  const bool pre = exact || (!mask_norm && (mask_n == nullptr || pmask_s || pmask_n));
  const int64_t blocks = fblocks * N * TS;
  if (blocks > (int64_t)1 << 30) return APS_ERR_UNSUPPORTED;
  dim3 grid((unsigned)((blocks + 7) / 8 * 8));
  if (exact) {
    hipLaunchKernelGGL(mask_max_kernel, dim3((unsigned)fblocks32, (unsigned)N), dim3(256), 0, st,
                       mask_s, mask_n, x_len, T, F, mask_ld, pre_div);
  }
  CovArgs a{store, mask_s, mask_n, x_len, pre ? pre_div : nullptr, partial, pmask_s, pmask_n, T, F,
            stride_n, stride_c, stride_t, mask_norm, seg_len, mask_ld, (int32_t)fblocks, (int32_t)N, (int32_t)TS};
  APS_DISPATCH_C(C, {
    size_t lds = (size_t)4 * CovLayout<kC>::NV * bins * sizeof(float);
    if (bins == 64) {
      if (lds > 48 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void*>(&covariance_partial_kernel<kC, 64>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((covariance_partial_kernel<kC, 64>), grid, dim3(256), lds, st, a);
    } else {
      if (lds > 48 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void*>(&covariance_partial_kernel<kC, 32>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((covariance_partial_kernel<kC, 32>), grid, dim3(256), lds, st, a);
    }
  });
  *ts_out = (int)TS;
  *pre_out = pre ? 1 : 0;
  return APS_OK;
}

extern "C" int aps_mvdr_covariance(const float* store, int64_t N, int64_t C, int64_t T, int64_t F,
                                   int64_t stride_n, int64_t stride_c, int64_t stride_t,
                                   const float* mask_s, const float* mask_n, int64_t mask_ld,
                                   const int64_t* x_len,
                                   int32_t mask_norm, float* cov_s, float* cov_n, float* offdiag,
                                   float* pmask_s, float* pmask_n, float* workspace,
                                   void* stream) {
  APS_CHECK_ARG(store && mask_s && cov_s && cov_n && workspace);
  APS_CHECK_ARG(N > 0 && N <= 65535 && T > 0 && F > 0 && (mask_ld == 0 || mask_ld >= F));
  if (C < 2 || C > 8) return APS_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int TS = 1, pre = 0;
  int rc = launch_cov_partials(store, N, C, T, F, stride_n, stride_c, stride_t, mask_s, mask_n, mask_ld,
                               x_len, mask_norm, pmask_s, pmask_n, workspace, st, &TS, &pre);
  if (rc != APS_OK) return rc;
  const float* partial = workspace;
  APS_DISPATCH_C(C, {
    const dim3 fgrid((unsigned)((N * F + 63) / 64));
    if (TS <= 4)
      hipLaunchKernelGGL((covariance_finalize_kernel<kC, 4>), fgrid, dim3(64), 0, st, partial,
                         N * F, F, TS, (int)mask_norm, pre, cov_s, cov_n, offdiag, nullptr);
    else
      hipLaunchKernelGGL((covariance_finalize_kernel<kC, kCovMaxSegments>), fgrid, dim3(64), 0, st,
                         partial, N * F, F, TS, (int)mask_norm, pre, cov_s, cov_n, offdiag, nullptr);
  });
  return aps_launch_status();
}

static int attention_chunks(int64_t C, int64_t A);

extern "C" int64_t aps_mvdr_weights_workspace(int64_t N, int64_t C, int64_t T, int64_t F,
                                              int64_t A) {
  const int64_t cov = aps_mvdr_covariance_workspace(N, C, T, F);
  if (cov < 0 || A <= 0) return -1;
  // + packed Rs|Rn [N][2 NU][F] + offdiag [N][C][F] + attention scores [N][C][chunks]
  return cov + (N * 2 * C * (C + 1) * F + N * C * F + N * C * attention_chunks(C, A)) *
                   (int64_t)sizeof(float);
}

extern "C" int aps_mvdr_weights(const float* store, int64_t N, int64_t C, int64_t T, int64_t F,
                                int64_t stride_n, int64_t stride_c, int64_t stride_t,
                                const float* mask_s, const float* mask_n, int64_t mask_ld,
                                const int64_t* x_len,
                                int32_t mask_norm, int64_t A, const float* proj_w,
                                const float* proj_b, const float* gvec_w, const float* gvec_b,
                                float eps, float* workspace, float* cov_s, float* cov_n,
                                float* u_out, float* weight_out, int32_t* singular_count, void* stream) {
  APS_CHECK_ARG(store && mask_s && workspace && proj_w && proj_b && gvec_w && gvec_b && u_out &&
                weight_out);
  APS_CHECK_ARG((cov_s == nullptr) == (cov_n == nullptr));
  APS_CHECK_ARG(N > 0 && N <= 65535 && T > 0 && F > 0 && A > 0 && (mask_ld == 0 || mask_ld >= F));
  if (C < 2 || C > 8) return APS_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int TS = 1, pre = 0;
  int rc = launch_cov_partials(store, N, C, T, F, stride_n, stride_c, stride_t, mask_s, mask_n, mask_ld,
                               x_len, mask_norm, nullptr, nullptr, workspace, st, &TS, &pre);
  if (rc != APS_OK) return rc;
  const float* partial = workspace;
  float* packed = workspace + aps_mvdr_covariance_workspace(N, C, T, F) / (int64_t)sizeof(float);
  float* offdiag = packed + N * 2 * C * (C + 1) * F;
  float* scores = offdiag + N * C * F;
  const int nchunk = attention_chunks(C, A);
  const int64_t NF = N * F;
  // Four launches: segment partials, fold, channel attention, solve.  Measured and NOT kept (each slower
  // than this sequence's 45 us at 32 utterances per launch):
  //  * round 2: fold + attention + softmax + solve of one utterance per workgroup (61 us: one workgroup
  //    per utterance serialises what the three launches spread over N x 8 workgroups);
  //  * round 4: covariance + fold + per-bin solve in one launch with nothing crossing a workgroup (a
  //    workgroup owns 32 or 16 bins x ALL frames), then attention, then the projection of Y / tr on u:
  //    49 / 55 us -- 42 us for the covariance launch alone against 21 + 10 for partials + fold
  //    (profiles/r04_frontend_cov_solve_variant_kernel_stats.csv): 9 x N or 17 x N workgroups keep too
  //    few requests in flight for a launch bound by the spectrogram's way out of HBM, and the solve runs
  //    in one lane of 8 or 16;
  //  * round 5: fold + attention + solve as ONE launch behind the partials, the last of an utterance's 5
  //    workgroups to arrive running its attention (8 waves x 16 rows of P per pass) and its 257 solves:
  //    75 us for the stage against 44.5 (profiles/r05_rejected_experiments.txt (3)).
  APS_DISPATCH_C(C, {
    const dim3 fgrid((unsigned)((NF + 63) / 64));
    if (TS <= 4)
      hipLaunchKernelGGL((covariance_finalize_kernel<kC, 4>), fgrid, dim3(64), 0, st, partial, NF,
                         F, TS, (int)mask_norm, pre, cov_s, cov_n, offdiag, packed);
    else
      hipLaunchKernelGGL((covariance_finalize_kernel<kC, kCovMaxSegments>), fgrid, dim3(64), 0, st,
                         partial, NF, F, TS, (int)mask_norm, pre, cov_s, cov_n, offdiag, packed);
    size_t lds = (size_t)kC * F * sizeof(float);
    if (lds > 150 * 1024) return APS_ERR_UNSUPPORTED;
    if (lds > 48 * 1024)
      hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_partial_kernel<kC, true>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((attention_partial_kernel<kC, true>), dim3((unsigned)nchunk, (unsigned)N),
                       dim3(256), lds, st, offdiag, F, A, proj_w, proj_b, gvec_w, scores);
    hipLaunchKernelGGL((weight_kernel<kC, true, true>), fgrid, dim3(64), 0, st, packed, nullptr,
                       nullptr, NF, F, eps, weight_out, scores, nchunk, gvec_b, u_out, singular_count);
  });
  return aps_launch_status();
}

static int attention_chunks(int64_t C, int64_t A) {
  const int rows_per_block = 4 * (int)(64 / C);
  return (int)((A + rows_per_block - 1) / rows_per_block);
}

extern "C" int64_t aps_mvdr_attention_scratch(int64_t N, int64_t C, int64_t A) {
  if (N <= 0 || C < 2 || C > 8 || A <= 0) return -1;
  return N * C * attention_chunks(C, A) * (int64_t)sizeof(float);
}

extern "C" int aps_mvdr_channel_attention(const float* cov_s, int64_t N, int64_t C, int64_t F,
                                          int64_t A, const float* proj_w, const float* proj_b,
                                          const float* gvec_w, const float* gvec_b, float* scratch,
                                          float* u_out, void* stream) {
  APS_CHECK_ARG(cov_s && proj_w && proj_b && gvec_w && gvec_b && scratch && u_out);
  APS_CHECK_ARG(N > 0 && N <= 65535 && F > 0 && A > 0);
  if (C < 2 || C > 8) return APS_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nchunk = attention_chunks(C, A);
  dim3 grid((unsigned)nchunk, (unsigned)N);
  APS_DISPATCH_C(C, {
    size_t lds = (size_t)kC * F * sizeof(float);
    if (lds > 150 * 1024) return APS_ERR_UNSUPPORTED;
    if (lds > 48 * 1024)
      hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_partial_kernel<kC, false>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((attention_partial_kernel<kC, false>), grid, dim3(256), lds, st, cov_s, F, A,
                       proj_w, proj_b, gvec_w, scratch);
  });
  hipLaunchKernelGGL(attention_finalize_kernel, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, st,
                     scratch, N, (int)C, nchunk, gvec_b, u_out);
  return aps_launch_status();
}

extern "C" int aps_mvdr_weight(const float* cov_s, const float* cov_n, const float* u, int64_t N,
                               int64_t C, int64_t F, float eps, float* weight_out, int32_t* singular_count,
                               void* stream) {
  APS_CHECK_ARG(cov_s && cov_n && u && weight_out && N > 0 && F > 0);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t NF = N * F;
  dim3 grid((unsigned)((NF + 63) / 64));
  APS_DISPATCH_C(C, {
    hipLaunchKernelGGL((weight_kernel<kC, false, false>), grid, dim3(64), 0, st, cov_s, cov_n, u, NF, F,
                       eps, weight_out, nullptr, 0, nullptr, nullptr, singular_count);
  });
  return aps_launch_status();
}

extern "C" int aps_mvdr_attention_weight(const float* cov_s, const float* cov_n,
                                         const float* offdiag, int64_t N, int64_t C, int64_t F,
                                         int64_t A, const float* proj_w,
                                         const float* proj_b, const float* gvec_w,
                                         const float* gvec_b, float eps, float* scratch,
                                         float* u_out, float* weight_out, int32_t* singular_count,
                                         void* stream) {
  APS_CHECK_ARG(cov_s && cov_n && proj_w && proj_b && gvec_w && gvec_b && scratch && u_out &&
                weight_out);
  APS_CHECK_ARG(N > 0 && N <= 65535 && F > 0 && A > 0);
  if (C < 2 || C > 8) return APS_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nchunk = attention_chunks(C, A);
  const int64_t NF = N * F;
  APS_DISPATCH_C(C, {
    size_t lds = (size_t)kC * F * sizeof(float);
    if (lds > 150 * 1024) return APS_ERR_UNSUPPORTED;
    if (offdiag) {
      if (lds > 48 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_partial_kernel<kC, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((attention_partial_kernel<kC, true>), dim3((unsigned)nchunk, (unsigned)N),
                         dim3(256), lds, st, offdiag, F, A, proj_w, proj_b, gvec_w, scratch);
    } else {
      if (lds > 48 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_partial_kernel<kC, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((attention_partial_kernel<kC, false>), dim3((unsigned)nchunk, (unsigned)N),
                         dim3(256), lds, st, cov_s, F, A, proj_w, proj_b, gvec_w, scratch);
    }
    hipLaunchKernelGGL((weight_kernel<kC, true, false>), dim3((unsigned)((NF + 63) / 64)), dim3(64), 0,
                       st, cov_s, cov_n, nullptr, NF, F, eps, weight_out, scratch, nchunk, gvec_b,
                       u_out, singular_count);
  });
  return aps_launch_status();
}

extern "C" int aps_mvdr_beamform(const float* store, const float* weight, int64_t N, int64_t C,
                                 int64_t T, int64_t F, int64_t stride_n, int64_t stride_c,
                                 int64_t stride_t, float* y_out, void* stream) {
  APS_CHECK_ARG(store && weight && y_out && N > 0 && N <= 65535 && T > 0 && F > 0);
  hipStream_t st = static_cast<hipStream_t>(stream);
  // frames per wavefront: reuse of the per-bin weights vs. enough wavefronts to fill the chip
  int fpw = 4;
  while (fpw > 1 && ((T + 4 * fpw - 1) / (4 * fpw)) * N < 1024) fpw >>= 1;
  const char* tune = getenv("APS_BF_FRAMES");  // tuning only
  if (tune && tune[0] >= '1' && tune[0] <= '9') fpw = tune[0] - '0';
  dim3 grid((unsigned)((T + 4 * fpw - 1) / (4 * fpw)), (unsigned)N);
  APS_DISPATCH_C(C, {
    // the register-resident variant (all loads of a row up front, NITER = 5) measured 22 us vs 18 us
    // for the rolled one at N=32: occupancy wins over per-wave load batching here
    hipLaunchKernelGGL((beamform_kernel<kC, 0>), grid, dim3(256), 0, st, store, weight, T, F,
                       stride_n, stride_c, stride_t, y_out, fpw);
  });
  return aps_launch_status();
}
