// 2-D convolution / transposed convolution, channels-last, as an implicit GEMM on fp32 MFMA with
// the bias / BatchNorm-affine / activation / residual epilogue fused in.  Serves
//   * the Conv2d + BatchNorm2d + ReLU subsampling blocks of the encoder projection
//     (aps/asr/base/component.py:251-307, aps/asr/base/encoder.py:367-441), and
//   * the (complex) Conv2d / ConvTranspose2d + BatchNorm2d + LeakyReLU blocks of DCCRN / DCUNet
//     (aps/sse/enh/dcunet.py:24-170), a complex layer being ONE real layer on real|imag-stacked
//     channels with the block weight [[Wr, -Wi], [Wi, Wr]] (built on the host).
//
// Layout: activations x [N, H, W, Ci], y [N, Ho, Wo, Co] (channels fastest), weights
// w [Co, KH, KW, Ci] (K = (kh, kw, ci), ci fastest) -- so an output pixel is a GEMM row, its
// receptive field a sequence of Ci-long contiguous runs, the weight matrix K-contiguous like an
// nn.Linear weight, and the output of one layer is the input of the next without any transpose.
//   M = N Ho Wo,  N_gemm = Co,  K = KH KW Ci
// The kernel is the 64 x 64 x 32 GEMM of nn.hip (row-major LDS tiles, b128 operand fetch, loads two
// K tiles ahead, dual accumulators) with a gathering A loader: a K tile of 32 lies inside one tap
// (Ci % 32 == 0), so per tile and staged row there is one validity test (padding / stride
// divisibility of the transposed form) and one base offset; invalid rows aim outside the buffer
// and read zeros through the descriptor's range check.  Layers with tiny Ci (the first layer:
// 1 or 2 channels) take a VALU kernel instead (conv_smallk_kernel for K <= 32, else
// conv_direct_kernel).
//
// Strided transposed convolutions: output pixel (ho, wo) only meets the taps with
// kh = (ho + ph) mod sh, kw = (wo + pw) mod sw (the others fall into the stride holes), so the GEMM
// rows are ordered by residue class (ho mod sh, wo mod sw): every 64-row tile is of ONE class and
// simply iterates over that class's taps (an arithmetic progression) -- no MFMA is spent on holes
// (DCCRN's (1, 2)-strided decoder: 1.5 of 3 frequency taps on average, i.e. half the work).
#include <type_traits>

#include "common.h"
#include "conv_core.h"

namespace aps {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kCT = 64, kCBK = 32, kCPitch = kCBK + 4;

__global__ __launch_bounds__(256, 3) void conv_mfma_kernel(ConvArgs g) {
  extern __shared__ __attribute__((aligned(16))) float s_conv[];  // [2][128][kCPitch] | row pixel [64]
  constexpr int kBufFloats = 2 * kCT * kCPitch;
  int* s_pix = reinterpret_cast<int*>(s_conv + 2 * kBufFloats);
  const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
  const int wm = wv >> 1, wn = wv & 1;
  const int tiles_n = (g.Co + kCT - 1) / kCT;
  int64_t mt = blockIdx.x / tiles_n;  // row tile
  const int n0 = (blockIdx.x % tiles_n) * kCT;
  const int sr = tid >> 3, sc = (tid & 7) * 4;  // staged rows sr, sr + 32; float4 column sc

  f32x16 acc[2];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;

  // row ordering: plain (n, ho, wo), or by stride residue class (transposed form): class (qh, qw)
  // holds the pixels ho = qh + sh jh, wo = qw + sw jw in (n, jh, jw) order
  int qh = 0, qw = 0, Hc = g.Ho, Wc = g.Wo, step_h = 1, step_w = 1;
  int64_t Mc = g.M;
  if (g.by_class) {
    step_h = g.sh, step_w = g.sw;
    for (int cls = 0; cls < g.sh * g.sw; ++cls) {
      qh = cls / g.sw, qw = cls - qh * g.sw;
      Mc = class_rows(g.N, g.Ho, g.Wo, g.sh, g.sw, qh, qw, Hc, Wc);
      const int64_t tc = (Mc + kCT - 1) / kCT;
      if (mt < tc) break;
      mt -= tc;
    }
  }
  const int64_t m0 = mt * kCT;
  // live taps of this tile: kh = kh0 + step_h i (i < nkh), kw = kw0 + step_w i (i < nkw)
  const int kh0 = g.by_class ? (qh + g.ph) % g.sh : 0, kw0 = g.by_class ? (qw + g.pw) % g.sw : 0;
  const int nkh = kh0 < g.KH ? (g.KH - kh0 + step_h - 1) / step_h : 0;
  const int nkw = kw0 < g.KW ? (g.KW - kw0 + step_w - 1) / step_w : 0;

  // staged rows -> (image, output row, output column)
  int rn[2], rho[2], rwo[2];
  bool rvalid[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int64_t m = m0 + sr + 32 * i;
    rvalid[i] = m < Mc;
    const int64_t mm = rvalid[i] ? m : 0;
    rwo[i] = qw + step_w * (int)(mm % Wc);
    rho[i] = qh + step_h * (int)((mm / Wc) % Hc);
    rn[i] = (int)(mm / ((int64_t)Wc * Hc));
    if ((tid & 7) == 0)
      s_pix[sr + 32 * i] = rvalid[i] ? (rn[i] * g.Ho + rho[i]) * g.Wo + rwo[i] : -1;
  }
  const int chunks = g.Ci / kCBK;              // K tiles per tap
  const int ntiles = nkh * nkw * chunks;
  const uint32_t x_bytes = (uint32_t)((int64_t)g.N * g.H * g.W * g.Ci * 4);
  const uint32_t w_bytes = (uint32_t)((int64_t)g.Co * g.KH * g.KW * g.Ci * 4);
  auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.x), 0, x_bytes, 0x00020000);
  auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.w), 0, w_bytes, 0x00020000);
  const int K = g.KH * g.KW * g.Ci;
  int32_t vb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) vb[i] = (int32_t)(min(n0 + sr + 32 * i, g.Co - 1) * (int64_t)K * 4) + sc * 4;

  u32x4 ra[2][2], rb[2][2];
  // one b128 request of K tile `tile`: q = 0, 1 the two staged A rows (gathered), 2, 3 the W rows
  auto load1 = [&](auto stage, int q, int tile) {
    constexpr int P = decltype(stage)::value;
    tile = min(tile, ntiles - 1);
    const int tap = tile / chunks, c0 = (tile - tap * chunks) * kCBK;
    const int ih = tap / nkw;
    const int kh = kh0 + step_h * ih, kw = kw0 + step_w * (tap - ih * nkw);
    if (q < 2) {
      int hi, wi;
      const bool ok = tap_coord(rho[q], kh, g.sh, g.ph, g.H, g.transposed, hi) &
                      tap_coord(rwo[q], kw, g.sw, g.pw, g.W, g.transposed, wi) & rvalid[q];
      const uint32_t off = ok ? (uint32_t)((((int64_t)rn[q] * g.H + hi) * g.W + wi) * g.Ci + c0 + sc) * 4u
                              : 0xfffffff0u;  // outside the buffer: reads zeros
      ra[P][q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, off, 0, 0);
    } else {
      const int32_t soff = ((kh * g.KW + kw) * g.Ci + c0) * 4;
      rb[P][q - 2] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, vb[q - 2], soff, 0);
    }
  };
  auto gload = [&](auto stage, int tile) {
#pragma unroll
    for (int q = 0; q < 4; ++q) load1(stage, q, tile);
  };
  auto store1 = [&](auto stage, int q, int buf) {
    constexpr int P = decltype(stage)::value;
    float* sa = s_conv + buf * kBufFloats;
    if (q < 2)
      *reinterpret_cast<u32x4*>(sa + (sr + 32 * q) * kCPitch + sc) = ra[P][q];
    else
      *reinterpret_cast<u32x4*>(sa + (kCT + sr + 32 * (q - 2)) * kCPitch + sc) = rb[P][q - 2];
  };
  auto sstore = [&](auto stage, int buf) {
#pragma unroll
    for (int q = 0; q < 4; ++q) store1(stage, q, buf);
  };
  // K step scheduled by hand like the GEMM's (nn.hip, SWP): two operand register sets, ONE memory
  // instruction in the shadow of every MFMA, barrier that only waits for this wave's LDS writes
  const int frow = ln & 31, fk = (ln >> 5) * 4;
  const float* fa = s_conv + (wm * 32 + frow) * kCPitch + fk;
  const float* fb = s_conv + (kCT + wn * 32 + frow) * kCPitch + fk;
  f32x4 xo[4], yo[4];  // k groups 0-1 (X) / 2-3 (Y); elements 0, 1: A, 2, 3: W
  auto read1 = [&](f32x4 (&o)[4], int q, int buf, int koff) {
    const float* p = ((q < 2) ? fa : fb) + buf * kBufFloats + koff + (q & 1) * 8;
    o[q] = *reinterpret_cast<const f32x4*>(p);
  };
  auto mfma1 = [&](const f32x4 (&o)[4], int i) {
    const int q = i >> 2, e = i & 3;
    acc[e & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(o[q][e], o[2 + q][e], acc[e & 1], 0, 0, 0);
  };
  auto half_a = [&](auto stage, int cur, int nxt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      mfma1(xo, i);
      store1(stage, i, nxt);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      mfma1(xo, 4 + i);
      read1(yo, (i & 1) * 2 + (i >> 1), cur, 16);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto barrier_after_writes = [&]() {
    asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");  // LDS returns in order: the writes are done
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto half_b = [&](auto stage, int nxt, int tile) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      mfma1(yo, 2 * i);
      read1(xo, (i & 1) * 2 + (i >> 1), nxt, 0);
      __builtin_amdgcn_sched_barrier(0);
      mfma1(yo, 2 * i + 1);
      load1(stage, i, tile);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  if (ntiles > 0) {  // (a class without live taps only gets the epilogue's shift)
    // tile t travels through register stage t & 1 and LDS buffer t & 1
    gload(S0{}, 0);
    gload(S1{}, 1);
    sstore(S0{}, 0);
    gload(S0{}, 2);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) read1(xo, q, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    int s = 0;
    for (; s + 1 < ntiles; s += 2) {
      half_a(S1{}, 0, 1);
      barrier_after_writes();
      half_b(S1{}, 1, s + 3);
      half_a(S0{}, 1, 0);
      barrier_after_writes();
      half_b(S0{}, 0, s + 4);
    }
    if (s < ntiles) {
#pragma unroll
      for (int q = 0; q < 4; ++q) read1(yo, q, 0, 16);
#pragma unroll
      for (int i = 0; i < 8; ++i) mfma1(xo, i);
#pragma unroll
      for (int i = 0; i < 8; ++i) mfma1(yo, i);
    }
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[0][e] += acc[1][e];

  // epilogue: C/D layout col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
  const int col = n0 + wn * 32 + (ln & 31);
  if (col >= g.Co) return;
  const float sc_ = g.scale ? g.scale[col] : 1.f, sh_ = g.shift ? g.shift[col] : 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int64_t row = s_pix[wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * (ln >> 5)];
    if (row < 0) continue;
    float v = conv_act(acc[0][e] * sc_ + sh_, g.act, g.slope);
    if (g.residual) v += g.residual[row * g.Co + col];
    g.y[row * g.Co + col] = v;
  }
}

// Direct form for any Ci (used when Ci % 32 != 0: the 1- / 2-channel first layers).  A workgroup
// owns 64 output channels x 64 consecutive output pixels: the receptive fields of its pixels are
// gathered once into LDS as an im2col patch [64][K] (one bounds-checked global load per element,
// all in flight together), the weights of its channels sit in LDS as [K][64]; a wave's 64 lanes are
// the channels (coalesced 256-byte stores), the patch values are LDS broadcasts.
constexpr int kDirectPix = 64;  // upper bound; fewer when the patch would not fit in LDS

__global__ __launch_bounds__(256) void conv_direct_kernel(ConvArgs g) {
  extern __shared__ float s_dir[];  // weights [K][64] | patch [64][K + 1]
  const int K = g.KH * g.KW * g.Ci;
  float* s_w = s_dir;
  float* s_x = s_dir + 64 * K;
  const int co0 = blockIdx.y * 64;
  const int pix = g.direct_pix;
  const int64_t m0 = (int64_t)blockIdx.x * pix;
  for (int i = threadIdx.x; i < 64 * K; i += 256) {
    const int c = i & 63, k = i >> 6;
    s_w[i] = (co0 + c < g.Co) ? g.w[(int64_t)(co0 + c) * K + k] : 0.f;
  }
  for (int i = threadIdx.x; i < pix * K; i += 256) {
    const int p = i / K, k = i - p * K;
    const int64_t m = m0 + p;
    float v = 0.f;
    if (m < g.M) {
      const int wo = (int)(m % g.Wo), ho = (int)((m / g.Wo) % g.Ho);
      const int n = (int)(m / ((int64_t)g.Wo * g.Ho));
      const int ci = k % g.Ci, tap = k / g.Ci, kw = tap % g.KW, kh = tap / g.KW;
      int hi, wi;
      if (tap_coord(ho, kh, g.sh, g.ph, g.H, g.transposed, hi) &&
          tap_coord(wo, kw, g.sw, g.pw, g.W, g.transposed, wi))
        v = g.x[(((int64_t)n * g.H + hi) * g.W + wi) * g.Ci + ci];
    }
    s_x[p * (K + 1) + k] = v;
  }
  __syncthreads();
  const int c = threadIdx.x & 63, co = co0 + c, wv = threadIdx.x >> 6;
  if (co >= g.Co) return;
  const float sc_ = g.scale ? g.scale[co] : 1.f, sh_ = g.shift ? g.shift[co] : 0.f;
  for (int r = 0; r < pix / 4; ++r) {
    const int p = wv * (pix / 4) + r;
    const int64_t m = m0 + p;
    if (m >= g.M) break;
    const float* xp = s_x + p * (K + 1);
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc += xp[k] * s_w[k * 64 + c];
    float v = conv_act(acc * sc_ + sh_, g.act, g.slope);
    if (g.residual) v += g.residual[m * g.Co + co];
    g.y[m * g.Co + co] = v;
  }
}

// Small receptive fields (K = KH KW Ci <= 32: the 1- / 2-channel first layers, whose cost is the
// N Ho Wo Co output they write, not their arithmetic).  A workgroup owns 64 consecutive output
// pixels: their K-long patches are gathered once into LDS (coordinates and tap decomposition computed
// once per pixel / per tap, not per element), a wavefront's 64 lanes are 64 consecutive output
// channels with their K weights in registers, the patch values are LDS broadcasts (b128), and a
// pixel's Co outputs leave as coalesced 256-byte stores.
template <int KP>
__global__ __launch_bounds__(256) void conv_smallk_kernel(ConvArgs g) {
  constexpr int PIX = 64;
  __shared__ __attribute__((aligned(16))) float s_patch[PIX][KP];
  __shared__ int s_n[PIX], s_ho[PIX], s_wo[PIX];
  __shared__ int s_kh[KP], s_kw[KP], s_ci[KP];
  const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
  const int K = g.KH * g.KW * g.Ci;
  const int64_t m0 = (int64_t)blockIdx.x * PIX;
  if (tid < PIX) {
    const int64_t m = min(m0 + tid, g.M - 1);
    s_wo[tid] = (int)(m % g.Wo);
    s_ho[tid] = (int)((m / g.Wo) % g.Ho);
    s_n[tid] = (int)(m / ((int64_t)g.Wo * g.Ho));
  } else if (tid < PIX + KP) {
    const int k = tid - PIX, tap = k / g.Ci;
    s_ci[k] = k - tap * g.Ci;
    s_kw[k] = tap % g.KW;
    s_kh[k] = tap / g.KW;
  }
  __syncthreads();
  for (int e = tid; e < PIX * KP; e += 256) {
    const int p = e / KP, k = e % KP;
    float v = 0.f;
    int hi, wi;
    if (k < K && m0 + p < g.M &&
        tap_coord(s_ho[p], s_kh[k], g.sh, g.ph, g.H, g.transposed, hi) &&
        tap_coord(s_wo[p], s_kw[k], g.sw, g.pw, g.W, g.transposed, wi))
      v = g.x[(((int64_t)s_n[p] * g.H + hi) * g.W + wi) * g.Ci + s_ci[k]];
    s_patch[p][k] = v;
  }
  __syncthreads();
  // wave -> (channel group, pixel phase): groups of 64 channels round robin over the 4 waves; when
  // there are fewer than 4 groups the spare waves split the pixels of a group
  const int groups = (g.Co + 63) / 64;
  const int gw = groups < 4 ? groups : 4;       // waves that differ in channel group
  const int share = 4 / gw;                     // waves sharing a channel group (pixel phases)
  if (wv >= gw * share) return;
  const int phase = wv / gw;
  for (int cg = wv % gw; cg < groups; cg += gw) {
    const int co = cg * 64 + ln;
    const int coc = min(co, g.Co - 1);
    float wr[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) wr[k] = (k < K) ? g.w[(int64_t)coc * K + k] : 0.f;
    const float sc_ = g.scale ? g.scale[coc] : 1.f, sh_ = g.shift ? g.shift[coc] : 0.f;
    for (int p = phase; p < PIX; p += share) {
      const int64_t m = m0 + p;
      if (m >= g.M) break;
      float acc = 0.f;
#pragma unroll
      for (int k4 = 0; k4 < KP; k4 += 4) {
        const float4 x = *reinterpret_cast<const float4*>(&s_patch[p][k4]);
        acc += x.x * wr[k4] + x.y * wr[k4 + 1] + x.z * wr[k4 + 2] + x.w * wr[k4 + 3];
      }
      float v = conv_act(acc * sc_ + sh_, g.act, g.slope);
      if (co < g.Co) {
        if (g.residual) v += g.residual[m * g.Co + co];
        g.y[m * g.Co + co] = v;
      }
    }
  }
}


// ------------------------------------------------------------------------------------------
// Few output channels (Co <= 4: the mask layer that closes DCCRN's decoder, 32 -> 2 S channels).
// On the MFMA tile 4 of 64 columns are live and the layer is bound by the 131 MB of input it reads
// (4 TF on conv_mfma_kernel: 560 us).  Here a workgroup owns ONE output row (n, ho): the KH input
// rows it meets are staged in LDS once (coalesced 16-byte requests; row pitch Ci + 1 floats, so the
// lanes of a wave -- consecutive wo, i.e. input columns one or sw apart -- fall on distinct banks),
// the weights as [tap][ci][4], and every thread accumulates the 4 channels of its output pixels
// from LDS: one ds_read_b32 of x and one broadcast ds_read_b128 of w per 4 FMAs.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_fewout_kernel(ConvArgs g) {
  extern __shared__ __attribute__((aligned(16))) float s_few[];  // w [KH KW][Ci][4] | x [KH][W][Ci + 1]
  const int taps = g.KH * g.KW, pitch = g.Ci + 1;
  float* s_w = s_few;
  float* s_x = s_few + taps * g.Ci * 4;
  const int tid = threadIdx.x;
  const int n = blockIdx.x / g.Ho, ho = blockIdx.x - n * g.Ho;
  for (int e = tid; e < taps * g.Ci * 4; e += 256) {
    const int co = e & 3, ci = (e >> 2) % g.Ci, tap = (e >> 2) / g.Ci;
    s_w[e] = co < g.Co ? g.w[((int64_t)co * taps + tap) * g.Ci + ci] : 0.f;
  }
  // input rows: hi(kh) or nothing when the tap falls into padding / a stride hole
  const int row_f4 = g.W * g.Ci / 4;
  for (int kh = 0; kh < g.KH; ++kh) {
    int hi;
    const bool ok = tap_coord(ho, kh, g.sh, g.ph, g.H, g.transposed, hi);
    const float4* src = reinterpret_cast<const float4*>(g.x + (((int64_t)n * g.H + (ok ? hi : 0)) * g.W) * g.Ci);
    float* dst = s_x + kh * g.W * pitch;
    for (int e = tid; e < row_f4; e += 256) {
      const float4 v = ok ? src[e] : make_float4(0.f, 0.f, 0.f, 0.f);
      const int wi = (e * 4) / g.Ci, ci = (e * 4) - wi * g.Ci;
      float* d = dst + wi * pitch + ci;
      d[0] = v.x, d[1] = v.y, d[2] = v.z, d[3] = v.w;
    }
  }
  __syncthreads();
  // output columns by stride residue class (transposed form): the lanes of a wave then share their
  // live taps (no divergence) and read consecutive input columns (distinct banks)
  const int classes = g.transposed ? g.sw : 1;
  const int per_class = (g.Wo + classes - 1) / classes;
  for (int idx = tid; idx < classes * per_class; idx += 256) {
    const int wo = (idx / per_class) + classes * (idx % per_class);
    if (wo >= g.Wo) continue;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int kh = 0; kh < g.KH; ++kh) {
      for (int kw = 0; kw < g.KW; ++kw) {
        int wi;
        if (!tap_coord(wo, kw, g.sw, g.pw, g.W, g.transposed, wi)) continue;
        const float* xs = s_x + (kh * g.W + wi) * pitch;
        const float4* ws = reinterpret_cast<const float4*>(s_w + (kh * g.KW + kw) * g.Ci * 4);
#pragma unroll 8
        for (int ci = 0; ci < g.Ci; ++ci) {
          const float xv = xs[ci];
          const float4 wv = ws[ci];
          acc[0] = fmaf(xv, wv.x, acc[0]);
          acc[1] = fmaf(xv, wv.y, acc[1]);
          acc[2] = fmaf(xv, wv.z, acc[2]);
          acc[3] = fmaf(xv, wv.w, acc[3]);
        }
      }
    }
    const int64_t row = ((int64_t)n * g.Ho + ho) * g.Wo + wo;
    for (int co = 0; co < g.Co; ++co) {
      const float sc_ = g.scale ? g.scale[co] : 1.f, sh_ = g.shift ? g.shift[co] : 0.f;
      float v = conv_act(acc[co] * sc_ + sh_, g.act, g.slope);
      if (g.residual) v += g.residual[row * g.Co + co];
      g.y[row * g.Co + co] = v;
    }
  }
}


}  // namespace aps

using namespace aps;

extern "C" int aps_conv2d_nhwc(const float* x, const float* w, const float* scale,
                               const float* shift, const float* residual, float* y, int64_t N,
                               int64_t H, int64_t W, int64_t Ci, int64_t Co, int64_t KH, int64_t KW,
                               int64_t sh, int64_t sw, int64_t ph, int64_t pw, int64_t Ho,
                               int64_t Wo, int32_t transposed, int32_t act, float slope,
                               void* stream) {
  APS_CHECK_ARG(x && w && y && N > 0 && H > 0 && W > 0 && Ci > 0 && Co > 0 && KH > 0 && KW > 0);
  APS_CHECK_ARG(sh > 0 && sw > 0 && ph >= 0 && pw >= 0 && Ho > 0 && Wo > 0);
  APS_CHECK_ARG(act == 0 || act == 1 || act == 5);
  const int64_t M = N * Ho * Wo;
  if (N * H * W * Ci * 4 >= ((int64_t)1 << 32) - 64 || Co * KH * KW * Ci * 4 >= ((int64_t)1 << 31) ||
      M >= ((int64_t)1 << 31))
    return APS_ERR_UNSUPPORTED;
  ConvArgs g{x, w, scale, shift, residual, y, (int32_t)N, (int32_t)H, (int32_t)W, (int32_t)Ci,
             (int32_t)Ho, (int32_t)Wo, (int32_t)Co, (int32_t)KH, (int32_t)KW, (int32_t)sh,
             (int32_t)sw, (int32_t)ph, (int32_t)pw, transposed, act, slope, M, 0, 0};
  hipStream_t st = static_cast<hipStream_t>(stream);
  // few output channels on a wide input: one workgroup per output row, rows staged in LDS
  const size_t few_lds = ((size_t)KH * KW * Ci * 4 + (size_t)KH * W * (Ci + 1)) * sizeof(float);
  if (Co <= 4 && Ci % 4 == 0 && Ci >= 16 && few_lds <= 64 * 1024 && ((uintptr_t)x & 15) == 0 &&
      N * Ho <= 0x7fffffff && !getenv("APS_CONV_NO_FEWOUT")) {
    hipLaunchKernelGGL(conv_fewout_kernel, dim3((unsigned)(N * Ho)), dim3(256), few_lds, st, g);
    return aps_launch_status();
  }
  if (Ci % kCBK == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0) {
    int64_t tiles_m = (M + kCT - 1) / kCT;
    if (transposed && sh * sw > 1 && sh * sw <= 64 && !getenv("APS_CONV_NO_CLASS")) {
      g.by_class = 1;
      tiles_m = 0;
      for (int cls = 0; cls < sh * sw; ++cls) {
        int Hc, Wc;
        tiles_m += (class_rows(g.N, g.Ho, g.Wo, g.sh, g.sw, cls / g.sw, cls % g.sw, Hc, Wc) + kCT - 1) / kCT;
      }
    }
    const int64_t tiles = tiles_m * ((Co + kCT - 1) / kCT);
    if (tiles > 0x7fffffff) return APS_ERR_UNSUPPORTED;
    const size_t lds = 2 * 2 * (size_t)kCT * kCPitch * sizeof(float) + kCT * sizeof(int);
    hipLaunchKernelGGL(conv_mfma_kernel, dim3((unsigned)tiles), dim3(256), lds, st, g);
  } else if (KH * KW * Ci <= 32 && !getenv("APS_CONV_NO_SMALLK")) {
    const int64_t K = KH * KW * Ci;
    dim3 grid((unsigned)((M + 63) / 64));
    if (K <= 12)
      hipLaunchKernelGGL(conv_smallk_kernel<12>, grid, dim3(256), 0, st, g);
    else if (K <= 20)
      hipLaunchKernelGGL(conv_smallk_kernel<20>, grid, dim3(256), 0, st, g);
    else
      hipLaunchKernelGGL(conv_smallk_kernel<32>, grid, dim3(256), 0, st, g);
  } else {
    const int64_t K = KH * KW * Ci;
    const int64_t budget = 64 * 1024 / 4 - 64 * K;  // floats left for the patch
    int64_t pix = budget / (K + 1) / 4 * 4;
    if (pix > kDirectPix) pix = kDirectPix;
    if (pix < 4) return APS_ERR_UNSUPPORTED;
    g.direct_pix = (int32_t)pix;
    const size_t lds = (size_t)(64 * K + pix * (K + 1)) * sizeof(float);
    dim3 grid((unsigned)((M + pix - 1) / pix), (unsigned)((Co + 63) / 64));
    if (grid.y > 65535) return APS_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(conv_direct_kernel, grid, dim3(256), lds, st, g);
  }
  return aps_launch_status();
}
