// fp32 GEMMs on the bf16 matrix pipe of gfx950 (aps_linear_split*): the encoder projections at
// batch sizes where the fp32 MFMA rate (1/16 of the bf16 rate) is the bound.
//
// Arithmetic.  Every fp32 operand is split BY TRUNCATION into three bf16 planes
//     x = h + m + l        h = top 8 significant bits, m = the next 8, l = the last 8   (exact)
// and a product a b is evaluated as six v_mfma_f32_32x32x16_bf16 with fp32 accumulation:
//     a b ~ m m + h l + l h + h m + m h + h h          (dropped: m l + l m + l l <= 2^-20 |a b|,
// typically 2^-22).  Measured against float64 on MI355X (scripts/split_probe.py, K = 512 / 2048,
// three operand distributions): max error 1.1e-6 / rms 0.9e-7 of the output scale -- the same as
// v_mfma_f32_32x32x2_f32 on the same operands (1.1e-6 / 1.1e-7) and as rocBLAS fp32.  Three
// products (h h + h m + m h) would be 2.9e-5: not used.  Six bf16 MFMAs of K = 16 take 6 x 32
// cycles where eight fp32 MFMAs of K = 2 take 8 x 64: 2.67 x the fp32 matrix rate.
//
// Data.  W is split once per weight (aps_linear_split_weight) into a FRAGMENT-ORDERED image (see the
// kernel below); A stays fp32 in HBM and is split on its way from the staging registers into LDS
// (5.5 VALU operations per element).  Round 2 also carried three kernels that staged BOTH operands'
// planes through LDS (a compiler-scheduled 128 x 64 / 128 x 128 tile, a hand-scheduled one, and a
// producer / consumer wave-specialised one on a ring of three 48 KB buffers): all LDS-bound and slower
// than the weight-direct kernel at every shape of the path (DESIGN.md "Round 2, second half");
// removed in round 3 together with their row-image weight layout and their environment switches.
#include <stdint.h>

#include <type_traits>

#include "common.h"
#include "conv_core.h"

namespace aps {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct SplitGemmArgs {
  const float* A;
  const void* Wp;         // planes image of W (aps_linear_split_weight)
  const float* bias;      // [N] or null
  const float* residual;  // [M, N] (ldc) or null
  float* C;
  int64_t M, N, K;
  int64_t lda, ldc;
  int32_t act;
  float alpha;
  int32_t tiles_n, remap, ksteps;
  const float* ln_cs;  // LayerNorm fold (see GemmArgs in nn.hip): column sums of W diag(gamma)
  float ln_eps;
};

// low halves <- top 16 bits of x0, high halves <- top 16 bits of x1
__device__ __forceinline__ uint32_t pack_hi16(uint32_t x1, uint32_t x0) {
  return __builtin_amdgcn_perm(x1, x0, 0x07060302u);
}

struct Planes8 {
  u32x4 h, m, l;
};

// 8 consecutive fp32 (two float4) -> three packed bf16x8 planes
__device__ __forceinline__ Planes8 split8(u32x4 lo, u32x4 hi) {
  uint32_t x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  uint32_t r1[8], r2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float xf = __uint_as_float(x[e]);
    const float a = xf - __uint_as_float(x[e] & 0xffff0000u);
    r1[e] = __float_as_uint(a);
    r2[e] = __float_as_uint(a - __uint_as_float(r1[e] & 0xffff0000u));
  }
  Planes8 o;
  o.h = u32x4{pack_hi16(x[1], x[0]), pack_hi16(x[3], x[2]), pack_hi16(x[5], x[4]), pack_hi16(x[7], x[6])};
  o.m = u32x4{pack_hi16(r1[1], r1[0]), pack_hi16(r1[3], r1[2]), pack_hi16(r1[5], r1[4]),
              pack_hi16(r1[7], r1[6])};
  o.l = u32x4{pack_hi16(r2[1], r2[0]), pack_hi16(r2[3], r2[2]), pack_hi16(r2[5], r2[4]),
              pack_hi16(r2[7], r2[6])};
  return o;
}

__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                 __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------
// "B direct" form: 64 x 128 tile, four wavefronts side by side along N (each 64 x 32).  A wave is
// then the ONLY reader of its 32 weight columns, so the weight planes never pass through LDS: they
// come from a FRAGMENT-ORDERED image (layout 1: for every K step and 32-column group the six
// 1 KB operand registers of a wave -- 2 MFMA K steps x 3 planes -- lane after lane), one
// buffer_load_b128 per operand register, 8 full cache lines per request.  LDS only holds the three
// planes of the 64 A rows (12 KB per buffer, double buffered, ONE barrier per K step):
//   LDS traffic per MFMA  v1 128 x 64: 384 B written + 0.75 fetches    this: 128 B + 0.5 fetches
// -- the LDS pipe, which bounds the 128 x 64 kernel (its writes + fetches take as long as its
// MFMAs), is at < 50 % here.  ~100 VGPRs, 24 KB of LDS: four and more workgroups per CU.
// ------------------------------------------------------------------------------------------
template <int TM, bool LN>
// (launch bounds: the LayerNorm-fold form needs 138 VGPRs unconstrained = three waves per SIMD; held
// to 128 -- one spilled dword -- it runs four like the plain form: 13 190 / 13 210 -> 13 260 / 13 320 utt/s
// on the joint step, same box, alternating builds)
__global__ __launch_bounds__(256, (LN ? 4 : 2)) void gemm_split_bd_kernel(SplitGemmArgs g) {
  constexpr int TN = 128, SM = TM / 32;
  constexpr int kRowB = 64;
  constexpr int kBuf = 3 * TM * kRowB;  // 12 KB: the three A planes of one K step
  constexpr int PA = TM / 32;           // staging passes over the A rows
  __shared__ __attribute__((aligned(16))) unsigned char s_a[2 * kBuf];
  const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
  int64_t lin = blockIdx.x;
  if (g.remap) {
    const int64_t per = gridDim.x / 8;
    lin = (lin & 7) * per + (lin >> 3);
  }
  const int64_t m0 = (lin / g.tiles_n) * TM, n0 = (lin % g.tiles_n) * TN;
  const int arow = tid >> 3, aq = tid & 7;
  const int asw = ((((aq >> 1) ^ ((arow >> 2) & 3)) << 4) | ((aq & 1) << 3));

  f32x16 acc[SM];
#pragma unroll
  for (int i = 0; i < SM; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

  auto rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0,
                                                  (uint32_t)(g.M * g.lda * 4), 0x00020000);
  const int64_t groups = ((g.N + 127) / 128) * 4;  // 32-column groups of the image
  const int32_t wstep_bytes = (int32_t)(groups * 6144);
  auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.Wp), 0,
                                                  (uint32_t)(wstep_bytes * g.ksteps), 0x00020000);
  int32_t va[PA];
#pragma unroll
  for (int i = 0; i < PA; ++i)
    va[i] = (int32_t)(min(m0 + arow + 32 * i, g.M - 1) * g.lda * 4) + aq * 16;
  const int32_t vw = (int32_t)((n0 / 32 + wv) * 6144) + ln * 16;

  const int nsteps = g.ksteps;
  const bool ragged = (g.K & 31) != 0;
  const int rot = (int)((lin / g.tiles_n) % nsteps);
  u32x4 ra[PA];
  u32x4 wb[2][2][3];  // [register stage][MFMA K step][plane]
  // tile_at(s) = (s + rot) mod nsteps without a division per step (s + rot < 2 nsteps)
  auto tile_at = [&](int s) { return (s + rot >= nsteps) ? s + rot - nsteps : s + rot; };
  auto gload_a = [&](int s) {
    const int step = tile_at(s);
    const int32_t soff = step * 128;
    if (ragged && step == nsteps - 1) {
      const int64_t kk = (int64_t)step * 32 + aq * 4;
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        u32x4 v = u32x4{0u, 0u, 0u, 0u};
        if (kk < g.K) v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, va[i], soff, 0);
        v.x = (kk + 0 < g.K) ? v.x : 0u;
        v.y = (kk + 1 < g.K) ? v.y : 0u;
        v.z = (kk + 2 < g.K) ? v.z : 0u;
        v.w = (kk + 3 < g.K) ? v.w : 0u;
        ra[i] = v;
      }
    } else {
#pragma unroll
      for (int i = 0; i < PA; ++i) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, va[i], soff, 0);
    }
  };
  auto gload_w = [&](auto stage, int s) {
    constexpr int P = decltype(stage)::value;
    const int32_t soff = tile_at(s) * wstep_bytes;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        wb[P][kk][p] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, vw, soff + (kk * 3 + p) * 1024, 0);
  };
  float ln_s1[PA], ln_s2[PA];
#pragma unroll
  for (int i = 0; i < PA; ++i) ln_s1[i] = ln_s2[i] = 0.f;
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  auto sstore = [&](int buf) {
    unsigned char* sA = s_a + buf * kBuf;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const uint32_t x[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
      uint32_t r1[4], r2[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float f = __uint_as_float(x[e]);
        if (LN) {
          ln_s1[i] += f;
          ln_s2[i] = fmaf(f, f, ln_s2[i]);
        }
        const float a = f - __uint_as_float(x[e] & 0xffff0000u);
        r1[e] = __float_as_uint(a);
        r2[e] = __float_as_uint(a - __uint_as_float(r1[e] & 0xffff0000u));
      }
      unsigned char* dst = sA + (arow + 32 * i) * kRowB + asw;
      *reinterpret_cast<u32x2*>(dst) = u32x2{pack_hi16(x[1], x[0]), pack_hi16(x[3], x[2])};
      *reinterpret_cast<u32x2*>(dst + TM * kRowB) = u32x2{pack_hi16(r1[1], r1[0]), pack_hi16(r1[3], r1[2])};
      *reinterpret_cast<u32x2*>(dst + 2 * TM * kRowB) = u32x2{pack_hi16(r2[1], r2[0]), pack_hi16(r2[3], r2[2])};
    }
  };
  const int frow = ln & 31, fsw = (frow >> 2) & 3;
  auto compute = [&](auto stage, int buf) {
    constexpr int P = decltype(stage)::value;
    const unsigned char* fa = s_a + buf * kBuf + frow * kRowB;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int off = ((kk * 2 + (ln >> 5)) ^ fsw) << 4;
      u32x4 a[SM][3];
#pragma unroll
      for (int i = 0; i < SM; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          a[i][p] = *reinterpret_cast<const u32x4*>(fa + p * TM * kRowB + i * 32 * kRowB + off);
      constexpr int pa[6] = {1, 0, 2, 0, 1, 0};
      constexpr int pb[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int i = 0; i < SM; ++i) acc[i] = mfma_bf16(a[i][pa[q]], wb[P][kk][pb[q]], acc[i]);
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  // the barrier of a step: the LDS writes of this wave have landed (the global requests for the next
  // step stay in flight across it)
  auto step_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };

  gload_a(0);
  gload_w(S0{}, 0);
  sstore(0);
  step_barrier();
  int s = 0;
  for (; s + 1 < nsteps; s += 2) {
    gload_a(s + 1);
    gload_w(S1{}, s + 1);
    compute(S0{}, 0);
    sstore(1);
    step_barrier();
    const bool more = s + 2 < nsteps;
    if (more) {
      gload_a(s + 2);
      gload_w(S0{}, s + 2);
    }
    compute(S1{}, 1);
    if (more) sstore(0);
    step_barrier();
  }
  if (s < nsteps) compute(S0{}, 0);

  float* s_stat = reinterpret_cast<float*>(s_a);  // [TM][2]
  if (LN) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      float a = ln_s1[i], b = ln_s2[i];
#pragma unroll
      for (int o = 1; o < 8; o <<= 1) {
        a += __shfl_xor(a, o, 64);
        b += __shfl_xor(b, o, 64);
      }
      if (aq == 0) {
        const float mean = a / (float)g.K;
        const float var = fmaxf(b / (float)g.K - mean * mean, 0.f);
        s_stat[(arow + 32 * i) * 2 + 0] = mean;
        s_stat[(arow + 32 * i) * 2 + 1] = 1.0f / sqrtf(var + g.ln_eps);
      }
    }
    __syncthreads();
  }

  const int li = ln & 31, lk = ln >> 5;
  const int64_t col = n0 + wv * 32 + li;
  if (col >= g.N) return;
  const float bv = g.bias ? g.bias[col] : 0.f;
  const float cs = LN ? g.ln_cs[col] : 0.f;
#pragma unroll
  for (int i = 0; i < SM; ++i) {
    float res[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int64_t row = min(m0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk, g.M - 1);
      res[e] = g.residual ? g.residual[row * g.ldc + col] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int trow = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
      const int64_t row = m0 + trow;
      if (row >= g.M) continue;
      float v = acc[i][e];
      if (LN) v = s_stat[trow * 2 + 1] * (v - s_stat[trow * 2] * cs);
      v += bv;
      if (g.act == 1) v = fmaxf(v, 0.f);
      if (g.act == 2) v = v / (1.0f + __expf(-v));
      if (g.act == 3) v = 1.0f / (1.0f + __expf(-v));
      if (g.act == 4) v = tanhf(v);
      if (g.act == 5) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
      g.C[row * g.ldc + col] = v * g.alpha + res[e];
    }
  }
}

// ------------------------------------------------------------------------------------------
// The channels-last convolution (implicit GEMM of conv.hip: rows = output pixels, K = taps x
// input channels, columns = output channels) on the same arithmetic and the same structure as
// gemm_split_bd_kernel: 64 pixels x 128 output channels per workgroup, the weight operands of a
// wave straight from the fragment image of w [Co, KH KW Ci] (a K step = 32 input channels of one
// tap), the pixels' channel runs gathered by the staging threads (8 lanes per 128-byte run), split
// and written to LDS as three planes.  Transposed convolutions keep conv_mfma_kernel's ordering of
// the rows by stride residue class (a tile iterates over its class's live taps only).
// ------------------------------------------------------------------------------------------
// WGN wavefronts side by side along the output channels, 4 / WGN along the pixels, each SM x 32
// pixels by 32 channels: <4, 2> = 64 x 128 (Co >= 128), <2, 2> = 128 x 64 (Co <= 64: no wave idles on
// columns past Co; the two waves of a column group fetch the same weight operands, the second from
// L1), <1, 1> = 128 x 32 (Co <= 32).
template <int WGN, int SM>
__global__ __launch_bounds__(256, 2) void conv_split_kernel(ConvArgs g, const void* planes) {
  constexpr int TM = (4 / WGN) * SM * 32, TN = WGN * 32;
  constexpr int kRowB = 64, kBuf = 3 * TM * kRowB, PA = TM / 32;
  __shared__ __attribute__((aligned(16))) unsigned char s_a[2 * kBuf];
  __shared__ int s_pix[TM];
  const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
  const int wm = wv / WGN, wn = wv % WGN;
  const int tiles_n = (g.Co + TN - 1) / TN;
  int64_t mt = blockIdx.x / tiles_n;
  const int n0 = (blockIdx.x % tiles_n) * TN;
  const int arow = tid >> 3, aq = tid & 7;
  const int asw = ((((aq >> 1) ^ ((arow >> 2) & 3)) << 4) | ((aq & 1) << 3));

  f32x16 acc[SM];
#pragma unroll
  for (int i = 0; i < SM; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

  int qh = 0, qw = 0, Hc = g.Ho, Wc = g.Wo, step_h = 1, step_w = 1;
  int64_t Mc = g.M;
  if (g.by_class) {
    step_h = g.sh, step_w = g.sw;
    for (int cls = 0; cls < g.sh * g.sw; ++cls) {
      qh = cls / g.sw, qw = cls - qh * g.sw;
      Mc = class_rows(g.N, g.Ho, g.Wo, g.sh, g.sw, qh, qw, Hc, Wc);
      const int64_t tc = (Mc + TM - 1) / TM;
      if (mt < tc) break;
      mt -= tc;
    }
  }
  const int64_t m0 = mt * TM;
  const int kh0 = g.by_class ? (qh + g.ph) % g.sh : 0, kw0 = g.by_class ? (qw + g.pw) % g.sw : 0;
  const int nkh = kh0 < g.KH ? (g.KH - kh0 + step_h - 1) / step_h : 0;
  const int nkw = kw0 < g.KW ? (g.KW - kw0 + step_w - 1) / step_w : 0;
  int rn[PA], rho[PA], rwo[PA];
  bool rvalid[PA];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int64_t m = m0 + arow + 32 * i;
    rvalid[i] = m < Mc;
    const int64_t mm = rvalid[i] ? m : 0;
    rwo[i] = qw + step_w * (int)(mm % Wc);
    rho[i] = qh + step_h * (int)((mm / Wc) % Hc);
    rn[i] = (int)(mm / ((int64_t)Wc * Hc));
    if (aq == 0) s_pix[arow + 32 * i] = rvalid[i] ? (rn[i] * g.Ho + rho[i]) * g.Wo + rwo[i] : -1;
  }
  const int chunks = g.Ci / 32;
  const int ntiles = nkh * nkw * chunks;
  const uint32_t x_bytes = (uint32_t)((int64_t)g.N * g.H * g.W * g.Ci * 4);
  auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.x), 0, x_bytes, 0x00020000);
  const int64_t groups = ((g.Co + 127) / 128) * 4;
  const int32_t wstep_bytes = (int32_t)(groups * 6144);
  const int ksteps = g.KH * g.KW * chunks;
  auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(planes), 0,
                                                  (uint32_t)(wstep_bytes * ksteps), 0x00020000);
  const int32_t vw = (int32_t)((n0 / 32 + wn) * 6144) + ln * 16;

  u32x4 ra[PA];
  u32x4 wb[2][2][3];
  // K tile t of this row tile: tap t / chunks of the tile's live taps, channels 32 (t % chunks) ..
  auto tap_of = [&](int t, int& kh, int& kw, int& c0) {
    const int tap = t / chunks;
    c0 = (t - tap * chunks) * 32;
    const int ih = tap / nkw;
    kh = kh0 + step_h * ih;
    kw = kw0 + step_w * (tap - ih * nkw);
  };
  auto gload_a = [&](int t) {
    int kh, kw, c0;
    tap_of(t, kh, kw, c0);
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      int hi, wi;
      const bool ok = tap_coord(rho[i], kh, g.sh, g.ph, g.H, g.transposed, hi) &
                      tap_coord(rwo[i], kw, g.sw, g.pw, g.W, g.transposed, wi) & rvalid[i];
      const uint32_t off = ok ? (uint32_t)((((int64_t)rn[i] * g.H + hi) * g.W + wi) * g.Ci + c0 + aq * 4) * 4u
                              : 0xfffffff0u;  // outside the buffer: reads zeros
      ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, off, 0, 0);
    }
  };
  auto gload_w = [&](auto stage, int t) {
    constexpr int P = decltype(stage)::value;
    int kh, kw, c0;
    tap_of(t, kh, kw, c0);
    const int32_t soff = ((kh * g.KW + kw) * chunks + c0 / 32) * wstep_bytes;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        wb[P][kk][p] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, vw, soff + (kk * 3 + p) * 1024, 0);
  };
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  auto sstore = [&](int buf) {
    unsigned char* sA = s_a + buf * kBuf;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const uint32_t x[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
      uint32_t r1[4], r2[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float f = __uint_as_float(x[e]);
        const float a = f - __uint_as_float(x[e] & 0xffff0000u);
        r1[e] = __float_as_uint(a);
        r2[e] = __float_as_uint(a - __uint_as_float(r1[e] & 0xffff0000u));
      }
      unsigned char* dst = sA + (arow + 32 * i) * kRowB + asw;
      *reinterpret_cast<u32x2*>(dst) = u32x2{pack_hi16(x[1], x[0]), pack_hi16(x[3], x[2])};
      *reinterpret_cast<u32x2*>(dst + TM * kRowB) = u32x2{pack_hi16(r1[1], r1[0]), pack_hi16(r1[3], r1[2])};
      *reinterpret_cast<u32x2*>(dst + 2 * TM * kRowB) = u32x2{pack_hi16(r2[1], r2[0]), pack_hi16(r2[3], r2[2])};
    }
  };
  const int frow = ln & 31, fsw = (frow >> 2) & 3;
  auto compute = [&](auto stage, int buf) {
    constexpr int P = decltype(stage)::value;
    const unsigned char* fa = s_a + buf * kBuf + (wm * SM * 32 + frow) * kRowB;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int off = ((kk * 2 + (ln >> 5)) ^ fsw) << 4;
      u32x4 a[SM][3];
#pragma unroll
      for (int i = 0; i < SM; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          a[i][p] = *reinterpret_cast<const u32x4*>(fa + p * TM * kRowB + i * 32 * kRowB + off);
      constexpr int pa[6] = {1, 0, 2, 0, 1, 0};
      constexpr int pb[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int i = 0; i < SM; ++i) acc[i] = mfma_bf16(a[i][pa[q]], wb[P][kk][pb[q]], acc[i]);
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  auto step_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  if (ntiles > 0) {  // (a class without live taps only gets the epilogue's shift)
    gload_a(0);
    gload_w(S0{}, 0);
    sstore(0);
    step_barrier();
    int s = 0;
    for (; s + 1 < ntiles; s += 2) {
      gload_a(s + 1);
      gload_w(S1{}, s + 1);
      compute(S0{}, 0);
      sstore(1);
      step_barrier();
      const bool more = s + 2 < ntiles;
      if (more) {
        gload_a(s + 2);
        gload_w(S0{}, s + 2);
      }
      compute(S1{}, 1);
      if (more) sstore(0);
      step_barrier();
    }
    if (s < ntiles) compute(S0{}, 0);
  } else {
    __syncthreads();  // s_pix
  }

  const int col = n0 + wn * 32 + (ln & 31);
  if (col >= g.Co) return;
  const float sc_ = g.scale ? g.scale[col] : 1.f, sh_ = g.shift ? g.shift[col] : 0.f;
#pragma unroll
  for (int i = 0; i < SM; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int64_t row = s_pix[wm * SM * 32 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (ln >> 5)];
      if (row < 0) continue;
      float v = conv_act(acc[i][e] * sc_ + sh_, g.act, g.slope);
      if (g.residual) v += g.residual[row * g.Co + col];
      g.y[row * g.Co + col] = v;
    }
}

// W [N, K] -> fragment-ordered image (layout 1): [K step][32-column group][MFMA K step 2][plane 3]
// [lane 64][8 bf16]; lane l of a group holds column 32 g + (l & 31), k = 32 s + 16 kk + 8 (l >> 5) ..
__global__ __launch_bounds__(256) void split_weight_frag_kernel(const float* __restrict__ W,
                                                               u32x4* __restrict__ planes, int64_t N,
                                                               int64_t K, int64_t ldw, int64_t groups,
                                                               int64_t ksteps) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (step, group, kk, lane)
  if (idx >= ksteps * groups * 128) return;
  const int lane = (int)(idx & 63), kk = (int)((idx >> 6) & 1);
  const int64_t grp = (idx >> 7) % groups, step = (idx >> 7) / groups;
  const int64_t row = grp * 32 + (lane & 31);
  uint32_t x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int64_t k = step * 32 + kk * 16 + (lane >> 5) * 8 + e;
    x[e] = (row < N && k < K) ? __float_as_uint(W[row * ldw + k]) : 0u;
  }
  const Planes8 p = split8(u32x4{x[0], x[1], x[2], x[3]}, u32x4{x[4], x[5], x[6], x[7]});
  u32x4* dst = planes + ((step * groups + grp) * 6 + kk * 3) * 64 + lane;
  dst[0] = p.h;
  dst[64] = p.m;
  dst[128] = p.l;
}

template <int TM, bool LN>
static int launch_split_bd(SplitGemmArgs g, hipStream_t st) {
  const int64_t tiles_m = (g.M + TM - 1) / TM, tiles_n = (g.N + 127) / 128;
  const int64_t total = tiles_m * tiles_n;
  if (total > 0x7fffffff) return APS_ERR_INVALID;
  g.tiles_n = (int32_t)tiles_n;
  g.remap = (total % 8 == 0) ? 1 : 0;
  hipLaunchKernelGGL((gemm_split_bd_kernel<TM, LN>), dim3((unsigned)total), dim3(256), 0, st, g);
  return aps_launch_status();
}

}  // namespace aps

using namespace aps;

extern "C" int64_t aps_linear_split_size(int64_t N, int64_t K) {
  if (N <= 0 || K <= 0) return 0;
  return ((N + 127) / 128) * 128 * ((K + 31) / 32) * 192;
}

extern "C" int aps_linear_split_weight(const float* W, void* planes, int64_t N, int64_t K,
                                       int64_t ldw, int32_t layout, void* stream) {
  APS_CHECK_ARG(W && planes && N > 0 && K > 0 && ldw >= K);
  APS_CHECK_ARG(((uintptr_t)planes & 15) == 0 && layout == 1);
  const int64_t np = ((N + 127) / 128) * 128, ksteps = (K + 31) / 32;
  if (np * ksteps * 192 >= ((int64_t)1 << 31)) return APS_ERR_UNSUPPORTED;
  const int64_t groups = np / 32, threads = ksteps * groups * 128;
  hipLaunchKernelGGL(split_weight_frag_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), W, reinterpret_cast<u32x4*>(planes), N, K, ldw,
                     groups, ksteps);
  return aps_launch_status();
}

extern "C" int aps_linear_split(const float* A, const void* planes, const float* bias,
                                const float* colsum, const float* residual, float* C, int64_t M,
                                int64_t N, int64_t K, int64_t lda, int64_t ldc, int32_t act,
                                float alpha, float eps, int32_t layout, void* stream) {
  APS_CHECK_ARG(A && planes && C && M > 0 && N > 0 && K > 0 && layout == 1);
  APS_CHECK_ARG(lda >= K && ldc >= N && lda % 4 == 0 && ((uintptr_t)A & 15) == 0 &&
                ((uintptr_t)planes & 15) == 0);
  APS_CHECK_ARG(act >= 0 && act <= 5);
  const int64_t ksteps = (K + 31) / 32;
  if (M * lda * 4 >= ((int64_t)1 << 31) || aps_linear_split_size(N, K) >= ((int64_t)1 << 31))
    return APS_ERR_UNSUPPORTED;
  SplitGemmArgs g{A, planes, bias, residual, C, M, N, K, lda, ldc, act, alpha, 0, 0, (int32_t)ksteps,
                  colsum, eps};
  hipStream_t st = static_cast<hipStream_t>(stream);
#ifdef APS_DEBUG_DISTURBANCE
  // experiments only (scripts/build_disturbance_libs.sh): the 32-row form that, as the other stream's
  // kernel, made packed-fp32 instructions of a co-resident STFT wavefront go wrong
  if (getenv("APS_SPLIT_TM") && atoi(getenv("APS_SPLIT_TM")) == 32)
    return colsum ? launch_split_bd<32, true>(g, st) : launch_split_bd<32, false>(g, st);
#endif
  return colsum ? launch_split_bd<64, true>(g, st) : launch_split_bd<64, false>(g, st);
}

extern "C" int aps_conv2d_nhwc_split(const float* x, const void* planes, const float* scale,
                                     const float* shift, const float* residual, float* y, int64_t N,
                                     int64_t H, int64_t W, int64_t Ci, int64_t Co, int64_t KH,
                                     int64_t KW, int64_t sh, int64_t sw, int64_t ph, int64_t pw,
                                     int64_t Ho, int64_t Wo, int32_t transposed, int32_t act,
                                     float slope, void* stream) {
  APS_CHECK_ARG(x && planes && y && N > 0 && H > 0 && W > 0 && Ci > 0 && Co > 0 && KH > 0 && KW > 0);
  APS_CHECK_ARG(sh > 0 && sw > 0 && ph >= 0 && pw >= 0 && Ho > 0 && Wo > 0);
  APS_CHECK_ARG(act == 0 || act == 1 || act == 5);
  APS_CHECK_ARG(Ci % 32 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)planes & 15) == 0);
  const int64_t M = N * Ho * Wo;
  if (N * H * W * Ci * 4 >= ((int64_t)1 << 32) - 64 || M >= ((int64_t)1 << 31) ||
      aps_linear_split_size(Co, KH * KW * Ci) >= ((int64_t)1 << 31))
    return APS_ERR_UNSUPPORTED;
  ConvArgs g{x, nullptr, scale, shift, residual, y, (int32_t)N, (int32_t)H, (int32_t)W, (int32_t)Ci,
             (int32_t)Ho, (int32_t)Wo, (int32_t)Co, (int32_t)KH, (int32_t)KW, (int32_t)sh,
             (int32_t)sw, (int32_t)ph, (int32_t)pw, transposed, act, slope, M, 0, 0};
  // tile shape by the number of output channels: 64 x 128, 128 x 64 (Co <= 64), 128 x 32 (Co <= 32)
  const int tn = Co > 64 ? 128 : (Co > 32 ? 64 : 32), tm = Co > 64 ? 64 : 128;
  int64_t tiles_m = (M + tm - 1) / tm;
  if (transposed && sh * sw > 1 && sh * sw <= 64) {
    g.by_class = 1;
    tiles_m = 0;
    for (int cls = 0; cls < sh * sw; ++cls) {
      int Hc, Wc;
      tiles_m += (class_rows(g.N, g.Ho, g.Wo, g.sh, g.sw, cls / g.sw, cls % g.sw, Hc, Wc) + tm - 1) / tm;
    }
  }
  const int64_t tiles = tiles_m * ((Co + tn - 1) / tn);
  if (tiles > 0x7fffffff) return APS_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (tn == 128)
    hipLaunchKernelGGL((conv_split_kernel<4, 2>), dim3((unsigned)tiles), dim3(256), 0, st, g, planes);
  else if (tn == 64)
    hipLaunchKernelGGL((conv_split_kernel<2, 2>), dim3((unsigned)tiles), dim3(256), 0, st, g, planes);
  else
    hipLaunchKernelGGL((conv_split_kernel<1, 1>), dim3((unsigned)tiles), dim3(256), 0, st, g, planes);
  return aps_launch_status();
}
