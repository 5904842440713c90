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
// Data.  W is split once per weight (aps_linear_split_weight) into a tile-interleaved image
//     planes[K step s][plane p][row n][32 k]  (bf16; rows padded to 128, K to 32 with zeros)
// so the 16 rows x 64 bytes one wave-wide request covers are 1 KB of contiguous memory (8 full
// cache lines; with the planes of a row side by side the same request touched 16 lines half used).  A stays fp32 in HBM and is split on its way
// from the staging registers into LDS (5.5 VALU operations per element, ~1/4 of the issue slots the
// MFMAs leave free).  LDS holds the three planes of both operands as [plane][row][32 k] with 64-byte
// rows and the 16-byte chunk index XOR-swizzled by (row >> 2) & 3: the 8-lane groups of the
// ds_write_b128 staging and the 16-lane groups of the ds_read_b128 operand fetches are both
// conflict free.
//
// Tile 128 x TN (TN = 128 / 64), four wavefronts 2 x 2, each 64 x TN/2 as 32 x 32 MFMA tiles; K
// consumed 32 at a time (two MFMA K steps); global loads register-staged one tile ahead; LDS single
// buffered, 2-3 workgroups per CU overlap each other's staging and MFMA phases.
#include <stdint.h>

#include <type_traits>

#include "common.h"
#include "conv_core.h"

// timing experiments only (scripts/micro/split_ablate.py): leave out parts of the steady-state loop
#ifndef APS_SPLIT_ABLATE
#define APS_SPLIT_ABLATE 0
#endif

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

template <int TN, bool LN>
__global__ __launch_bounds__(256, 2) void gemm_split_kernel(SplitGemmArgs g) {
  constexpr int TM = 128, WM = 64, WN = TN / 2, SM = 2, SN = WN / 32;
  constexpr int LA = TM / 64, LB = TN / 64;  // rows per staging thread
  constexpr int kRowB = 64;                  // bytes of one LDS row: 32 bf16
  extern __shared__ __attribute__((aligned(16))) unsigned char s_split[];
  unsigned char* sA = s_split;                    // [3][TM][64 B]
  unsigned char* sB = s_split + 3 * TM * kRowB;   // [3][TN][64 B]
  const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
  const int wm = wv >> 1, wn = wv & 1;
  int64_t lin = blockIdx.x;
  if (g.remap) {
    const int64_t per = gridDim.x / 8;
    lin = (lin & 7) * per + (lin >> 3);
  }
  const int64_t m0 = (lin / g.tiles_n) * TM, n0 = (lin % g.tiles_n) * TN;
  // staging roles.  A: 8 lanes cover the 128 bytes a row contributes to a K step (a wave-wide request
  // = 8 full cache lines); thread (arow + 32 i, aq) stages 4 floats -> 4 bf16 = 8 bytes per plane.
  // B planes: 16-byte chunk sc of rows srow + 64 i (16 consecutive rows per request = 1 KB).
  const int srow = tid >> 2, sc = tid & 3;
  const int ssw = ((sc ^ ((srow >> 2) & 3)) << 4);  // swizzled chunk offset inside the LDS row
  const int arow = tid >> 3, aq = tid & 7;
  const int asw = ((((aq >> 1) ^ ((arow >> 2) & 3)) << 4) | ((aq & 1) << 3));
  constexpr int PA = TM / 32;

  f32x16 acc[SM][SN];
#pragma unroll
  for (int i = 0; i < SM; ++i)
#pragma unroll
    for (int j = 0; j < SN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  auto rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0,
                                                  (uint32_t)(g.M * g.lda * 4), 0x00020000);
  const int64_t np = ((g.N + 127) / 128) * 128;
  const int32_t wplane_bytes = (int32_t)(np * 64), wstep_bytes = 3 * wplane_bytes;
  auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.Wp), 0,
                                                  (uint32_t)(wstep_bytes * g.ksteps), 0x00020000);
  int32_t va[PA], vb[LB];
#pragma unroll
  for (int i = 0; i < PA; ++i)
    va[i] = (int32_t)(min(m0 + arow + 32 * i, g.M - 1) * g.lda * 4) + aq * 16;
#pragma unroll
  for (int i = 0; i < LB; ++i) vb[i] = (int32_t)((n0 + srow + 64 * i) * 64) + sc * 16;

  u32x4 ra[PA], rb[LB][3];
  const int nsteps = g.ksteps;
  const bool ragged = (g.K & 31) != 0;
  // K steps are walked from a per-row-panel start (wrapping around): workgroups of different row
  // panels then read different 128-byte columns of A at any one time.  A is [M, K] row major, so
  // the rows of one column sit at a 4 K byte pitch -- all CUs on the same column would queue on the
  // few L2 channels that pitch maps to.
  const int rot = (int)((lin / g.tiles_n) % nsteps);
  auto gload = [&](int s) {
    const int step = (s + rot) % nsteps;
    const int32_t soff_a = step * 128;
    if (ragged && step == nsteps - 1) {
      // K remainder: float4 requests past K are dropped (lda is a multiple of 4 >= K) and the
      // components with k >= K zeroed
      const int64_t kk = (int64_t)step * 32 + aq * 4;  // kk < K implies kk + 4 <= lda
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        u32x4 v = u32x4{0u, 0u, 0u, 0u};
        if (kk < g.K) v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, va[i], soff_a, 0);
        v.x = (kk + 0 < g.K) ? v.x : 0u;
        v.y = (kk + 1 < g.K) ? v.y : 0u;
        v.z = (kk + 2 < g.K) ? v.z : 0u;
        v.w = (kk + 3 < g.K) ? v.w : 0u;
        ra[i] = v;
      }
    } else {
#pragma unroll
      for (int i = 0; i < PA; ++i) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, va[i], soff_a, 0);
    }
    const int32_t soff_w = step * wstep_bytes;
#pragma unroll
    for (int i = 0; i < LB; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        rb[i][p] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, vb[i], soff_w + p * wplane_bytes, 0);
  };
  float ln_s1[PA], ln_s2[PA];
#pragma unroll
  for (int i = 0; i < PA; ++i) ln_s1[i] = ln_s2[i] = 0.f;
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  auto sstore = [&]() {
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
#pragma unroll
    for (int i = 0; i < LB; ++i) {
      unsigned char* dst = sB + (srow + 64 * i) * kRowB + ssw;
#pragma unroll
      for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(dst + p * TN * kRowB) = rb[i][p];
    }
  };
  // operand fetch: lane ln holds row (ln & 31), k = 8 (ln >> 5) .. + 7 of an MFMA K step
  const int frow = ln & 31, fsw = (frow >> 2) & 3;
  const unsigned char* fa = sA + (wm * WM + frow) * kRowB;
  const unsigned char* fb = sB + (wn * WN + frow) * kRowB;
  auto compute = [&]() {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int off = ((kk * 2 + (ln >> 5)) ^ fsw) << 4;
      u32x4 a[SM][3], b[SN][3];
#pragma unroll
      for (int i = 0; i < SM; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          a[i][p] = *reinterpret_cast<const u32x4*>(fa + p * TM * kRowB + i * 32 * kRowB + off);
#pragma unroll
      for (int j = 0; j < SN; ++j)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          b[j][p] = *reinterpret_cast<const u32x4*>(fb + p * TN * kRowB + j * 32 * kRowB + off);
      // smallest terms first; consecutive MFMAs alternate accumulators
      constexpr int pa[6] = {1, 0, 2, 0, 1, 0};
      constexpr int pb[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int i = 0; i < SM; ++i)
#pragma unroll
          for (int j = 0; j < SN; ++j) acc[i][j] = mfma_bf16(a[i][pa[q]], b[j][pb[q]], acc[i][j]);
    }
  };

  gload(0);
  sstore();
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    const bool more = s + 1 < nsteps;
    if (more) gload(s + 1);
    compute();
    __syncthreads();
    if (more) {
      sstore();
      __syncthreads();
    }
  }

  float* s_stat = reinterpret_cast<float*>(s_split);  // [TM][2]
  if (LN) {
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

  // epilogue: C/D layout col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
  const int li = ln & 31, lk = ln >> 5;
#pragma unroll
  for (int i = 0; i < SM; ++i)
#pragma unroll
    for (int j = 0; j < SN; ++j) {
      const int64_t col = n0 + wn * WN + j * 32 + li;
      if (col >= g.N) continue;
      const float bv = g.bias ? g.bias[col] : 0.f;
      const float cs = LN ? g.ln_cs[col] : 0.f;
      float res[16];  // all 16 requests of the tile in flight before the first is used
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t row = min(m0 + wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk, g.M - 1);
        res[e] = g.residual ? g.residual[row * g.ldc + col] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int trow = wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
        const int64_t row = m0 + trow;
        if (row >= g.M) continue;
        float v = acc[i][j][e];
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
// Software-pipelined form: LDS double buffered, ONE workgroup barrier per K step, and every other
// instruction of a step -- the 2 LA + 3 LB global requests of tile s + 2, the operand fetches of the
// second MFMA K step, the split (VALU) and the LDS writes of tile s + 1, the operand fetches of the
// first K step of tile s + 1 -- placed by hand behind one of the step's 12 SM SN MFMAs, so that its
// issue cycles fall into the 32-cycle shadow of that MFMA (sched_barrier(0) pins the order; the
// same technique as gemm_f32_kernel<SWP>).  A single wavefront per SIMD then keeps the matrix pipe
// fed on its own: the 128 x 128 tile (96 KB of LDS, one workgroup per CU) needs no co-resident
// workgroup to hide its staging.
//   step s:   phase A  MFMAs on X (K step 0 of tile s)   | requests for tile s + 2 -> stage s & 1
//                                                         | fetch Y (K step 1 of tile s)
//                                                         | split + write rows of tile s + 1
//             phase B  MFMAs on Y                         | remaining writes of tile s + 1
//                                                         | barrier (writes landed; buffer s & 1 is
//                                                         |   no longer read by anyone)
//                                                         | fetch X of tile s + 1
// ------------------------------------------------------------------------------------------
template <int TN, bool LN, bool RAGGED>
__global__ __launch_bounds__(256, (TN == 128 ? 1 : 2)) void gemm_split_swp_kernel(SplitGemmArgs g) {
  constexpr int TM = 128, WM = 64, WN = TN / 2, SM = 2, SN = WN / 32;
  constexpr int LA = TM / 64, LB = TN / 64;
  constexpr int kRowB = 64;
  constexpr int kBuf = 3 * (TM + TN) * kRowB;  // bytes of one LDS buffer
  constexpr int NT = 6 * SM * SN;              // MFMAs (= slots) per phase
  extern __shared__ __attribute__((aligned(16))) unsigned char s_split[];
  const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
  const int wm = wv >> 1, wn = wv & 1;
  int64_t lin = blockIdx.x;
  if (g.remap) {
    const int64_t per = gridDim.x / 8;
    lin = (lin & 7) * per + (lin >> 3);
  }
  const int64_t m0 = (lin / g.tiles_n) * TM, n0 = (lin % g.tiles_n) * TN;
  const int srow = tid >> 2, sc = tid & 3;
  const int ssw = ((sc ^ ((srow >> 2) & 3)) << 4);

  f32x16 acc[SM][SN];
#pragma unroll
  for (int i = 0; i < SM; ++i)
#pragma unroll
    for (int j = 0; j < SN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  auto rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0,
                                                  (uint32_t)(g.M * g.lda * 4), 0x00020000);
  const int64_t np = ((g.N + 127) / 128) * 128;
  const int32_t wplane_bytes = (int32_t)(np * 64), wstep_bytes = 3 * wplane_bytes;
  auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.Wp), 0,
                                                  (uint32_t)(wstep_bytes * g.ksteps), 0x00020000);
  int32_t va[LA], vb[LB];
#pragma unroll
  for (int i = 0; i < LA; ++i)
    va[i] = (int32_t)(min(m0 + srow + 64 * i, g.M - 1) * g.lda * 4) + sc * 32;
#pragma unroll
  for (int i = 0; i < LB; ++i) vb[i] = (int32_t)((n0 + srow + 64 * i) * 64) + sc * 16;

  const int nsteps = g.ksteps;
  constexpr bool ragged = RAGGED;  // K % 32 != 0: the last tile is masked in the staging registers
  u32x4 ra[2][LA][2], rb[2][LB][3];  // two register stages: tile t travels through stage t & 1
  constexpr int NL = 2 * LA + 3 * LB;
  // request q of tile `step` (clamped to the last tile: the extra requests are never consumed)
  // K steps are walked from a per-row-panel start, wrapping around (see gemm_split_kernel)
  const int rot = (int)((lin / g.tiles_n) % nsteps);
  auto tile_of = [&](int step) { return (min(step, nsteps - 1) + rot) % nsteps; };
  auto load1 = [&](auto stage, int q, int step) {
    constexpr int P = decltype(stage)::value;
    const int st = tile_of(step);
    if (q < 2 * LA)
      ra[P][q >> 1][q & 1] =
          __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, va[q >> 1] + 16 * (q & 1), st * 128, 0);
    else
      rb[P][(q - 2 * LA) / 3][(q - 2 * LA) % 3] = __builtin_amdgcn_raw_buffer_load_b128(
          rsrc_w, vb[(q - 2 * LA) / 3], st * wstep_bytes + ((q - 2 * LA) % 3) * wplane_bytes, 0);
  };
  // K remainder: components with k >= K of the last tile are zeroed in the staging registers (the
  // requests themselves may run into the next row / past the buffer: masked, or 0 by range check)
  auto mask_stage = [&](auto stage) {
    constexpr int P = decltype(stage)::value;
    const int64_t k = (int64_t)(nsteps - 1) * 32 + sc * 8;
#pragma unroll
    for (int i = 0; i < LA; ++i)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        u32x4 v = ra[P][i][h];
        const int64_t kk = k + 4 * h;
        v.x = (kk + 0 < g.K) ? v.x : 0u;
        v.y = (kk + 1 < g.K) ? v.y : 0u;
        v.z = (kk + 2 < g.K) ? v.z : 0u;
        v.w = (kk + 3 < g.K) ? v.w : 0u;
        ra[P][i][h] = v;
      }
  };
  float ln_s1[LA], ln_s2[LA];
#pragma unroll
  for (int i = 0; i < LA; ++i) ln_s1[i] = ln_s2[i] = 0.f;
  u32x4 ph[LA], pm[LA], pl[LA];  // planes of the row chunks being split
  uint32_t ev_x, ev_a, ev_b;     // the even element of a pair waits here for its odd partner
  // element e (0-7) of row chunk i: 4 VALU operations for the two residuals; the odd element of a
  // pair also packs dword e / 2 of the three planes (3 v_perm_b32)
  auto elem = [&](auto stage, int i, int e, float keep) {
    constexpr int P = decltype(stage)::value;
    const u32x4 v = ra[P][i][e >> 2];
    const uint32_t x = ((e & 3) == 0) ? v.x : ((e & 3) == 1) ? v.y : ((e & 3) == 2) ? v.z : v.w;
    const float f = __uint_as_float(x);
    if (LN) {
      ln_s1[i] = fmaf(keep, f, ln_s1[i]);
      ln_s2[i] = fmaf(keep * f, f, ln_s2[i]);
    }
    const float a = f - __uint_as_float(x & 0xffff0000u);
    const float b = a - __uint_as_float(__float_as_uint(a) & 0xffff0000u);
    if ((e & 1) == 0) {
      ev_x = x, ev_a = __float_as_uint(a), ev_b = __float_as_uint(b);
    } else {
      ph[i][e >> 1] = pack_hi16(x, ev_x);
      pm[i][e >> 1] = pack_hi16(__float_as_uint(a), ev_a);
      pl[i][e >> 1] = pack_hi16(__float_as_uint(b), ev_b);
    }
  };
  // LDS write w of a tile into buffer `buf`: w < 3 LA: plane w % 3 of A row chunk w / 3, then the B planes
  auto write1 = [&](auto stage, int w, int buf) {
    constexpr int P = decltype(stage)::value;
    unsigned char* base = s_split + buf * kBuf;
    if (w < 3 * LA) {
      const int i = w / 3, p = w % 3;
      unsigned char* dst = base + p * TM * kRowB + (srow + 64 * i) * kRowB + ssw;
      *reinterpret_cast<u32x4*>(dst) = (p == 0) ? ph[i] : (p == 1) ? pm[i] : pl[i];
    } else {
      const int i = (w - 3 * LA) / 3, p = (w - 3 * LA) % 3;
      unsigned char* dst = base + 3 * TM * kRowB + p * TN * kRowB + (srow + 64 * i) * kRowB + ssw;
      *reinterpret_cast<u32x4*>(dst) = rb[P][i][p];
    }
  };
  const int frow = ln & 31, fsw = (frow >> 2) & 3;
  const int foff0 = ((0 + (ln >> 5)) ^ fsw) << 4, foff1 = ((2 + (ln >> 5)) ^ fsw) << 4;
  const unsigned char* fa = s_split + (wm * WM + frow) * kRowB;
  const unsigned char* fb = s_split + 3 * TM * kRowB + (wn * WN + frow) * kRowB;
  u32x4 xa[SM][3], xb[SN][3], ya[SM][3], yb[SN][3];
  constexpr int NR = 3 * (SM + SN);
  // operand fetch r of K step kk of buffer buf, in the order the MFMAs need the planes:
  // (m, m) first, then (h, l), (l, h): A m, B m, A h, B l, A l, B h
  auto read1 = [&](u32x4 (&oa)[SM][3], u32x4 (&ob)[SN][3], int r, int buf, int kk) {
    constexpr int planeA[3] = {1, 0, 2}, planeB[3] = {1, 2, 0};
    const int grp = r / (SM + SN), w = r % (SM + SN);
    const int off = buf * kBuf + (kk ? foff1 : foff0);
    if (w < SM) {
      const int p = planeA[grp];
      oa[w][p] = *reinterpret_cast<const u32x4*>(fa + off + p * TM * kRowB + w * 32 * kRowB);
    } else {
      const int p = planeB[grp], j = w - SM;
      ob[j][p] = *reinterpret_cast<const u32x4*>(fb + off + p * TN * kRowB + j * 32 * kRowB);
    }
  };
  auto mfma1 = [&](const u32x4 (&oa)[SM][3], const u32x4 (&ob)[SN][3], int t) {
    constexpr int pa[6] = {1, 0, 2, 0, 1, 0};
    constexpr int pb[6] = {1, 2, 0, 1, 0, 0};
    const int q = t / (SM * SN), i = (t % (SM * SN)) / SN, j = t % SN;
    acc[i][j] = mfma_bf16(oa[i][pa[q]], ob[j][pb[q]], acc[i][j]);
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  // Item lists.  Chunk i of the A rows = 8 elements + 3 plane writes (11 items).
  // phase A of step s (tile in buffer cur, X fetched): `stage` holds tile s + 1 (-> buffer nxt),
  // the requests of tile s + 2 go to `other`.  Slots 0.. carry one request and one Y fetch each,
  // the chunks 0 .. LA - 2 follow behind them (from slot 0 when there is no room behind).
  constexpr int NIA = 11 * (LA - 1);
  constexpr int SA0 = (NT - NR >= NIA) ? NR : 0;
  auto chunk_item = [&](auto stage, int i, int w, float keep, int nxt) {
    if (w < 8) {
      if (!(APS_SPLIT_ABLATE & 4)) elem(stage, i, w, keep);
    } else if (!(APS_SPLIT_ABLATE & 2)) {
      write1(stage, 3 * i + (w - 8), nxt);
    }
  };
  constexpr int ABL = APS_SPLIT_ABLATE;
  auto phase_a = [&](auto stage, auto other, int cur, int nxt, float keep, int step) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (!(ABL & 32)) mfma1(xa, xb, t);
      if (t < NL && !(ABL & 1)) load1(other, t, step + 2);
      if (t < NR && !(ABL & 16)) read1(ya, yb, t, cur, 1);
#pragma unroll
      for (int k = 0; k < NIA; ++k)
        if (SA0 + (k * (NT - SA0)) / NIA == t) chunk_item(stage, k / 11, k % 11, keep, nxt);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // phase B: the last chunk and the B plane writes, the barrier two slots behind the last write,
  // then the X fetches of the next tile (two per slot)
  constexpr int NIB = 11 + 3 * LB;
  constexpr int NB = NT - (NR + 1) / 2 - 2;  // slots in front of the barrier
  auto phase_b = [&](auto stage, float keep, int nxt) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (!(ABL & 32)) mfma1(ya, yb, t);
#pragma unroll
      for (int k = 0; k < NIB; ++k) {
        if ((k * NB) / NIB != t) continue;
        if (k < 11) chunk_item(stage, LA - 1, k, keep, nxt);
        else if (!(ABL & 2)) write1(stage, 3 * LA + (k - 11), nxt);
      }
      if (t == NB + 1 && !(ABL & 8)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
#pragma unroll
      for (int r = 0; r < NR; ++r)
        if (NB + 2 + (r * (NT - NB - 2)) / NR == t && !(ABL & 16)) read1(xa, xb, r, nxt, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // prologue: tiles 0 and 1 requested, tile 0 split into buffer 0, X of tile 0 fetched
#pragma unroll
  for (int q = 0; q < NL; ++q) load1(S0{}, q, 0);
#pragma unroll
  for (int q = 0; q < NL; ++q) load1(S1{}, q, 1);
  if (ragged && tile_of(0) == nsteps - 1) mask_stage(S0{});
#pragma unroll
  for (int i = 0; i < LA; ++i) {
#pragma unroll
    for (int e = 0; e < 8; ++e) elem(S0{}, i, e, 1.0f);
#pragma unroll
    for (int p = 0; p < 3; ++p) write1(S0{}, 3 * i + p, 0);
  }
#pragma unroll
  for (int w = 3 * LA; w < 3 * LA + 3 * LB; ++w) write1(S0{}, w, 0);
  __syncthreads();
#pragma unroll
  for (int r = 0; r < NR; ++r) read1(xa, xb, r, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  int s = 0;
  for (; s + 1 < nsteps; s += 2) {
    // step s: tile s in buffer 0, tile s + 1 in stage 1 -> buffer 1, requests of tile s + 2 -> stage 0
    if (ragged && tile_of(s + 1) == nsteps - 1) mask_stage(S1{});
    phase_a(S1{}, S0{}, 0, 1, 1.0f, s);
    phase_b(S1{}, 1.0f, 1);
    // step s + 1: tile s + 1 in buffer 1, tile s + 2 in stage 0 -> buffer 0 (an unused repeat of
    // the last tile when s + 2 = nsteps)
    if (ragged && s + 2 < nsteps && tile_of(s + 2) == nsteps - 1) mask_stage(S0{});
    const float keep = (s + 2 < nsteps) ? 1.0f : 0.0f;
    phase_a(S0{}, S1{}, 1, 0, keep, s + 1);
    phase_b(S0{}, keep, 0);
  }
  if (s < nsteps) {  // last tile of an odd count (buffer 0); what it stages is never used
    phase_a(S1{}, S0{}, 0, 1, 0.0f, s);
    phase_b(S1{}, 0.0f, 1);
  }
  __syncthreads();  // the LayerNorm statistics reuse the buffers

  float* s_stat = reinterpret_cast<float*>(s_split);  // [TM][2]
  if (LN) {
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      float a = ln_s1[i], b = ln_s2[i];
      a += __shfl_xor(a, 1, 64);
      b += __shfl_xor(b, 1, 64);
      a += __shfl_xor(a, 2, 64);
      b += __shfl_xor(b, 2, 64);
      if (sc == 0) {
        const float mean = a / (float)g.K;
        const float var = fmaxf(b / (float)g.K - mean * mean, 0.f);
        s_stat[(srow + 64 * i) * 2 + 0] = mean;
        s_stat[(srow + 64 * i) * 2 + 1] = 1.0f / sqrtf(var + g.ln_eps);
      }
    }
    __syncthreads();
  }

  const int li = ln & 31, lk = ln >> 5;
#pragma unroll
  for (int i = 0; i < SM; ++i)
#pragma unroll
    for (int j = 0; j < SN; ++j) {
      const int64_t col = n0 + wn * WN + j * 32 + li;
      if (col >= g.N) continue;
      const float bv = g.bias ? g.bias[col] : 0.f;
      const float cs = LN ? g.ln_cs[col] : 0.f;
      float res[16];  // all 16 requests of the tile in flight before the first is used
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t row = min(m0 + wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk, g.M - 1);
        res[e] = g.residual ? g.residual[row * g.ldc + col] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int trow = wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
        const int64_t row = m0 + trow;
        if (row >= g.M) continue;
        float v = acc[i][j][e];
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
// Producer / consumer form (128 x 128 tile, 512 threads): wavefronts 0-3 only feed the matrix pipe
// (operand fetches + MFMAs, a 2 x 2 grid of 64 x 64 tiles), wavefronts 4-7 only stage (global
// requests, the bf16 split, LDS writes); every SIMD hosts one of each, so the staging instructions
// issue from another wave while the consumer's MFMAs execute -- a single wave cannot hide them (a
// ds_write_b128 / buffer_load holds its wave's issue for 12-28 cycles of the 32 an MFMA lasts;
// measured in scripts/micro/split_ablate.py: every part of the staging lengthened the one-wave loop).
// LDS is a ring of three tile buffers and there is ONE workgroup barrier per K step:
//   slot t (between barriers t - 1 and t):  consumers: MFMAs of tile t; fetch K step 1 of tile t
//                                                      (buffer t % 3) and K step 0 of tile t + 1
//                                           producers: split + write tile t + 2 (buffer (t + 2) % 3),
//                                                      request tile t + 4
// Tile t + 1 was completed in slot t - 1 and the buffer written in slot t + 1 is the one the
// consumers stopped reading in slot t: no second barrier, no fetch in front of an MFMA.
// ------------------------------------------------------------------------------------------
template <bool LN, bool RAGGED>
__global__ __launch_bounds__(512, 2) void gemm_split_pc_kernel(SplitGemmArgs g) {
  constexpr int TM = 128, TN = 128, WM = 64, WN = 64, SM = 2, SN = 2;
  constexpr int LA = 2, LB = 2;
  constexpr int kRowB = 64;
  constexpr int kBuf = 3 * (TM + TN) * kRowB;  // 48 KB per ring buffer
  constexpr int NT = 6 * SM * SN;
  extern __shared__ __attribute__((aligned(16))) unsigned char s_split[];
  const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
  int64_t lin = blockIdx.x;
  if (g.remap) {
    const int64_t per = gridDim.x / 8;
    lin = (lin & 7) * per + (lin >> 3);
  }
  const int64_t m0 = (lin / g.tiles_n) * TM, n0 = (lin % g.tiles_n) * TN;
  const int nsteps = g.ksteps;
  float* s_stat = reinterpret_cast<float*>(s_split);  // [TM][2] after the loop
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  if (wv >= 4) {
    // ---------------------------------------------------------------- producers
    // A: 8 lanes cover the 128 bytes one row contributes to a K step (a wave-wide request = 8 full
    // cache lines); thread (arow + 32 i, aq) stages 4 floats -> 4 bf16 = 8 bytes per plane.
    // B planes: 4 lanes cover the 64 bytes of a row, 16 consecutive rows per request = 1 KB.
    const int ptid = tid - 256;
    const int arow = ptid >> 3, aq = ptid & 7;
    const int brow = ptid >> 2, bc = ptid & 3;
    constexpr int PA = TM / 32, PB = TN / 64;  // passes over the rows
    const int asw = ((((aq >> 1) ^ ((arow >> 2) & 3)) << 4) | ((aq & 1) << 3));
    const int bsw = ((bc ^ ((brow >> 2) & 3)) << 4);
    auto rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0,
                                                    (uint32_t)(g.M * g.lda * 4), 0x00020000);
    const int64_t np = ((g.N + 127) / 128) * 128;
    const int32_t wplane_bytes = (int32_t)(np * 64), wstep_bytes = 3 * wplane_bytes;
    auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.Wp), 0,
                                                    (uint32_t)(wstep_bytes * g.ksteps), 0x00020000);
    int32_t va[PA], vb[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i)
      va[i] = (int32_t)(min(m0 + arow + 32 * i, g.M - 1) * g.lda * 4) + aq * 16;
#pragma unroll
    for (int i = 0; i < PB; ++i) vb[i] = (int32_t)((n0 + brow + 64 * i) * 64) + bc * 16;
    const int rot = (int)((lin / g.tiles_n) % nsteps);
    auto tile_of = [&](int step) { return (min(step, nsteps - 1) + rot) % nsteps; };
    u32x4 ra[2][PA], rb[2][PB][3];
    auto request = [&](auto stage, int step) {
      constexpr int P = decltype(stage)::value;
      const int st = tile_of(step);
#pragma unroll
      for (int i = 0; i < PA; ++i)
        ra[P][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, va[i], st * 128, 0);
#pragma unroll
      for (int i = 0; i < PB; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          rb[P][i][p] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, vb[i],
                                                              st * wstep_bytes + p * wplane_bytes, 0);
    };
    float ln_s1[PA], ln_s2[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) ln_s1[i] = ln_s2[i] = 0.f;
    // split the staged tile `step` into ring buffer `buf`
    auto produce = [&](auto stage, int step, int buf) {
      constexpr int P = decltype(stage)::value;
      unsigned char* base = s_split + buf * kBuf;
      const bool last = RAGGED && tile_of(step) == nsteps - 1;
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        u32x4 v = ra[P][i];
        if (last) {  // K remainder: components with k >= K are zeroed
          const int64_t k = (int64_t)(nsteps - 1) * 32 + aq * 4;
          v.x = (k + 0 < g.K) ? v.x : 0u;
          v.y = (k + 1 < g.K) ? v.y : 0u;
          v.z = (k + 2 < g.K) ? v.z : 0u;
          v.w = (k + 3 < g.K) ? v.w : 0u;
        }
        const uint32_t x[4] = {v.x, v.y, v.z, v.w};
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
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        unsigned char* dst = base + (arow + 32 * i) * kRowB + asw;
        *reinterpret_cast<u32x2*>(dst) = u32x2{pack_hi16(x[1], x[0]), pack_hi16(x[3], x[2])};
        *reinterpret_cast<u32x2*>(dst + TM * kRowB) = u32x2{pack_hi16(r1[1], r1[0]), pack_hi16(r1[3], r1[2])};
        *reinterpret_cast<u32x2*>(dst + 2 * TM * kRowB) =
            u32x2{pack_hi16(r2[1], r2[0]), pack_hi16(r2[3], r2[2])};
      }
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        unsigned char* dst = base + 3 * TM * kRowB + (brow + 64 * i) * kRowB + bsw;
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(dst + p * TN * kRowB) = rb[P][i][p];
      }
    };
    // barrier of a producer: its LDS writes have landed; the global requests stay in flight
    auto publish = [&]() {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    };
    request(S0{}, 0);
    request(S1{}, 1);
    produce(S0{}, 0, 0);
    request(S0{}, 2);
    if (nsteps > 1) produce(S1{}, 1, 1);
    request(S1{}, 3);
    publish();
    // slot t: tile t + 2 (stage t & 1) -> buffer (t + 2) % 3, then the requests of tile t + 4
    int buf = 2;
    int t = 0;
    constexpr int ABL = APS_SPLIT_ABLATE;
    for (; t + 1 < nsteps; t += 2) {
      if (t + 2 < nsteps && !(ABL & 2)) produce(S0{}, t + 2, buf);
      if (!(ABL & 1)) request(S0{}, t + 4);
      if (!(ABL & 8)) publish();
      buf = (buf == 2) ? 0 : buf + 1;
      if (t + 3 < nsteps && !(ABL & 2)) produce(S1{}, t + 3, buf);
      if (!(ABL & 1)) request(S1{}, t + 5);
      if (!(ABL & 8)) publish();
      buf = (buf == 2) ? 0 : buf + 1;
    }
    if (t < nsteps && !(ABL & 8)) publish();  // odd count: the consumers' last slot
    __builtin_amdgcn_s_barrier();  // the consumers are done with the ring
    if (LN) {
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
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    return;
  }

  // ------------------------------------------------------------------ consumers
  const int wm = wv >> 1, wn = wv & 1;
  f32x16 acc[SM][SN];
#pragma unroll
  for (int i = 0; i < SM; ++i)
#pragma unroll
    for (int j = 0; j < SN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int frow = ln & 31, fsw = (frow >> 2) & 3;
  const int foff0 = ((0 + (ln >> 5)) ^ fsw) << 4, foff1 = ((2 + (ln >> 5)) ^ fsw) << 4;
  const unsigned char* fa = s_split + (wm * WM + frow) * kRowB;
  const unsigned char* fb = s_split + 3 * TM * kRowB + (wn * WN + frow) * kRowB;
  u32x4 xa[SM][3], xb[SN][3], ya[SM][3], yb[SN][3];
  constexpr int NR = 3 * (SM + SN);
  auto read1 = [&](u32x4 (&oa)[SM][3], u32x4 (&ob)[SN][3], int r, int boff, int kk) {
    constexpr int planeA[3] = {1, 0, 2}, planeB[3] = {1, 2, 0};
    const int grp = r / (SM + SN), w = r % (SM + SN);
    const int off = boff + (kk ? foff1 : foff0);
    if (w < SM) {
      const int p = planeA[grp];
      oa[w][p] = *reinterpret_cast<const u32x4*>(fa + off + p * TM * kRowB + w * 32 * kRowB);
    } else {
      const int p = planeB[grp], j = w - SM;
      ob[j][p] = *reinterpret_cast<const u32x4*>(fb + off + p * TN * kRowB + j * 32 * kRowB);
    }
  };
  auto mfma1 = [&](const u32x4 (&oa)[SM][3], const u32x4 (&ob)[SN][3], int t) {
    constexpr int pa[6] = {1, 0, 2, 0, 1, 0};
    constexpr int pb[6] = {1, 2, 0, 1, 0, 0};
    const int q = t / (SM * SN), i = (t % (SM * SN)) / SN, j = t % SN;
    acc[i][j] = mfma_bf16(oa[i][pa[q]], ob[j][pb[q]], acc[i][j]);
  };
  // epilogue operands requested ahead of the loop
  const int li = ln & 31, lk = ln >> 5;
  float e_bias[SN], e_cs[SN];
#pragma unroll
  for (int j = 0; j < SN; ++j) {
    const int64_t col = min(n0 + wn * WN + j * 32 + li, g.N - 1);
    e_bias[j] = g.bias ? g.bias[col] : 0.f;
    e_cs[j] = LN ? g.ln_cs[col] : 0.f;
  }

  __builtin_amdgcn_s_barrier();  // tiles 0 and 1 are in the ring
#pragma unroll
  for (int r = 0; r < NR; ++r) read1(xa, xb, r, 0, 0);
  int cur = 0;  // byte offset of the ring buffer of tile t
  for (int t = 0; t < nsteps; ++t) {
    const int nxt = (cur == 2 * kBuf) ? 0 : cur + kBuf;
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      if (!(APS_SPLIT_ABLATE & 32)) mfma1(xa, xb, u);
      if (u < NR && !(APS_SPLIT_ABLATE & 16)) read1(ya, yb, u, cur, 1);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      if (!(APS_SPLIT_ABLATE & 32)) mfma1(ya, yb, u);
      // one fetch behind each of the first MFMAs: all have landed when the loop comes round (the
      // compiler waits for every outstanding fetch there).  (An unused fetch after the last tile.)
      if (u < NR && !(APS_SPLIT_ABLATE & 16)) read1(xa, xb, u, nxt, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!(APS_SPLIT_ABLATE & 8)) __builtin_amdgcn_s_barrier();
    cur = nxt;
  }
  __builtin_amdgcn_s_barrier();  // every consumer is done with the ring
  if (LN) __builtin_amdgcn_s_barrier();  // the producers have published the row statistics

#pragma unroll
  for (int i = 0; i < SM; ++i)
#pragma unroll
    for (int j = 0; j < SN; ++j) {
      const int64_t col = n0 + wn * WN + j * 32 + li;
      if (col >= g.N) continue;
      const float bv = e_bias[j], cs = e_cs[j];
      float res[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t row = min(m0 + wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk, g.M - 1);
        res[e] = g.residual ? g.residual[row * g.ldc + col] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int trow = wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
        const int64_t row = m0 + trow;
        if (row >= g.M) continue;
        float v = acc[i][j][e];
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
#if defined(APS_DEBUG_DISTURBANCE) && defined(APS_DEBUG_FORCE_128_VGPRS)
  asm volatile("v_mov_b32 v127, 0" ::: "v127");  // experiment: the allocation of the 128-VGPR build
#endif
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

// W [N, K] (row pitch ldw) -> planes[ksteps][3][Np][32] bf16, zero padded
__global__ __launch_bounds__(256) void split_weight_kernel(const float* __restrict__ W,
                                                          u32x4* __restrict__ planes, int64_t N,
                                                          int64_t K, int64_t ldw, int64_t np,
                                                          int64_t ksteps) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (row, step, chunk)
  if (idx >= np * ksteps * 4) return;
  const int64_t row = idx / (ksteps * 4), rem = idx % (ksteps * 4);
  const int64_t step = rem >> 2, c = rem & 3;
  uint32_t x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int64_t k = step * 32 + c * 8 + e;
    x[e] = (row < N && k < K) ? __float_as_uint(W[row * ldw + k]) : 0u;
  }
  const Planes8 p = split8(u32x4{x[0], x[1], x[2], x[3]}, u32x4{x[4], x[5], x[6], x[7]});
  u32x4* dst = planes + (step * 3 * np + row) * 4 + c;  // 64 B = 4 x 16 B per (step, plane, row)
  dst[0] = p.h;
  dst[np * 4] = p.m;
  dst[np * 8] = p.l;
}

template <int TN, bool LN, bool RAGGED>
static int launch_split_swp(SplitGemmArgs g, hipStream_t st) {
  const int64_t tiles_m = (g.M + 127) / 128, tiles_n = (g.N + TN - 1) / TN;
  const int64_t total = tiles_m * tiles_n;
  if (total > 0x7fffffff) return APS_ERR_INVALID;
  g.tiles_n = (int32_t)tiles_n;
  g.remap = (total % 8 == 0) ? 1 : 0;
  constexpr size_t lds = 2 * 3 * (size_t)(128 + TN) * 64;
  static ApsPerDevice attr_set;  // > 64 KB of dynamic LDS needs the opt-in once per device
  if (lds > 64 * 1024 &&
      !aps_lds_opt_in(attr_set, reinterpret_cast<const void*>(&gemm_split_swp_kernel<TN, LN, RAGGED>),
                      (int)lds))
    return APS_ERR_LAUNCH;
  hipLaunchKernelGGL((gemm_split_swp_kernel<TN, LN, RAGGED>), dim3((unsigned)total), dim3(256), lds, st,
                     g);
  return aps_launch_status();
}

template <bool LN, bool RAGGED>
static int launch_split_pc(SplitGemmArgs g, hipStream_t st) {
  const int64_t tiles_m = (g.M + 127) / 128, tiles_n = (g.N + 127) / 128;
  const int64_t total = tiles_m * tiles_n;
  if (total > 0x7fffffff) return APS_ERR_INVALID;
  g.tiles_n = (int32_t)tiles_n;
  g.remap = (total % 8 == 0) ? 1 : 0;
  constexpr size_t lds = 3 * 3 * (size_t)(128 + 128) * 64;  // ring of three 48 KB buffers
  static ApsPerDevice attr_set;
  if (!aps_lds_opt_in(attr_set, reinterpret_cast<const void*>(&gemm_split_pc_kernel<LN, RAGGED>), (int)lds))
    return APS_ERR_LAUNCH;
  hipLaunchKernelGGL((gemm_split_pc_kernel<LN, RAGGED>), dim3((unsigned)total), dim3(512), lds, st, g);
  return aps_launch_status();
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

template <int TN, bool LN>
static int launch_split(SplitGemmArgs g, hipStream_t st) {
  const int64_t tiles_m = (g.M + 127) / 128, tiles_n = (g.N + TN - 1) / TN;
  const int64_t total = tiles_m * tiles_n;
  if (total > 0x7fffffff) return APS_ERR_INVALID;
  g.tiles_n = (int32_t)tiles_n;
  g.remap = (total % 8 == 0) ? 1 : 0;
  constexpr size_t lds = 3 * (size_t)(128 + TN) * 64;
  hipLaunchKernelGGL((gemm_split_kernel<TN, LN>), dim3((unsigned)total), dim3(256), lds, st, g);
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
  APS_CHECK_ARG(((uintptr_t)planes & 15) == 0 && (layout == 0 || layout == 1));
  const int64_t np = ((N + 127) / 128) * 128, ksteps = (K + 31) / 32;
  if (np * ksteps * 192 >= ((int64_t)1 << 31)) return APS_ERR_UNSUPPORTED;
  if (layout == 1) {
    const int64_t groups = np / 32, threads = ksteps * groups * 128;
    hipLaunchKernelGGL(split_weight_frag_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), W, reinterpret_cast<u32x4*>(planes), N, K, ldw,
                       groups, ksteps);
    return aps_launch_status();
  }
  const int64_t threads = np * ksteps * 4;
  hipLaunchKernelGGL(split_weight_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), W, reinterpret_cast<u32x4*>(planes), N, K, ldw,
                     np, ksteps);
  return aps_launch_status();
}

extern "C" int aps_linear_split(const float* A, const void* planes, const float* bias,
                                const float* colsum, const float* residual, float* C, int64_t M,
                                int64_t N, int64_t K, int64_t lda, int64_t ldc, int32_t act,
                                float alpha, float eps, int32_t layout, void* stream) {
  APS_CHECK_ARG(A && planes && C && M > 0 && N > 0 && K > 0 && (layout == 0 || layout == 1));
  APS_CHECK_ARG(lda >= K && ldc >= N && lda % 4 == 0 && ((uintptr_t)A & 15) == 0 &&
                ((uintptr_t)planes & 15) == 0);
  APS_CHECK_ARG(act >= 0 && act <= 5);
  const int64_t ksteps = (K + 31) / 32;
  if (M * lda * 4 >= ((int64_t)1 << 31) || aps_linear_split_size(N, K) >= ((int64_t)1 << 31))
    return APS_ERR_UNSUPPORTED;
  SplitGemmArgs g{A, planes, bias, residual, C, M, N, K, lda, ldc, act, alpha, 0, 0, (int32_t)ksteps,
                  colsum, eps};
  hipStream_t st = static_cast<hipStream_t>(stream);
  // (a ring of THREE A buffers -- the tile of step s + 2 written before the MFMAs of step s, so that
  // the barrier's lgkmcnt wait finds the writes long done -- measured slower too: 36.8 against 34-36 us
  // at N = 512, 115 against 109 at N = 2048, joint step 12 860 against 13 250 utt/s)
  // (a 128-row form of the same kernel -- half the weight re-reads and barriers per MFMA, but 190-210
  // VGPRs = two workgroups per CU -- measured slower at every shape, M = 8064 and 31872: 117 against
  // 111 us at N = 2048, 46 against 36 us at N = 512; occupancy buys more here than reuse)
#ifdef APS_DEBUG_DISTURBANCE
  // experiments only: the 32-row form that triggers the cross-stream disturbance
  if (layout == 1 && getenv("APS_SPLIT_TM") && atoi(getenv("APS_SPLIT_TM")) == 32)
    return colsum ? launch_split_bd<32, true>(g, st) : launch_split_bd<32, false>(g, st);
#endif
  if (layout == 1) return colsum ? launch_split_bd<64, true>(g, st) : launch_split_bd<64, false>(g, st);
  // Kernel choice (APS_SPLIT_KERNEL = v1 | swp | pc forces one, APS_SPLIT_TN the tile width of the
  // first two).  In isolation (M = 8064, scripts/split_gemm_bench.py) the producer / consumer
  // kernel wins the long K loops (K = 2048: 96 us against 119 / 110 us) and the occupancy-driven
  // 128 x 64 kernel the short ones (K = 512, N >= 1024).  Inside the joint step, where two
  // batches share the chip, the 128 x 64 kernel (36 KB of LDS, <= 112 VGPRs: four workgroups per CU
  // and room beside the other stream's kernels) wins everywhere: 12 500 utt/s against 10 600 with
  // the producer / consumer kernel on the K >= 1024 launches and 10 300 with it on all of them
  // (its 144 KB of LDS take a CU for themselves); fp32 MFMA: 10 370 (scripts/gpu_split_ab.sh).
  const char* kenv = getenv("APS_SPLIT_KERNEL");
  const char* tenv = getenv("APS_SPLIT_TN");
  int kernel = 0;  // 0: v1, 1: swp, 2: pc
  if (kenv) kernel = (kenv[0] == 'p') ? 2 : (kenv[0] == 's') ? 1 : 0;
  const int tn = tenv ? atoi(tenv) : 64;
  const bool ragged = (K & 31) != 0;
  if (kernel == 2) {
    if (colsum) return ragged ? launch_split_pc<true, true>(g, st) : launch_split_pc<true, false>(g, st);
    return ragged ? launch_split_pc<false, true>(g, st) : launch_split_pc<false, false>(g, st);
  }
  if (kernel == 1) {
#define APS_SPLIT_CASE(TN_, LN_, RG_) \
  if (tn == TN_ && (colsum != nullptr) == LN_ && ragged == RG_) return launch_split_swp<TN_, LN_, RG_>(g, st);
    APS_SPLIT_CASE(128, false, false)
    APS_SPLIT_CASE(128, true, false)
    APS_SPLIT_CASE(128, false, true)
    APS_SPLIT_CASE(128, true, true)
    APS_SPLIT_CASE(64, false, false)
    APS_SPLIT_CASE(64, true, false)
    APS_SPLIT_CASE(64, false, true)
    APS_SPLIT_CASE(64, true, true)
#undef APS_SPLIT_CASE
    return APS_ERR_INVALID;
  }
  if (tn == 128) return colsum ? launch_split<128, true>(g, st) : launch_split<128, false>(g, st);
  return colsum ? launch_split<64, true>(g, st) : launch_split<64, false>(g, st);
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
