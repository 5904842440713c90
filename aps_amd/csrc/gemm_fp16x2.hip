// fp32 GEMMs on the fp16 matrix pipe with HALF the matrix work of gemm_split.hip (aps_linear_fp16x2*):
// two fp16 planes and three products instead of three bf16 planes and six.
//
// Arithmetic (round 3: the low plane carries its own power of two, the cross terms their own
// accumulator, and a tile whose operands do not fit is recomputed in fp32 -- the round-2 form lost the
// low plane of every element more than 2^17 below its row maximum, which broke the bound whenever the
// row maximum met a zero weight).  Both operands carry a power-of-two scale PER ROW (the row of A, the
// output column of W) that puts the row's largest magnitude into [2^14, 2^15):
//     a' = a 2^ea[m]    w' = w 2^ew[n]
//     h = rn_f16(x')    l = rn_f16((x' - h) 2^11)          x' = h + l 2^-11 + e,  |e| <= 2^-22 |x'|
//     C[m, n] = 2^-(ea[m] + ew[n]) ( sum_k ha hw  +  2^-11 sum_k (ha lw + la hw) )
// Every plane product is exact in fp32 (11 x 11 bits); the two sums live in two fp32 accumulators
// (main, cross) and meet once in the epilogue; dropped: la lw <= 2^-22 |a w| and the two e terms.
// h is a normal fp16 down to 2^-14 and the scaled residue (x' - h) 2^11 down to the same, so an element
// keeps 22 bits as long as |x'| >= 2^-14, i.e. within 2^-28 of its row maximum; below that the pair
// rounds at 2^-36 absolute.  Elements with 0 < |x'| < 2^-16 (more than 2^30 below the row maximum:
// relative error above 2^-20) -- and elements at or above 2^15, which only a stale row-maximum hint
// can produce -- are DETECTED where the planes are formed (the staging lanes of A, the image builder
// of W) and the 64 x 128 tile they touch is recomputed on the fp32 MFMA from the fp32 operands (same
// launch, same epilogue; `wide_count` counts such tiles).  Hence, for ANY finite input, what the
// REPRESENTATION loses (element error 2^-20, the dropped l l term, the planes' own 2 x 2^-22) is
//     <= 2^-19 sum_k |a_k| |w_k|                    (2^-20.5 measured when no element lies more than
//                                                    2^28 below its row maximum)
// next to the rounding of the fp32 ACCUMULATION that every fp32 evaluation carries: K / 16 sequential
// additions per output on the planes (the 16 products of an MFMA are summed exactly), K / 2 on the
// fp32 path -- the fp32 MFMA kernel's own figure (2^-21 typical, 2^-18.7 on heavy-tailed rows of K = 768).
// scripts/split_fp16_emulation.py emulates the arithmetic exactly (nine operand distributions and
// the outlier-column x zero-weight case at in-row ranges 1e5 .. 1e10); tests/test_fp16x2_arithmetic.py
// holds the bound on the CPU, tests/test_gpu_encoder.py on the kernel.
//
// Data.  The row exponents of A are one pass over A (row_exp_kernel, 16 lanes per row) -- or no pass
// at all when A was written by this kernel: on request its epilogue leaves one maximum of |C| per
// row and wave (32 columns; five DPP steps per value), [N / 32][M] floats that the consumer's
// staging lanes fold into their row's exponent while the first tiles are in flight (a wrong hint
// cannot corrupt a result: an element that overflows its scale sends the tile to the fp32 path); the
// weight image (aps_linear_fp16x2_weight) is the fragment-ordered image of gemm_split.hip with two
// planes -- [K step][32-column group][MFMA K step 2][plane 2][lane 64][8 f16], 4 KB per group-step --
// followed by the int32 exponents of the N weight rows and their int32 "wide" flags.  The kernel is
// gemm_split_bd_kernel's structure: 64 x 128 tile, four waves side by side along N, weight operands
// straight from the image into registers, the two planes of the 64 A rows through a double-buffered
// LDS (8 KB per buffer, swizzled 64-byte rows), one barrier per K step; the A rows of K step s + 2
// are requested while step s computes.
#include <stdint.h>

#include <type_traits>

#include "common.h"
#include "conv_core.h"

namespace aps {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// the low plane holds (x' - h) 2^kLowShift; the cross accumulator is folded in with 2^-kLowShift
constexpr int kLowShift = 11;
constexpr float kLowUp = 2048.f, kLowDown = 1.0f / 2048.f;
// "fits" = 0, or 2^-16 <= |x'| < 2^15: frexp exponent e in [-15, 15] (|x'| in [2^(e-1), 2^e)); the
// staging code tracks the unsigned maximum of e + 15, which exceeds 30 exactly when some element does
// not fit (e < -15 wraps around); frexp gives 0 for zero, inf and NaN (those propagate as in fp32)
constexpr int kFitBias = 15;
constexpr uint32_t kFitMax = 30u;

// the exponent that brings a row maximum `mx` into [2^14, 2^15) (zero / subnormal rows are treated
// as the smallest normal, inf as the largest finite: their products are inf / nan either way)
__device__ __forceinline__ int32_t scale_exponent(float mx) {
  int be = (int)((__float_as_uint(mx) >> 23) & 0xffu);
  be = be < 1 ? 1 : (be > 254 ? 254 : be);
  return 141 - be;
}

// e + 16 of a scaled element (see kFitBias)
__device__ __forceinline__ uint32_t fit_key(float scaled) {
  return (uint32_t)(__builtin_amdgcn_frexp_expf(scaled) + kFitBias);
}

// ea[row] for rows of X [rows, K] (row pitch ldx floats, 16-byte aligned rows); 16 lanes per row
__global__ __launch_bounds__(256) void row_exp_kernel(const float* __restrict__ X,
                                                     int32_t* __restrict__ e, int64_t rows,
                                                     int64_t K, int64_t ldx) {
  const int q = threadIdx.x & 15;
  const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const float* x = X + (row < rows ? row : rows - 1) * ldx;
  float mx = 0.f;
  int64_t k = q * 4;
  for (; k + 3 < K; k += 64) {
    const float4 v = *reinterpret_cast<const float4*>(x + k);
    mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
  }
  if (q == 0)  // the K % 4 tail
    for (int64_t t = K & ~(int64_t)3; t < K; ++t) mx = fmaxf(mx, fabsf(x[t]));
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if (q == 0 && row < rows) e[row] = scale_exponent(mx);
}

// 4 scaled fp32 -> 4 h and 4 l halves (l = the residue times 2^kLowShift)
__device__ __forceinline__ void split4(const float s[4], u32x2& h, u32x2& l) {
  _Float16 hh[4], ll[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    hh[e] = (_Float16)s[e];
    ll[e] = (_Float16)((s[e] - (float)hh[e]) * kLowUp);
  }
  h = u32x2{__builtin_bit_cast(uint32_t, f16x2{hh[0], hh[1]}), __builtin_bit_cast(uint32_t, f16x2{hh[2], hh[3]})};
  l = u32x2{__builtin_bit_cast(uint32_t, f16x2{ll[0], ll[1]}), __builtin_bit_cast(uint32_t, f16x2{ll[2], ll[3]})};
}

// W [N, K] -> fragment image: lane l of (step, group, kk) holds column 32 g + (l & 31),
// k = 32 step + 16 kk + 8 (l >> 5) .. + 7, scaled by 2^ew[column]; wide[column] is raised when an
// element of the column does not fit its scale (the tiles of that column then take the fp32 path)
__global__ __launch_bounds__(256) void fp16x2_weight_kernel(const float* __restrict__ W,
                                                           u32x4* __restrict__ image,
                                                           const int32_t* __restrict__ ew,
                                                           int32_t* __restrict__ wide, int64_t N,
                                                           int64_t K, int64_t ldw, int64_t groups,
                                                           int64_t ksteps) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (step, group, kk, lane)
  if (idx >= ksteps * groups * 128) return;
  const int lane = (int)(idx & 63), kk = (int)((idx >> 6) & 1);
  const int64_t grp = (idx >> 7) % groups, step = (idx >> 7) / groups;
  const int64_t row = grp * 32 + (lane & 31);
  const int sc = row < N ? ew[row] : 0;
  float s[8];
  uint32_t key = 0;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int64_t k = step * 32 + kk * 16 + (lane >> 5) * 8 + e;
    s[e] = (row < N && k < K) ? ldexpf(W[row * ldw + k], sc) : 0.f;
    key = max(key, fit_key(s[e]));
  }
  if (key > kFitMax) atomicOr(&wide[row], 1);
  u32x2 h0, l0, h1, l1;
  split4(s, h0, l0);
  split4(s + 4, h1, l1);
  u32x4* dst = image + ((step * groups + grp) * 4 + kk * 2) * 64 + lane;
  dst[0] = u32x4{h0.x, h0.y, h1.x, h1.y};
  dst[64] = u32x4{l0.x, l0.y, l1.x, l1.y};
}

__device__ __forceinline__ f32x16 mfma_f16(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b),
                                                c, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------
// The planes of A are formed ONCE per GEMM input, by a pass of their own (round 3): the round-2 kernel
// split every A element again in each of the N / 128 column tiles that read it (4 to 16 times), with
// ~12 VALU instructions per element between the request of an A tile and its LDS image -- a third of
// the K step's dependent chain (scripts/gemm_trace.py: issue loads 290, first-half MFMAs + waits
// 410, second half 290, split + LDS write 440, barrier 170 of ~1 900 cycles per step).
//
// Image of A (fp16x2_split_kernel): [K step][plane h | l][row, padded to 64][32 k] f16 -- the 64 rows
// x 32 k of a row panel's K step are 4 KB contiguous per plane, so the consumer's staging lanes read
// whole 1 KB runs (one 16-byte request per lane) and hand them to LDS with ds_write_b128, nothing in
// between.  Next to it: rowinfo[row] = (exponent + 512) | wide << 16, and (mean, variance) of the RAW
// row for the LayerNorm fold.
// ------------------------------------------------------------------------------------------
constexpr int kInfoBias = 512;

// 8 scaled fp32 -> the 8 h and the 8 l halves of a lane's 16-byte pieces
__device__ __forceinline__ void split8(const float s[8], u32x4& h, u32x4& l) {
  u32x2 h0, l0, h1, l1;
  split4(s, h0, l0);
  split4(s + 4, h1, l1);
  h = u32x4{h0.x, h0.y, h1.x, h1.y};
  l = u32x4{l0.x, l0.y, l1.x, l1.y};
}

// 16 lanes per row, a lane owns the 8 consecutive k at 8 q + 128 j.  CHUNKS > 0: the row's
// 128 CHUNKS elements stay in registers between the maximum and the split (one pass over A: rows up to
// K = 1024); CHUNKS = 0: any K, the row is read a second time (from L1 / L2).
template <int CHUNKS>
__global__ __launch_bounds__(256) void fp16x2_split_kernel(const float* __restrict__ A,
                                                          u32x4* __restrict__ planes,
                                                          int32_t* __restrict__ rowinfo,
                                                          float2* __restrict__ stat, int64_t M,
                                                          int64_t Mp, int64_t K, int64_t ksteps,
                                                          int64_t lda) {
  const int q = threadIdx.x & 15;
  const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);  // < Mp (Mp is a multiple of 64)
  const bool live = row < M;
  const float* x = A + (live ? row : 0) * lda;
  const int64_t kend = ksteps * 32;
  auto fetch8 = [&](int64_t k, float v[8]) {  // elements k .. k + 7 of the row, zeros past K / M
    if (live && k + 7 < K) {
      const float4 t0 = *reinterpret_cast<const float4*>(x + k);
      const float4 t1 = *reinterpret_cast<const float4*>(x + k + 4);
      v[0] = t0.x, v[1] = t0.y, v[2] = t0.z, v[3] = t0.w;
      v[4] = t1.x, v[5] = t1.y, v[6] = t1.z, v[7] = t1.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (live && k + e < K) ? x[k + e] : 0.f;
    }
  };
  float mx = 0.f, s1 = 0.f, s2 = 0.f;
  auto scan8 = [&](const float v[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      mx = fmaxf(mx, fabsf(v[e]));
      s1 += v[e];
      s2 = fmaf(v[e], v[e], s2);
    }
  };
  constexpr int HELD = CHUNKS > 0 ? CHUNKS : 1;
  float held[HELD][8];
  if (CHUNKS > 0) {
#pragma unroll
    for (int j = 0; j < HELD; ++j) {
      fetch8(8 * q + 128 * j, held[j]);  // (chunks past the row read nothing and hold zeros)
      scan8(held[j]);
    }
  } else {
    for (int64_t k = 8 * q; k < K; k += 128) {
      float v[8];
      fetch8(k, v);
      scan8(v);
    }
  }
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) {
    mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    s1 += __shfl_xor(s1, o, 64);
    s2 += __shfl_xor(s2, o, 64);
  }
  const int32_t ex = scale_exponent(mx);
  uint32_t fit = 0;
  // a plane row of a K step = 32 f16 = 4 x 16 bytes; lane q writes piece q % 4 of K step 4 j + q / 4
  u32x4* dst = planes + row * 4 + (q & 3);
  auto emit8 = [&](int64_t k, float v[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v[e] = ldexpf(v[e], ex);
      fit = max(fit, fit_key(v[e]));
    }
    u32x4 h, l;
    split8(v, h, l);
    const int64_t ks = k >> 5;
    dst[(ks * 2 + 0) * Mp * 4] = h;
    dst[(ks * 2 + 1) * Mp * 4] = l;
  };
  if (CHUNKS > 0) {
#pragma unroll
    for (int j = 0; j < HELD; ++j)
      if (8 * q + 128 * j < kend) emit8(8 * q + 128 * j, held[j]);
  } else {
    for (int64_t k = 8 * q; k < kend; k += 128) {
      float v[8];
      fetch8(k, v);
      emit8(k, v);
    }
  }
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) fit = max(fit, (uint32_t)__shfl_xor((int)fit, o, 64));
  if (q == 0) {
    rowinfo[row] = live ? ((ex + kInfoBias) | (fit > kFitMax ? 0x10000 : 0)) : kInfoBias;
    const float mean = s1 / (float)K;
    stat[row] = make_float2(mean, fmaxf(s2 / (float)K - mean * mean, 0.f));
  }
}

struct Fp16GemmArgs {
  const void* Ap;          // planes image of A (fp16x2_split_kernel)
  const int32_t* rowinfo;  // [Mp]
  const float2* stat;      // [Mp] (mean, variance) of the raw rows
  const float* A32;        // the fp32 rows (row pitch lda): read only on the fp32 path
  const void* Wp;          // image of W (aps_linear_fp16x2_weight): fragments, int32 ew[Np], int32 wide[Np]
  const float* W32;        // the fp32 weight the image was made from [N, K] (row pitch ldw): the fp32 path
  const float* bias;       // [N] or null
  const float* residual;   // [M, N] (ldc) or null
  float* C;
  int32_t* wide_count;     // device counter of tiles recomputed in fp32 (or null)
  int64_t M, Mp, N, K;
  int64_t lda, ldw, ldc;
  int32_t act;
  float alpha;
  int32_t tiles_n, remap, ksteps;
  int32_t total;       // output tiles
  const float* ln_cs;  // LayerNorm fold: column sums of W diag(gamma) (null: plain)
  float ln_eps;
};

#ifdef APS_FP16X2_TRACE
// experiments only (scripts/gemm_trace.py with a library built with -DAPS_FP16X2_TRACE): s_memtime stamps
// of one lane per wave of eight workgroups, [workgroup slot 8][wave 4][K step 64][stamp 8]
__device__ unsigned long long g_fp16x2_trace[8 * 4 * 64 * 8];
// per workgroup (first 4096 of a launch): block id, tile, HW_ID, XCC_ID, s_memtime at kernel entry /
// first K step / after the last K step / after the epilogue's last store has left the wave
__device__ unsigned long long g_fp16x2_wgtrace[4096 * 8];
#define APS_TRACE_STAMP(k) \
  if (tracing) stamp[k] = __builtin_amdgcn_s_memtime();
#define APS_WG_STAMP(k) wgstamp[k] = __builtin_amdgcn_s_memtime();
#else
#define APS_TRACE_STAMP(k)
#define APS_WG_STAMP(k)
#endif

// A-prefetch depth (K steps between the request of an A tile and its hand-over to LDS: 1 = requested
// at the top of the step that stages it, 2 = a step earlier, 8 more VGPRs) and the weight-operand
// buffering (0 = two register stages, the next step's fragments requested at the top of a step;
// 1 = one stage, a fragment pair re-requested as soon as its last MFMA has issued: 16 VGPRs fewer).
// scripts/build_variant_lib.sh builds the other combinations for A/B runs.
#ifndef APS_FP16X2_APREF
#define APS_FP16X2_APREF 2
#endif
#ifndef APS_FP16X2_WJIT
#define APS_FP16X2_WJIT 1
#endif
#ifndef APS_FP16X2_FRAG_BY_BLOCK
#define APS_FP16X2_FRAG_BY_BLOCK 1
#endif
// workgroups per CU the kernel is compiled for (a register bound, not a promise)
#ifndef APS_FP16X2_MIN_WG
#define APS_FP16X2_MIN_WG 4
#endif
// 1: the grid is capped at what the chip holds at once and a workgroup walks several tiles (needs
// 149-153 VGPRs: three workgroups per CU; measured against the one-tile form in profiles/r03_gemm_ab.txt)
#ifndef APS_FP16X2_PERSISTENT
#define APS_FP16X2_PERSISTENT 0
#endif

// NARROW: a 64 x 64 tile -- waves 0, 1 own the upper 32 rows of column groups 0, 1, waves 2, 3 the lower
// 32 (one row block per wave: half the accumulators).  For launches whose 64 x 128 tiles would leave the
// chip with about one workgroup per CU (BASELINE's 32 utterances per GPU, M = 2016: 128 ... 384 tiles): twice
// the workgroups, each with half the MFMAs per K step, keep the matrix pipe busier while the requests
// of a K step are outstanding.  The weight fragments of a column group are requested by two waves
// then.  Measured (profiles/r03_gemm_ab.txt): the 32-utterance step 10 600 -> 11 160 utt/s, with the N = 512
// projections moved over from the fp32 kernel 11 650; at M = 8064, N = 512 (504 tiles) no gain (34.8 / 57.5 us
// against 34.6 / 54.9 per call), so the form stops at 400 tiles.
template <bool LN, bool NARROW = false>
__global__ __launch_bounds__(256, APS_FP16X2_MIN_WG) void gemm_fp16x2_kernel(Fp16GemmArgs g) {
  constexpr int TM = 64, TN = NARROW ? 64 : 128, SM = NARROW ? 1 : TM / 32;
  constexpr int APREF = APS_FP16X2_APREF, WJIT = APS_FP16X2_WJIT, WST = WJIT ? 1 : 2;
  constexpr int kRowB = 64;
  constexpr int kBuf = 2 * TM * kRowB;  // 8 KB: the two A planes of one K step
  // (2 x 8 KB of A planes in the K loop; 4 x 4.5 KB of C blocks in the epilogue)
  __shared__ __attribute__((aligned(16))) unsigned char s_a[(2 * kBuf > 4 * 32 * 36 * 4) ? 2 * kBuf : 4 * 32 * 36 * 4];
  __shared__ int32_t s_exp[TM + 4];  // row exponents; [TM] = "this tile takes the fp32 path"
  __shared__ float2 s_stat[TM];      // LayerNorm fold: (mean, 1 / sqrt(var + eps)) of the rows
  const int tid = threadIdx.x, ln = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // (wave-uniform: a scalar register)
  const int wn = NARROW ? (wv & 1) : wv;   // 32-column group of the tile this wave owns
  const int wr = NARROW ? (wv >> 1) : 0;   // its first 32-row block
#ifdef APS_FP16X2_TRACE
  unsigned long long wgstamp[4];
#endif
  // Persistent workgroups: the grid is at most what the chip holds at once (launch_fp16x2) and a
  // workgroup walks tiles blockIdx.x, + gridDim.x, ...: the stores of a tile drain while the next one's
  // first requests are in flight, nothing waits for a launch or a dispatch slot, and the workgroups
  // of a CU drift out of step -- one's epilogue runs beside the others' MFMAs (launched as one wave of
  // lock-step workgroups the chip alternated between a load burst, the MFMA loops and a store burst).
  // With a grid that is a multiple of 8 a workgroup stays on its XCD, and `remap` keeps giving every
  // XCD a contiguous range of row panels.
#if APS_FP16X2_PERSISTENT
  for (int32_t tile = (int32_t)blockIdx.x; tile < g.total; tile += (int32_t)gridDim.x) {
#else
  {  // (one tile per workgroup: the tile loop costs the K loop 26 VGPRs, i.e. the fourth workgroup per CU)
  const int32_t tile = (int32_t)blockIdx.x;
#endif
  APS_WG_STAMP(0)
  int32_t lin = tile;
  if (g.remap) lin = (tile & 7) * (g.total >> 3) + (tile >> 3);
  const int32_t panel = lin / g.tiles_n;
  const int32_t m0 = panel * TM, n0 = (lin - panel * g.tiles_n) * TN;  // (Mp, Np < 2^31 / 64: checked)

  f32x16 acc[SM], accx[SM];  // main (h h) and cross (h l + l h, at 2^kLowShift) sums
#pragma unroll
  for (int i = 0; i < SM; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = accx[i][e] = 0.f;

  const int64_t groups = ((g.N + 127) / 128) * 4;  // 32-column groups of the image
  const int32_t wstep_bytes = (int32_t)(groups * 4096);
  auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.Wp), 0,
                                                  (uint32_t)(wstep_bytes * g.ksteps), 0x00020000);
  const int32_t* ew_tab = reinterpret_cast<const int32_t*>(static_cast<const unsigned char*>(g.Wp) +
                                                           (int64_t)wstep_bytes * g.ksteps);
  // a K step of the A image: two planes of Mp rows x 64 bytes
  const int32_t astep_bytes = (int32_t)(g.Mp * 128), aplane_bytes = (int32_t)(g.Mp * 64);
  auto rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.Ap), 0,
                                                  (uint32_t)((int64_t)astep_bytes * g.ksteps), 0x00020000);
  // (the row information of the panel is requested BEHIND the first operand tiles, below: ahead of
  // them its two dependent round trips -- the weight columns' flags, then the rows' -- stood in front
  // of the first A / W request of every tile)
  // staging lane t hands over 16 bytes of each plane: row t / 4 of the panel, 16-byte chunk t % 4
  const int32_t va = m0 * 64 + tid * 16;
  const int srow = tid >> 2;
  unsigned char* const sdst = s_a + srow * kRowB + (((tid & 3) ^ ((srow >> 2) & 3)) << 4);
  const int32_t vw = (n0 / 32 + wn) * 4096 + ln * 16;

  const int nsteps = g.ksteps;
  const int rot = panel % nsteps;
  u32x4 ra[APREF][2];   // [slot][plane]
  u32x4 wb[WST][2][2];  // [register stage][MFMA K step][plane]
  auto tile_at = [&](int s) { return (s + rot >= nsteps) ? s + rot - nsteps : s + rot; };
  auto gload_a = [&](auto slot, int s) {
    constexpr int R = decltype(slot)::value;
    const int32_t soff = tile_at(s) * astep_bytes;
    ra[R][0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, va, soff, 0);
    ra[R][1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, va, soff + aplane_bytes, 0);
  };
  auto gload_w = [&](auto stage, auto kkc, int s) {  // the two planes of MFMA K step kk of K step s
    constexpr int P = decltype(stage)::value, kk = decltype(kkc)::value;
    const int32_t soff = tile_at(s) * wstep_bytes;
#pragma unroll
    for (int p = 0; p < 2; ++p)
      wb[P][kk][p] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, vw, soff + (kk * 2 + p) * 1024, 0);
  };
  auto sstore = [&](int buf, auto slot) {
    constexpr int R = decltype(slot)::value;
    *reinterpret_cast<u32x4*>(sdst + buf * kBuf) = ra[R][0];
    *reinterpret_cast<u32x4*>(sdst + buf * kBuf + TM * kRowB) = ra[R][1];
  };
  const int frow = ln & 31, fsw = (frow >> 2) & 3;
  auto compute = [&](auto stage, auto kkc, int buf) {
    constexpr int P = decltype(stage)::value, kk = decltype(kkc)::value;
    const unsigned char* fa = s_a + buf * kBuf + (wr * 32 + frow) * kRowB;  // (wr * 8 leaves fsw as it is)
    const int off = ((kk * 2 + (ln >> 5)) ^ fsw) << 4;
#if APS_FP16X2_FRAG_BY_BLOCK
    // one row block at a time: 8 fragment registers live instead of 16
#pragma unroll
    for (int i = 0; i < SM; ++i) {
      const u32x4 ah = *reinterpret_cast<const u32x4*>(fa + i * 32 * kRowB + off);
      const u32x4 al = *reinterpret_cast<const u32x4*>(fa + TM * kRowB + i * 32 * kRowB + off);
      accx[i] = mfma_f16(ah, wb[P][kk][1], accx[i]);  // h l
      acc[i] = mfma_f16(ah, wb[P][kk][0], acc[i]);    // h h
      accx[i] = mfma_f16(al, wb[P][kk][0], accx[i]);  // l h
    }
#else
    u32x4 a[SM][2];
#pragma unroll
    for (int i = 0; i < SM; ++i)
#pragma unroll
      for (int p = 0; p < 2; ++p)
        a[i][p] = *reinterpret_cast<const u32x4*>(fa + p * TM * kRowB + i * 32 * kRowB + off);
#pragma unroll
    for (int i = 0; i < SM; ++i) accx[i] = mfma_f16(a[i][0], wb[P][kk][1], accx[i]);  // h l
#pragma unroll
    for (int i = 0; i < SM; ++i) accx[i] = mfma_f16(a[i][1], wb[P][kk][0], accx[i]);  // l h
#pragma unroll
    for (int i = 0; i < SM; ++i) acc[i] = mfma_f16(a[i][0], wb[P][kk][0], acc[i]);    // h h
#endif
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  auto step_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
#ifdef APS_FP16X2_TRACE
  const int tslot = (lin >= 100 && lin < 104) ? (int)lin - 100 : (lin >= 500 && lin < 502) ? (int)lin - 496
                  : (lin >= 900 && lin < 902) ? (int)lin - 894 : -1;
  const bool tracing = tslot >= 0;
  unsigned long long stamp[8];
#endif
  // K step s (parity PAR): the A tile of step s + 1 goes to LDS while it computes
  auto kstep = [&](auto par, int s) {
    constexpr int PAR = decltype(par)::value;
    using RaNext = std::integral_constant<int, (APREF == 2) ? PAR : 0>;      // receives A(s + APREF)
    using RaUse = std::integral_constant<int, (APREF == 2) ? (PAR ^ 1) : 0>;  // holds A(s + 1)
    using WbUse = std::integral_constant<int, WJIT ? 0 : PAR>;
    using WbNext = std::integral_constant<int, WJIT ? 0 : (PAR ^ 1)>;
    const bool next = s + 1 < nsteps;
    APS_TRACE_STAMP(0)
    if (s + APREF < nsteps) gload_a(RaNext{}, s + APREF);
    if (!WJIT && next) {
      gload_w(WbNext{}, I0{}, s + 1);
      gload_w(WbNext{}, I1{}, s + 1);
    }
    APS_TRACE_STAMP(1)
    compute(WbUse{}, I0{}, PAR);
    APS_TRACE_STAMP(2)
    if (WJIT && next) gload_w(WbNext{}, I0{}, s + 1);
    compute(WbUse{}, I1{}, PAR);
    APS_TRACE_STAMP(3)
    if (WJIT && next) gload_w(WbNext{}, I1{}, s + 1);
    if (next) sstore(PAR ^ 1, RaUse{});
    APS_TRACE_STAMP(4)
#ifdef APS_FP16X2_TRACE
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    APS_TRACE_STAMP(5)
    __builtin_amdgcn_s_barrier();
    APS_TRACE_STAMP(6)
    if (tracing && ln == 0 && s < 64) {
      unsigned long long* t = g_fp16x2_trace + ((tslot * 4 + wv) * 64 + s) * 8;
#pragma unroll
      for (int k = 0; k < 7; ++k) t[k] = stamp[k];
    }
#else
    step_barrier();
#endif
  };

  gload_a(I0{}, 0);
  if constexpr (APREF == 2) {
    if (nsteps > 1) gload_a(I1{}, 1);
  }
  gload_w(I0{}, I0{}, 0);
  gload_w(I0{}, I1{}, 0);
  // the row information of the panel: exponents for the epilogue, the wide flags, the fold's
  // statistics -- in flight together with the tiles above
  const int32_t ew_flag = ew_tab[groups * 32 + n0 + wn * 32 + (ln & 31)];  // a wide weight column
  int32_t info = 0;
  float2 st = make_float2(0.f, 1.f);
  if (tid < TM) {
    info = g.rowinfo[m0 + tid];
    if (LN) st = g.stat[m0 + tid];
  }
  sstore(0, I0{});
  bool wide = __any(ew_flag != 0);
  if (tid < TM) {
    s_exp[tid] = (info & 0xffff) - kInfoBias;
    wide |= (info >> 16) != 0;
    if (LN) s_stat[tid] = make_float2(st.x, 1.0f / sqrtf(st.y + g.ln_eps));
  }
  if (tid == 0) s_exp[TM] = 0;
  step_barrier();
  // does any operand of this tile fail to fit its scale?  (cleared before the barrier above; every
  // K step ends with one more before the flag is read)
  if (wide) s_exp[TM] = 1;
  APS_WG_STAMP(1)
  int s = 0;
  for (; s + 1 < nsteps; s += 2) {
    kstep(I0{}, s);
    kstep(I1{}, s + 1);
  }
  if (s < nsteps) kstep(I0{}, s);
  APS_WG_STAMP(2)
  const bool tile_wide = s_exp[TM] != 0;

  // (lane-derived values of the epilogue are re-derived from an opaque copy of the lane id: hoisted out
  // of the tile loop they would stay in registers through every K loop -- 32 spilled dwords)
  int lane_e = ln;
  asm volatile("" : "+v"(lane_e));
  const int li = lane_e & 31, lk = lane_e >> 5;
  const int32_t col = n0 + wn * 32 + li;
  // (a lambda instantiated on both paths, so that the accumulators of the fp32 path and those of the
  // planes never meet in one set of registers: merged behind a branch they cost 64 VGPRs of copies)
  // Row-major hand-over: a wave's 32 x 32 block goes through its own 4.5 KB of LDS (the A buffers
  // are free now), so C leaves -- and the residual arrives -- as 16-byte runs of a row (4 requests
  // per block and lane instead of 16 four-byte ones whose lanes touch two rows each).  Needs N,
  // ldc multiples of 4 and 16-byte aligned C / residual; otherwise the element-wise form.
  const bool vec = ((g.N | g.ldc) & 3) == 0 && ((reinterpret_cast<uintptr_t>(g.C) |
                   reinterpret_cast<uintptr_t>(g.residual)) & 15) == 0;
  auto epilogue = [&](auto wide_path, const f32x16(&sum)[SM], const f32x16(&cross)[SM]) {
    constexpr bool WIDE = decltype(wide_path)::value;
    if (!vec && col >= g.N) return;
    const bool col_ok = col < g.N;
    const float bv = (g.bias && col_ok) ? g.bias[col] : 0.f;
    const float cs = (LN && col_ok) ? g.ln_cs[col] : 0.f;
    const int32_t ew = (WIDE || !col_ok) ? 0 : ew_tab[col];
    // C and the residual through buffer descriptors: ONE 32-bit lane offset, the row of an
    // accumulator element is a wave-uniform (scalar) offset, rows past M fall outside the descriptor
    // (reads give zero, writes are dropped) -- no 64-bit address per element, no bounds test
    const uint32_t c_bytes = (uint32_t)(g.M * g.ldc * 4);
    auto rsrc_c = __builtin_amdgcn_make_buffer_rsrc(g.C, 0, c_bytes, 0x00020000);
    auto rsrc_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.residual), 0,
                                                    g.residual ? c_bytes : 0u, 0x00020000);
    const int32_t ldc_bytes = (int32_t)(g.ldc * 4);
    auto value = [&](int i, int e) {
      const int trow = (wr + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
      float v = WIDE ? sum[i][e]
                     : ldexpf(fmaf(cross[i][e], kLowDown, sum[i][e]), -(s_exp[trow] + ew));
      if (LN) v = s_stat[trow].y * (v - s_stat[trow].x * cs);
      v += bv;
      if (g.act == 1) v = fmaxf(v, 0.f);
      if (g.act == 2) v = v / (1.0f + __expf(-v));
      if (g.act == 3) v = 1.0f / (1.0f + __expf(-v));
      if (g.act == 4) v = tanhf(v);
      if (g.act == 5) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
      return v * g.alpha;
    };
    if (vec) {
      constexpr int TP = 36;  // floats per row of the hand-over block (16-byte aligned rows)
      float* tb = reinterpret_cast<float*>(s_a) + wv * (32 * TP);
      const int rr = lane_e >> 3, c4 = (lane_e & 7) * 4;  // row-major role: rows rr + 8 j, 4 columns
      const int32_t qcol = n0 + wn * 32 + c4;
      // (a quad past N aims outside the descriptor: N is a multiple of 4, quads do not straddle it)
      const int32_t vq = qcol < g.N ? (int32_t)(((int64_t)(m0 + wr * 32 + rr) * g.ldc + qcol) * 4)
                                    : (int32_t)0x7ffffff0;
#pragma unroll
      for (int i = 0; i < SM; ++i) {
        u32x4 rq[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          rq[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_r, vq, (i * 32 + 8 * j) * ldc_bytes, 0);
#pragma unroll
        for (int e = 0; e < 16; ++e) tb[((e & 3) + 8 * (e >> 2) + 4 * lk) * TP + li] = value(i, e);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (one wave: LDS serves it in order)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 q = *reinterpret_cast<const f32x4*>(tb + (rr + 8 * j) * TP + c4);
          u32x4 o;
#pragma unroll
          for (int c = 0; c < 4; ++c) o[c] = __float_as_uint(q[c] + __uint_as_float(rq[j][c]));
          __builtin_amdgcn_raw_buffer_store_b128(o, rsrc_c, vq, (i * 32 + 8 * j) * ldc_bytes, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the block is read before it is rewritten
      }
    } else {
      const int32_t vc = (int32_t)(((int64_t)(m0 + wr * 32 + 4 * lk) * g.ldc + col) * 4);
#pragma unroll
      for (int i = 0; i < SM; ++i) {
        float res[16];
#pragma unroll
        for (int e = 0; e < 16; ++e)
          res[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
              rsrc_r, vc, (i * 32 + (e & 3) + 8 * (e >> 2)) * ldc_bytes, 0));
#pragma unroll
        for (int e = 0; e < 16; ++e)
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(value(i, e) + res[e]), rsrc_c, vc,
                                                (i * 32 + (e & 3) + 8 * (e >> 2)) * ldc_bytes, 0);
      }
    }
  };
  if (tile_wide) {
    // The fp32 path: the tile once more on v_mfma_f32_32x32x2_f32 from the fp32 operands (exact
    // products, fp32 accumulation; rare, so plain: every lane fetches its own operand rows, 4 k per
    // request -- lanes 0-31 take k0 .. k0 + 3, lanes 32-63 k0 + 4 .. k0 + 7 -- and the MFMA pairs
    // element j of both halves: a permutation of k the sum does not see).
    if (tid == 0 && g.wide_count) atomicAdd(g.wide_count, 1);
    f32x16 sum[SM];
#pragma unroll
    for (int i = 0; i < SM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) sum[i][e] = 0.f;
    auto rsrc_a32 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A32), 0,
                                                      (uint32_t)(g.M * g.lda * 4), 0x00020000);
    auto rsrc_w32 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.W32), 0,
                                                      (uint32_t)(g.N * g.ldw * 4), 0x00020000);
    const int32_t wo = (int32_t)(min((int64_t)(n0 + wn * 32 + li), g.N - 1) * g.ldw * 4) + lk * 16;
    int32_t ao[SM];
#pragma unroll
    for (int i = 0; i < SM; ++i)
      ao[i] = (int32_t)(min((int64_t)(m0 + (wr + i) * 32 + li), g.M - 1) * g.lda * 4) + lk * 16;
#pragma unroll 2
    for (int32_t k0 = 0; k0 < (int32_t)g.K; k0 += 8) {
      const int32_t kq = k0 + 4 * lk;
      u32x4 wq = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w32, wo, k0 * 4, 0);
      u32x4 aq4[SM];
#pragma unroll
      for (int i = 0; i < SM; ++i) aq4[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a32, ao[i], k0 * 4, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool in = kq + j < g.K;
        const float wj = in ? __uint_as_float(wq[j]) : 0.f;
#pragma unroll
        for (int i = 0; i < SM; ++i) {
          const float aj = in ? __uint_as_float(aq4[i][j]) : 0.f;
          sum[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(aj, wj, sum[i], 0, 0, 0);
        }
      }
    }
    epilogue(std::true_type{}, sum, sum);
  } else {
    epilogue(std::false_type{}, acc, accx);
  }
#ifdef APS_FP16X2_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the tile's stores have left the wave)
  APS_WG_STAMP(3)
  if (tile < 4096 && tid == 0) {
    unsigned long long* t = g_fp16x2_wgtrace + (size_t)tile * 8;
    t[0] = blockIdx.x;
    t[1] = (unsigned long long)lin;
    t[2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
    t[3] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
    t[4] = wgstamp[0]; t[5] = wgstamp[1]; t[6] = wgstamp[2]; t[7] = wgstamp[3];
  }
#endif
#if APS_FP16X2_PERSISTENT
  __syncthreads();  // the next tile rewrites s_exp / s_stat / s_a
#endif
  }  // tile loop
}

// ------------------------------------------------------------------------------------------
// The channels-last convolution on the same arithmetic (conv_split_kernel's structure, gemm_split.hip:
// rows = output pixels, a K step = 32 input channels of one tap, weight fragments straight from the
// image of w [Co, KH KW Ci], the pixels' channel runs gathered by the staging threads).  A row of
// the implicit GEMM spans several input pixels, so its exponent is the smallest of the exponents of
// the input pixels its live taps read: one row_exp_kernel pass over x viewed as [N H W, Ci] gives
// every pixel's, the staging threads of a row fold the <= KH KW they need.
// ------------------------------------------------------------------------------------------
template <int WGN, int SM>
__global__ __launch_bounds__(256, 2) void conv_fp16x2_kernel(ConvArgs g, const void* image,
                                                            const int32_t* __restrict__ pixexp,
                                                            int32_t* __restrict__ wide_count) {
  constexpr int TM = (4 / WGN) * SM * 32, TN = WGN * 32;
  constexpr int kRowB = 64, kBuf = 2 * TM * kRowB, PA = TM / 32;
  __shared__ __attribute__((aligned(16))) unsigned char s_a[2 * kBuf];
  __shared__ int s_pix[TM];
  __shared__ int32_t s_exp[TM + 4];  // [TM] = "this tile takes the fp32 path"
  const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
  const int wm = wv / WGN, wn = wv % WGN;
  const int tiles_n = (g.Co + TN - 1) / TN;
  int64_t mt = blockIdx.x / tiles_n;
  const int n0 = (blockIdx.x % tiles_n) * TN;
  const int arow = tid >> 3, aq = tid & 7;
  const int asw = ((((aq >> 1) ^ ((arow >> 2) & 3)) << 4) | ((aq & 1) << 3));
  if (tid == 0) s_exp[TM] = 0;

  f32x16 acc[SM], accx[SM];  // main and cross sums (gemm_fp16x2_kernel)
#pragma unroll
  for (int i = 0; i < SM; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = accx[i][e] = 0.f;

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
  int rn[PA], rho[PA], rwo[PA], ea[PA];
  uint32_t fit[PA];
  bool rvalid[PA];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    fit[i] = 0;
    const int64_t m = m0 + arow + 32 * i;
    rvalid[i] = m < Mc;
    const int64_t mm = rvalid[i] ? m : 0;
    rwo[i] = qw + step_w * (int)(mm % Wc);
    rho[i] = qh + step_h * (int)((mm / Wc) % Hc);
    rn[i] = (int)(mm / ((int64_t)Wc * Hc));
    // the row's exponent: the smallest (= largest magnitude) among the input pixels of its live taps
    int e = 0x7fffffff;
    if (rvalid[i])
      for (int ih = 0; ih < nkh; ++ih)
        for (int iw = 0; iw < nkw; ++iw) {
          int hi, wi;
          if (tap_coord(rho[i], kh0 + step_h * ih, g.sh, g.ph, g.H, g.transposed, hi) &
              tap_coord(rwo[i], kw0 + step_w * iw, g.sw, g.pw, g.W, g.transposed, wi))
            e = min(e, pixexp[((int64_t)rn[i] * g.H + hi) * g.W + wi]);
        }
    ea[i] = e == 0x7fffffff ? 0 : e;  // (a row that only reads padding: all its operands are zero)
    if (aq == 0) {
      s_pix[arow + 32 * i] = rvalid[i] ? (rn[i] * g.Ho + rho[i]) * g.Wo + rwo[i] : -1;
      s_exp[arow + 32 * i] = ea[i];
    }
  }
  const int chunks = g.Ci / 32;
  const int ntiles = nkh * nkw * chunks;
  const uint32_t x_bytes = (uint32_t)((int64_t)g.N * g.H * g.W * g.Ci * 4);
  auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.x), 0, x_bytes, 0x00020000);
  const int64_t groups = ((g.Co + 127) / 128) * 4;
  const int32_t wstep_bytes = (int32_t)(groups * 4096);
  const int ksteps = g.KH * g.KW * chunks;
  auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(image), 0,
                                                  (uint32_t)(wstep_bytes * ksteps), 0x00020000);
  const int32_t vw = (int32_t)((n0 / 32 + wn) * 4096) + ln * 16;
  const int32_t* ew_tab = reinterpret_cast<const int32_t*>(static_cast<const unsigned char*>(image) +
                                                           (int64_t)wstep_bytes * ksteps);
  const bool wide_w = __any(ew_tab[groups * 32 + n0 + wn * 32 + (ln & 31)] != 0);

  u32x4 ra[PA];
  u32x4 wb[2][2][2];
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
      for (int p = 0; p < 2; ++p)
        wb[P][kk][p] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, vw, soff + (kk * 2 + p) * 1024, 0);
  };
  auto sstore = [&](int buf) {
    unsigned char* sA = s_a + buf * kBuf;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const float sc[4] = {ldexpf(__uint_as_float(ra[i].x), ea[i]), ldexpf(__uint_as_float(ra[i].y), ea[i]),
                           ldexpf(__uint_as_float(ra[i].z), ea[i]), ldexpf(__uint_as_float(ra[i].w), ea[i])};
#pragma unroll
      for (int e = 0; e < 4; ++e) fit[i] = max(fit[i], fit_key(sc[e]));
      u32x2 h, l;
      split4(sc, h, l);
      unsigned char* dst = sA + (arow + 32 * i) * kRowB + asw;
      *reinterpret_cast<u32x2*>(dst) = h;
      *reinterpret_cast<u32x2*>(dst + TM * kRowB) = l;
    }
  };
  const int frow = ln & 31, fsw = (frow >> 2) & 3;
  auto compute = [&](auto stage, int buf) {
    constexpr int P = decltype(stage)::value;
    const unsigned char* fa = s_a + buf * kBuf + (wm * SM * 32 + frow) * kRowB;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int off = ((kk * 2 + (ln >> 5)) ^ fsw) << 4;
      u32x4 a[SM][2];
#pragma unroll
      for (int i = 0; i < SM; ++i)
#pragma unroll
        for (int p = 0; p < 2; ++p)
          a[i][p] = *reinterpret_cast<const u32x4*>(fa + p * TM * kRowB + i * 32 * kRowB + off);
#pragma unroll
      for (int i = 0; i < SM; ++i) accx[i] = mfma_f16(a[i][0], wb[P][kk][1], accx[i]);
#pragma unroll
      for (int i = 0; i < SM; ++i) accx[i] = mfma_f16(a[i][1], wb[P][kk][0], accx[i]);
#pragma unroll
      for (int i = 0; i < SM; ++i) acc[i] = mfma_f16(a[i][0], wb[P][kk][0], acc[i]);
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
  }
  __syncthreads();  // s_pix, s_exp; every read of s_a is behind us
  if (wide_w || [&]() {
        bool bad = false;
#pragma unroll
        for (int i = 0; i < PA; ++i) bad |= fit[i] > kFitMax;
        return bad;
      }())
    s_exp[TM] = 1;
  __syncthreads();
  const bool tile_wide = s_exp[TM] != 0;
  const int li = ln & 31, lk = ln >> 5;
  if (tile_wide && ntiles > 0) {
    // the fp32 path (gemm_fp16x2_kernel): this wave's 32 SM x 32 block once more on
    // v_mfma_f32_32x32x2_f32, every lane gathering the channel runs of its own output pixels
    if (tid == 0 && wide_count) atomicAdd(wide_count, 1);
#pragma unroll
    for (int i = 0; i < SM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = accx[i][e] = 0.f;
    auto rsrc_w32 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.w), 0,
                                                      (uint32_t)((int64_t)g.Co * ksteps * 32 * 4), 0x00020000);
    const int32_t wo = (int32_t)((int64_t)min(n0 + wn * 32 + li, g.Co - 1) * ksteps * 32 * 4) + lk * 16;
    int pn[SM], pho[SM], pwo[SM];
    bool pv[SM];
#pragma unroll
    for (int i = 0; i < SM; ++i) {
      const int64_t m = m0 + wm * SM * 32 + i * 32 + li;
      pv[i] = m < Mc;
      const int64_t mm = pv[i] ? m : 0;
      pwo[i] = qw + step_w * (int)(mm % Wc);
      pho[i] = qh + step_h * (int)((mm / Wc) % Hc);
      pn[i] = (int)(mm / ((int64_t)Wc * Hc));
    }
    for (int ih = 0; ih < nkh; ++ih)
      for (int iw = 0; iw < nkw; ++iw) {
        const int kh = kh0 + step_h * ih, kw = kw0 + step_w * iw;
        uint32_t xo[SM];
#pragma unroll
        for (int i = 0; i < SM; ++i) {
          int hi, wi;
          const bool ok = tap_coord(pho[i], kh, g.sh, g.ph, g.H, g.transposed, hi) &
                          tap_coord(pwo[i], kw, g.sw, g.pw, g.W, g.transposed, wi) & pv[i];
          xo[i] = ok ? (uint32_t)((((int64_t)pn[i] * g.H + hi) * g.W + wi) * g.Ci + lk * 4) * 4u : 0xfffffff0u;
        }
        const int32_t wtap = (kh * g.KW + kw) * g.Ci * 4;
#pragma unroll 2
        for (int c0 = 0; c0 < g.Ci; c0 += 8) {
          const u32x4 wq = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w32, wo, wtap + c0 * 4, 0);
          u32x4 xq[SM];
#pragma unroll
          for (int i = 0; i < SM; ++i)
            xq[i] = xo[i] == 0xfffffff0u ? u32x4{0u, 0u, 0u, 0u}
                                         : __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, xo[i], c0 * 4, 0);
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < SM; ++i)
              acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(xq[i][j]), __uint_as_float(wq[j]),
                                                            acc[i], 0, 0, 0);
        }
      }
  }

  const int col = n0 + wn * 32 + li;
  if (col >= g.Co) return;
  const float sc_ = g.scale ? g.scale[col] : 1.f, sh_ = g.shift ? g.shift[col] : 0.f;
  const int32_t ew = ew_tab[col];
#pragma unroll
  for (int i = 0; i < SM; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int trow = wm * SM * 32 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
      const int64_t row = s_pix[trow];
      if (row < 0) continue;
      const float sum = ldexpf(fmaf(accx[i][e], kLowDown, acc[i][e]), tile_wide ? 0 : -(s_exp[trow] + ew));
      float v = conv_act(sum * sc_ + sh_, g.act, g.slope);
      if (g.residual) v += g.residual[row * g.Co + col];
      g.y[row * g.Co + col] = v;
    }
}

// workgroups the chip holds at once: CUs x the workgroups per CU the kernel is compiled for
static ApsPerDevice g_fp16x2_slots;
static int64_t fp16x2_resident_slots() {
  const int dev = aps_current_device();
  if (dev < 0) return 256 * APS_FP16X2_MIN_WG;
  int n = g_fp16x2_slots.get(dev);
  if (!n) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    n = cus * APS_FP16X2_MIN_WG;
    g_fp16x2_slots.set(dev, n);
  }
  return n;
}

// at most this many 64 x 128 tiles -> the 64 x 64 form (twice the workgroups); APS_GEMM_NARROW_TILES=0: off
static int64_t fp16x2_narrow_tiles() {
  static const int64_t v = [] {
    const char* e = getenv("APS_GEMM_NARROW_TILES");
    return e ? (int64_t)atoll(e) : (int64_t)400;
  }();
  return v;
}

template <bool LN>
static int launch_fp16x2(Fp16GemmArgs g, hipStream_t st) {
  const int64_t tiles_m = g.Mp / 64;
  const bool narrow = tiles_m * ((g.N + 127) / 128) <= fp16x2_narrow_tiles();
  const int64_t tiles_n = narrow ? (g.N + 63) / 64 : (g.N + 127) / 128;
  const int64_t total = tiles_m * tiles_n;
  if (total > 0x7fffffff) return APS_ERR_INVALID;
  g.tiles_n = (int32_t)tiles_n;
  g.total = (int32_t)total;
  int64_t grid = total;
#if APS_FP16X2_PERSISTENT
  const int64_t slots = fp16x2_resident_slots();
  if (grid > slots) grid = slots;
#endif
  g.remap = (total % 8 == 0 && grid % 8 == 0) ? 1 : 0;
  if (narrow)
    hipLaunchKernelGGL((gemm_fp16x2_kernel<LN, true>), dim3((unsigned)grid), dim3(256), 0, st, g);
  else
    hipLaunchKernelGGL((gemm_fp16x2_kernel<LN, false>), dim3((unsigned)grid), dim3(256), 0, st, g);
  return aps_launch_status();
}

static int launch_row_exp(const float* X, int32_t* e, int64_t rows, int64_t K, int64_t ldx,
                          hipStream_t st) {
  hipLaunchKernelGGL(row_exp_kernel, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, st, X, e, rows, K,
                     ldx);
  return aps_launch_status();
}

}  // namespace aps

using namespace aps;

#ifdef APS_FP16X2_TRACE
extern "C" int aps_debug_fp16x2_wgtrace(void* host, int64_t bytes) {
  if (bytes > (int64_t)sizeof(g_fp16x2_wgtrace)) bytes = sizeof(g_fp16x2_wgtrace);
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_fp16x2_wgtrace), (size_t)bytes) == hipSuccess ? APS_OK : APS_ERR_LAUNCH;
}
extern "C" int aps_debug_fp16x2_trace(void* host, int64_t bytes) {
  if (bytes > (int64_t)sizeof(g_fp16x2_trace)) bytes = sizeof(g_fp16x2_trace);
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_fp16x2_trace), (size_t)bytes) == hipSuccess ? APS_OK : APS_ERR_LAUNCH;
}
#endif

extern "C" int64_t aps_linear_fp16x2_size(int64_t N, int64_t K) {
  if (N <= 0 || K <= 0) return 0;
  const int64_t np = ((N + 127) / 128) * 128;
  return np * ((K + 31) / 32) * 128 + np * 8;  // fragments, int32 exponents, int32 wide flags
}

extern "C" int aps_linear_fp16x2_weight(const float* W, void* image, int64_t N, int64_t K,
                                        int64_t ldw, void* stream) {
  APS_CHECK_ARG(W && image && N > 0 && K > 0 && ldw >= K);
  APS_CHECK_ARG(((uintptr_t)image & 15) == 0 && ((uintptr_t)W & 15) == 0 && ldw % 4 == 0);
  const int64_t np = ((N + 127) / 128) * 128, ksteps = (K + 31) / 32;
  if (aps_linear_fp16x2_size(N, K) >= ((int64_t)1 << 31)) return APS_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int32_t* ew = reinterpret_cast<int32_t*>(static_cast<unsigned char*>(image) + np * ksteps * 128);
  int rc = aps_fill_u32(ew, 0u, (size_t)(2 * np), st);  // exponents of the padding columns, all flags
  if (rc != APS_OK) return rc;
  rc = launch_row_exp(W, ew, N, K, ldw, st);
  if (rc != APS_OK) return rc;
  const int64_t groups = np / 32, threads = ksteps * groups * 128;
  hipLaunchKernelGGL(fp16x2_weight_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, W,
                     reinterpret_cast<u32x4*>(image), ew, ew + np, N, K, ldw, groups, ksteps);
  return aps_launch_status();
}

// workspace of one call: the planes image of A, its row information, its row statistics
static void fp16x2_workspace_layout(int64_t M, int64_t K, int64_t& Mp, int64_t& planes, int64_t& info,
                                    int64_t& stat, int64_t& total) {
  Mp = ((M + 63) / 64) * 64;
  planes = 0;
  info = Mp * ((K + 31) / 32) * 128;
  stat = info + Mp * 4;
  total = stat + Mp * 8;
}

extern "C" int64_t aps_linear_fp16x2_workspace(int64_t M, int64_t K) {
  if (M <= 0 || K <= 0) return 0;
  int64_t Mp, planes, info, stat, total;
  fp16x2_workspace_layout(M, K, Mp, planes, info, stat, total);
  return total;
}

extern "C" int aps_linear_fp16x2(const float* A, const void* image, const float* W32, const float* bias,
                                 const float* colsum, const float* residual, float* C,
                                 void* workspace, int32_t* wide_count, int64_t M, int64_t N,
                                 int64_t K, int64_t lda, int64_t ldw, int64_t ldc, int32_t act,
                                 float alpha, float eps, void* stream) {
  APS_CHECK_ARG(A && image && W32 && C && workspace && M > 0 && N > 0 && K > 0);
  APS_CHECK_ARG(lda >= K && ldc >= N && lda % 4 == 0 && ((uintptr_t)A & 15) == 0 &&
                ((uintptr_t)image & 15) == 0 && ((uintptr_t)workspace & 15) == 0);
  APS_CHECK_ARG(ldw >= K && ldw % 4 == 0 && ((uintptr_t)W32 & 15) == 0);
  APS_CHECK_ARG(act >= 0 && act <= 5);
  int64_t Mp, planes, info, stat, total;
  fp16x2_workspace_layout(M, K, Mp, planes, info, stat, total);
  if (M * lda * 4 >= ((int64_t)1 << 31) || N * ldw * 4 >= ((int64_t)1 << 31) ||
      M * ldc * 4 >= ((int64_t)1 << 31) || aps_linear_fp16x2_size(N, K) >= ((int64_t)1 << 31) ||
      info >= ((int64_t)1 << 31))
    return APS_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  unsigned char* ws = static_cast<unsigned char*>(workspace);
  const int64_t ksteps = (K + 31) / 32;
  {
    u32x4* pl = reinterpret_cast<u32x4*>(ws + planes);
    int32_t* ri = reinterpret_cast<int32_t*>(ws + info);
    float2* sp = reinterpret_cast<float2*>(ws + stat);
    const dim3 grid((unsigned)(Mp / 16)), block(256);
    const int64_t chunks = (ksteps * 32 + 127) / 128;  // 128-element chunks of a row
    if (chunks <= 2)
      hipLaunchKernelGGL(fp16x2_split_kernel<2>, grid, block, 0, st, A, pl, ri, sp, M, Mp, K, ksteps, lda);
    else if (chunks <= 4)
      hipLaunchKernelGGL(fp16x2_split_kernel<4>, grid, block, 0, st, A, pl, ri, sp, M, Mp, K, ksteps, lda);
    else if (chunks <= 8)
      hipLaunchKernelGGL(fp16x2_split_kernel<8>, grid, block, 0, st, A, pl, ri, sp, M, Mp, K, ksteps, lda);
    else
      hipLaunchKernelGGL(fp16x2_split_kernel<0>, grid, block, 0, st, A, pl, ri, sp, M, Mp, K, ksteps, lda);
  }
  int rc = aps_launch_status();
  if (rc != APS_OK) return rc;
  Fp16GemmArgs g{ws + planes, reinterpret_cast<const int32_t*>(ws + info),
                 reinterpret_cast<const float2*>(ws + stat), A, image, W32, bias, residual, C, wide_count,
                 M, Mp, N, K, lda, ldw, ldc, act, alpha, 0, 0, (int32_t)ksteps, 0, colsum, eps};
  return colsum ? launch_fp16x2<true>(g, st) : launch_fp16x2<false>(g, st);
}

extern "C" int aps_conv2d_nhwc_fp16x2(const float* x, const void* image, const float* w32,
                                      const float* scale, const float* shift, const float* residual,
                                      float* y, int32_t* pixexp, int32_t* wide_count, int64_t N,
                                      int64_t H, int64_t W, int64_t Ci, int64_t Co, int64_t KH,
                                      int64_t KW, int64_t sh, int64_t sw, int64_t ph, int64_t pw,
                                      int64_t Ho, int64_t Wo, int32_t transposed, int32_t act,
                                      float slope, void* stream) {
  APS_CHECK_ARG(x && image && w32 && y && pixexp && N > 0 && H > 0 && W > 0 && Ci > 0 && Co > 0 && KH > 0 &&
                KW > 0);
  APS_CHECK_ARG(sh > 0 && sw > 0 && ph >= 0 && pw >= 0 && Ho > 0 && Wo > 0);
  APS_CHECK_ARG(act == 0 || act == 1 || act == 5);
  APS_CHECK_ARG(Ci % 32 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)image & 15) == 0 &&
                ((uintptr_t)w32 & 15) == 0);
  const int64_t M = N * Ho * Wo;
  if (N * H * W * Ci * 4 >= ((int64_t)1 << 32) - 64 || M >= ((int64_t)1 << 31) ||
      Co * KH * KW * Ci * 4 >= ((int64_t)1 << 31) ||
      aps_linear_fp16x2_size(Co, KH * KW * Ci) >= ((int64_t)1 << 31))
    return APS_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int rc = launch_row_exp(x, pixexp, N * H * W, Ci, Ci, st);  // every input pixel's exponent
  if (rc != APS_OK) return rc;
  ConvArgs g{x, w32, scale, shift, residual, y, (int32_t)N, (int32_t)H, (int32_t)W, (int32_t)Ci,
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
  if (tn == 128)
    hipLaunchKernelGGL((conv_fp16x2_kernel<4, 2>), dim3((unsigned)tiles), dim3(256), 0, st, g, image, pixexp,
                       wide_count);
  else if (tn == 64)
    hipLaunchKernelGGL((conv_fp16x2_kernel<2, 2>), dim3((unsigned)tiles), dim3(256), 0, st, g, image, pixexp,
                       wide_count);
  else
    hipLaunchKernelGGL((conv_fp16x2_kernel<1, 1>), dim3((unsigned)tiles), dim3(256), 0, st, g, image, pixexp,
                       wide_count);
  return aps_launch_status();
}
