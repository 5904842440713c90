// The conformer encoder stack as ONE launch per batch: a workgroup owns an UTTERANCE (T <= 64 encoder frames = one
// 64-row tile) for all layers (aps_conformer_stack; round 6).
//
// Why.  At BASELINE configs[4]'s per-GPU share (32 utterances x 63 encoder frames: M = 2016 rows) the encoder is ~130
// dependent launches of 17 - 20 us each for 0.4 - 3 us of matrix work: a launch of 252 - 756 tiles of 32 x 128 lives
// exactly as long as ONE of its workgroups -- request the rows, split them, 4 - 8 chunks of round trips for the weight
// fragments, drain the stores -- and the round's traces (profiles/r06_panel_trace_under_load.txt) show that no tile
// shape, ring depth or stream placement moves the pipelined step off 2.05 - 2.10 ms: what the chip runs out of is
// workgroup SLOTS x lifetime (128 - 250 registers held for 25 - 50 us per 3 us of MFMAs), not any pipe.  Here the
// dependency chain of a conformer layer lives inside a workgroup, where it costs a barrier instead of a launch:
//   * activations never leave the CU's reach: the residual stream X (the layer input / output, [T, D] rows of the
//     caller's tensor), the FFN / GLU hidden H, QKV and the attention / convolution outputs are fp32 rows of a
//     per-workgroup scratch that stays in this XCD's L2; every projection's input is split ONCE into the two f16
//     planes of a [64, 512] LDS image (133 KB) that all eight waves read;
//   * a projection is a loop over its 32-column blocks, dealt round-robin to the eight waves: each wave streams ITS
//     weight fragments from the cached image (aps_linear_fp16x2_weight: the same image, arithmetic, detection rule
//     and fp32 recomputation as csrc/gemm_panel.hip; a power of two per row of the 512-wide phase) through a four-stage register ring
//     that never drains between blocks, 12 MFMAs per 4 KB of fragments (64 rows: half the bytes per product of the
//     32-row tiles), no barrier inside a projection;
//   * K = 1024 (the FFN's second projection) runs as two K = 512 phases chained through the residual operand;
//   * relative-position attention (two heads at a time, the arithmetic of nn.hip's attention_small_kernel<64, true>)
//     and GLU . depthwise conv . BatchNorm . swish are phases of the same workgroup.
// Reference: aps/asr/transformer/impl.py:432-541 (ConformerEncoderLayer, pre-norm), :225-296 (relative attention),
// :718-756 (the stack).  Host side: aps_amd/mega.py builds the per-layer tables; the per-launch path of
// aps_amd/asr/transformer/impl.py stays the library default and the oracle-checked twin of this kernel.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "common.h"

namespace aps {
namespace mega {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr float kLowUp = 2048.f, kLowDown = 1.0f / 2048.f;
constexpr int kFitBias = 15;

#ifndef APS_MEGA_W_AUX
#define APS_MEGA_W_AUX 0   // cache policy bits of the weight-fragment requests (experiments: 2 = nt)
#endif
constexpr int RT = 64;                 // rows of an utterance tile
constexpr int NT = 512, NWAVES = 8;    // threads, waves of a workgroup
constexpr int KP = 512;                // contraction of one projection phase
constexpr int PB = KP * 2 + 16;        // row pitch of a plane of the LDS image (bytes)
constexpr int PLANE = RT * PB, IMG = 2 * PLANE;
constexpr int kAttPitch = 68;
constexpr int MAIN = 152 * 1024;                           // the image (133 KB) | the attention phase's regions (149 KB) overlay it
constexpr int LDS_BYTES = MAIN + RT * 4 + RT * 8 + 64;     // + row exponents, row statistics, flags

__device__ __forceinline__ int32_t scale_exponent(float mx) {
  int be = (int)((__float_as_uint(mx) >> 23) & 0xffu);
  be = be < 1 ? 1 : (be > 254 ? 254 : be);
  return 141 - be;
}
__device__ __forceinline__ f32x16 mfma_f16(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<N, I + 1>(f);
  }
}
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// maximum / sum over an aligned group of 8 lanes, in every lane (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror)
__device__ __forceinline__ float group_max8(float v) {
  v = fmaxf(v, dpp_move<0xB1>(v));
  v = fmaxf(v, dpp_move<0x4E>(v));
  return fmaxf(v, dpp_move<0x141>(v));
}
__device__ __forceinline__ float group_sum8(float v) {
  v += dpp_move<0xB1>(v);
  v += dpp_move<0x4E>(v);
  return v + dpp_move<0x141>(v);
}

// A value every lane of the wave holds, moved to scalar registers dword by dword.  (The layer table lives in global
// memory the kernel also stores to, so the compiler cannot prove its loads unclobbered and issues them as VECTOR loads;
// a buffer descriptor built from vector registers then costs a waterfall loop per request.)
template <typename T>
__device__ __forceinline__ T uniform(const T& v) {
  static_assert(sizeof(T) % 4 == 0, "whole dwords");
  T out;
  const uint32_t* src = reinterpret_cast<const uint32_t*>(&v);
  uint32_t* dst = reinterpret_cast<uint32_t*>(&out);
#pragma unroll
  for (size_t i = 0; i < sizeof(T) / 4; ++i) dst[i] = __builtin_amdgcn_readfirstlane(src[i]);
  return out;
}

// threadIdx.x as a value the optimiser cannot see through: every phase derives its lane constants (rows, offsets,
// LDS addresses) from it INSIDE the phase loop; computed from the plain built-in they are loop invariant, get hoisted in
// front of the loop -- all phases' at once -- and live (or spill) across every phase.
__device__ __forceinline__ int lane_id_here() {
  int t = threadIdx.x;
#ifndef APS_MEGA_PLAIN_TID
  asm volatile("" : "+v"(t));
#endif
  return t;
}

// One projection phase: C[64, N] = epilogue(A[64, 512] W^T) on the weight image of W [N, K_total], K steps
// kstep0 .. kstep0 + 15.  (Mirrors ApsMegaGemm of include/aps_amd.h field by field.)
struct Gemm {
  const void* image;     // aps_linear_fp16x2_weight image of the (LayerNorm-folded) weight
  const float* w32;      // the fp32 weight the image was made from, row pitch ldw: the fp32 path
  const float* bias;     // [N] or null
  const float* colsum;   // LayerNorm fold: column sums of W diag(gamma), or null
  int32_t N, ksteps_total, kstep0, ldw;
  int32_t act, pad0;
  float alpha, ln_eps;
};
struct Layer {
  Gemm ff1_up, ff1_dn0, ff1_dn1, qkv, out, pw1, pw2, ff2_up, ff2_dn0, ff2_dn1;
  const float* dw_w;       // depthwise weights [D, 15]
  const float* dw_b;       // [D] or null
  const float* bn_scale;   // eval-mode BatchNorm as scale / shift [D], or null
  const float* bn_shift;
  int32_t conv_act, pad1;
};
struct ConvParams {   // (the tail of Layer)
  const float* dw_w;
  const float* dw_b;
  const float* bn_scale;
  const float* bn_shift;
  int32_t conv_act, pad1;
};
struct Args {
  float* x;                // [N, T, D] in place: the residual stream
  const int64_t* lens;     // [N] valid frames or null
  const Layer* layers;
  float* scratch;          // N x scratch_floats
  int32_t* wide_count;
  const float* rel;        // relative position table [rel_len, 64], shared by the layers (the encoder's inj_pose)
  int64_t rel_zero, rel_len;
  int64_t scratch_floats;
  int32_t num_layers, T, D, FF, heads, pad2;
  float att_scale;
  unsigned long long* trace;   // experiments (APS_MEGA_TRACE=1): [32] cycles of workgroup 0 per phase kind, or null
};

struct Smem {
  unsigned char* main;   // image | attention regions
  int32_t* exps;         // [64 rows]: the power of two of a staged row
  float2* stat;          // [64] (mean, 1 / sqrt(var + eps)) of the staged rows
  int32_t* flags;        // [0] an element of the staged rows does not fit its scale
};

// ---- stage: rows [T, 512] fp32 (row pitch ld) -> the two planes of the LDS image, a power of two per ROW (the whole
// 512-wide phase accumulates in one pair of accumulators); the raw rows' LayerNorm statistics ride along.  8 lanes per
// row, a lane owns 16 floats of every 128-chunk.
template <bool LN>
__device__ __forceinline__ bool stage_rows(const Smem& sm, const float* __restrict__ src, int64_t ld, int T, float ln_eps) {
  const int tid = lane_id_here();
  const int row = tid >> 3, q = tid & 7;
  // (a descriptor over exactly T rows: the rows beyond read as zeros)
  auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, (uint32_t)(((int64_t)(T - 1) * ld + KP) * 4), 0x00020000);
  const uint32_t voff = (uint32_t)(((int64_t)row * ld + q * 8) * 4);
  f32x4 v[4][2][2];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        v[c][j][h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, row < T ? voff : 0x80000000u,
                                                                                    (c * 128 + j * 64 + h * 4) * 4, 0));
  float s1 = 0.f, s2 = 0.f, mx = 0.f;
  int32_t fitmin = 0;
  unsigned char* const dst = sm.main + row * PB + q * 16;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x = v[c][j][h][e];
          mx = fmaxf(mx, fabsf(x));
          if constexpr (LN) {
            s1 += x;
            s2 = fmaf(x, x, s2);
          }
        }
  mx = group_max8(mx);
  const int32_t ex = scale_exponent(mx);
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      _Float16 hh[8], ll[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float x = v[c][j][e >> 2][e & 3];
        const float sc = ldexpf(x, ex);
        fitmin = min(fitmin, __builtin_amdgcn_frexp_expf(sc));
        hh[e] = (_Float16)sc;
        ll[e] = (_Float16)fmaf((float)hh[e], -kLowUp, ldexpf(x, ex + 11));
      }
      u32x4 h, l;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        h[e] = __builtin_bit_cast(uint32_t, f16x2{hh[2 * e], hh[2 * e + 1]});
        l[e] = __builtin_bit_cast(uint32_t, f16x2{ll[2 * e], ll[2 * e + 1]});
      }
      // k = 128 c + 64 j + 8 q .. + 7  ->  byte 2 k of the row
      *reinterpret_cast<u32x4*>(dst + c * 256 + j * 128) = h;
      *reinterpret_cast<u32x4*>(dst + PLANE + c * 256 + j * 128) = l;
    }
  if (q == 0) sm.exps[row] = ex;
  if constexpr (LN) {
    s1 = group_sum8(s1);
    s2 = group_sum8(s2);
    if (q == 0) {
      const float mean = s1 * (1.0f / KP);
      const float var = fmaxf(s2 * (1.0f / KP) - mean * mean, 0.f);
      sm.stat[row] = make_float2(mean, 1.0f / sqrtf(var + ln_eps));
    }
  }
  return fitmin < -kFitBias;   // an element of this lane's share does not fit the row's scale
}

// ---- a projection phase on the staged image.  Wave wv takes the 32-column blocks wv, wv + 8, ...; no barrier inside.
// src32 / ld_src: the fp32 rows the image was staged from (the fp32 path re-reads them); residual / dst rows of T frames.
__device__ __forceinline__ void gemm_phase(const Smem& sm, const Gemm& g, const float* __restrict__ src32, int ld_src,
                                           const float* residual, int ld_res, float* dst, int ld_dst, int T,
                                           int32_t* wide_count, bool wide_a) {
  const int tid = lane_id_here(), ln = tid & 63;
  const bool has_ln = g.colsum != nullptr;
  const int act = g.act;
  const float alpha = g.alpha;
  auto rsrc_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src32), 0, (uint32_t)(((T - 1) * ld_src + KP) * 4), 0x00020000);
  auto rsrc_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(residual), 0,
                                                  residual ? (uint32_t)(T * ld_res * 4) : 0u, 0x00020000);
  auto rsrc_d = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (uint32_t)(T * ld_dst * 4), 0x00020000);
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = ln & 31, lk = ln >> 5;
  const int nblk_all = g.N >> 5;
  const int my = (nblk_all - wv + NWAVES - 1) / NWAVES;   // blocks of this wave
  if (my <= 0) return;
  const int64_t groups = ((g.N + 127) / 128) * 4;
  const int32_t wstep_bytes = (int32_t)(groups * 4096);
  auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.image), 0,
                                                  (uint32_t)((int64_t)wstep_bytes * g.ksteps_total), 0x00020000);
  const int32_t* ew_tab = reinterpret_cast<const int32_t*>(static_cast<const unsigned char*>(g.image) +
                                                           (int64_t)wstep_bytes * g.ksteps_total);
  u32x4 wb[4][2][2];   // [ring stage][MFMA K step][plane]
  const int32_t last_lin = my * 16 - 1;
  auto load_stage = [&](auto stage, int32_t lin) {   // step `lin` of this wave's (block, K step) sequence, clamped
    constexpr int P = decltype(stage)::value;
    lin = lin < last_lin ? lin : last_lin;
    const int32_t blk = wv + NWAVES * (lin >> 4);
    const int32_t voff = blk * 4096 + ln * 16;
    const int32_t soff = (g.kstep0 + (lin & 15)) * wstep_bytes;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int p = 0; p < 2; ++p)
        wb[P][kk][p] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, voff, soff + (kk * 2 + p) * 1024, APS_MEGA_W_AUX);
  };
  static_for<3>([&](auto sc) { load_stage(sc, decltype(sc)::value); });
  const unsigned char* const frag = sm.main + li * PB + lk * 16;

  for (int bi = 0; bi < my; ++bi) {
    const int32_t blk = wv + NWAVES * bi;
    const int32_t col = blk * 32 + li;
    const int32_t ew = ew_tab[col];
    const int32_t ew_flag = ew_tab[groups * 32 + col];
    const float bv = g.bias ? g.bias[col] : 0.f;
    const float cs = g.colsum ? g.colsum[col] : 0.f;
    f32x16 acc[2], accx[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = accx[i][e] = 0.f;
    // The A fragments of K step ks + 1 are requested from LDS in front of step ks's MFMAs (two register sets), and a
    // scheduling barrier closes every step: left alone, the compiler hoists the fragment reads of ALL sixteen unrolled
    // steps to the top (512 registers' worth: hundreds of spills, whose scratch traffic then drains the weight ring).
    u32x4 fr[2][2][2][2];   // [set][MFMA K step][row block][plane]
    auto load_frags = [&](auto setc, int ks) {
      constexpr int S = decltype(setc)::value;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const unsigned char* fa = frag + i * 32 * PB + ks * 64 + kk * 32;
          fr[S][kk][i][0] = *reinterpret_cast<const u32x4*>(fa);
          fr[S][kk][i][1] = *reinterpret_cast<const u32x4*>(fa + PLANE);
        }
    };
    load_frags(std::integral_constant<int, 0>{}, 0);
    static_for<16>([&](auto ksc) {
      constexpr int ks = decltype(ksc)::value;
      constexpr int P = ks % 4, PN = (ks + 3) % 4, S = ks % 2;
      load_stage(std::integral_constant<int, PN>{}, bi * 16 + ks + 3);
      if constexpr (ks + 1 < 16) load_frags(std::integral_constant<int, 1 - S>{}, ks + 1);
      // (two MFMAs on other accumulators between the two that share the cross accumulator of a row block)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        accx[0] = mfma_f16(fr[S][kk][0][0], wb[P][kk][1], accx[0]);  // h l
        accx[1] = mfma_f16(fr[S][kk][1][0], wb[P][kk][1], accx[1]);
        acc[0] = mfma_f16(fr[S][kk][0][0], wb[P][kk][0], acc[0]);    // h h
        acc[1] = mfma_f16(fr[S][kk][1][0], wb[P][kk][0], acc[1]);
        accx[0] = mfma_f16(fr[S][kk][0][1], wb[P][kk][0], accx[0]);  // l h
        accx[1] = mfma_f16(fr[S][kk][1][1], wb[P][kk][0], accx[1]);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    // fold: 2^-(ea[row] + ew[col]) (main + 2^-11 cross)
    f32x16 sum[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const i32x4 ea = *reinterpret_cast<const i32x4*>(&sm.exps[i * 32 + 8 * r4 + 4 * lk]);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          sum[i][r4 * 4 + e] = ldexpf(fmaf(accx[i][r4 * 4 + e], kLowDown, acc[i][r4 * 4 + e]), -(ea[e] + ew));
      }
    // ---- the fp32 path (csrc/gemm_panel.hip's): an operand of this block does not fit its scale
    if (__any(wide_a || ew_flag != 0)) {
      if (ln == 0 && wide_count) atomicAdd(wide_count, 1);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) sum[i][e] = 0.f;
      const float* wr = g.w32 + (int64_t)col * g.ldw + g.kstep0 * 32 + lk * 4;
#pragma unroll 2
      for (int k0 = 0; k0 < KP; k0 += 8) {
        const f32x4 wq = *reinterpret_cast<const f32x4*>(wr + k0);
        f32x4 aq[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
          aq[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                rsrc_s, (uint32_t)(((int64_t)(i * 32 + li) * ld_src + k0 + lk * 4) * 4), 0, 0));
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i)
            sum[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[i][j], wq[j], sum[i], 0, 0, 0);
      }
    }
    // ---- epilogue: LayerNorm fold, bias, activation, alpha, residual; a lane owns 16 rows of one column per row block.
    // ONE scalar dispatch per block on (activation, LayerNorm) into straight-line code: with the activation chain
    // inside the element loop the block's epilogue was ~3 900 instructions of per-element branches (IEEE division,
    // tanh and erf expansions behind exec masks) and cost 275 k of a layer's 775 k cycles.
    // (descriptors over exactly T rows: the residual of a row beyond reads as zero, its store is dropped; the row
    // part of an element's offset is a compile-time multiple of the pitch: scalar arithmetic)
    const uint32_t off_r = (uint32_t)((4 * lk * ld_res + col) * 4), off_d = (uint32_t)((4 * lk * ld_dst + col) * 4);
    auto finish = [&](auto actc, auto lnc) {
      constexpr int ACT = decltype(actc)::value;
      constexpr bool LN = decltype(lnc)::value;
      // (requested here, not in front of the fold: measured the same, and two registers spill there)
      float res[2][16];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          res[i][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
              rsrc_r, off_r, (i * 32 + (e & 3) + 8 * (e >> 2)) * ld_res * 4, 0));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
          float v = sum[i][e];
          if constexpr (LN) {
            const float2 st = sm.stat[row];
            v = st.y * (v - st.x * cs);
          }
          v += bv;
          if constexpr (ACT == 1) v = fmaxf(v, 0.f);
          if constexpr (ACT == 2) v = v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
          if constexpr (ACT == 3) v = __builtin_amdgcn_rcpf(1.0f + __expf(-v));
          if constexpr (ACT == 4) v = tanhf(v);
          if constexpr (ACT == 5) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
          v = fmaf(v, alpha, res[i][e]);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsrc_d, off_d,
                                                (i * 32 + (e & 3) + 8 * (e >> 2)) * ld_dst * 4, 0);
        }
    };
    using std::integral_constant;
    using std::true_type;
    using std::false_type;
    if (has_ln) {
      if (act == 2) finish(integral_constant<int, 2>{}, true_type{});
      else if (act == 0) finish(integral_constant<int, 0>{}, true_type{});
      else if (act == 1) finish(integral_constant<int, 1>{}, true_type{});
      else if (act == 5) finish(integral_constant<int, 5>{}, true_type{});
      else if (act == 3) finish(integral_constant<int, 3>{}, true_type{});
      else finish(integral_constant<int, 4>{}, true_type{});
    } else {
      if (act == 0) finish(integral_constant<int, 0>{}, false_type{});
      else if (act == 2) finish(integral_constant<int, 2>{}, false_type{});
      else if (act == 1) finish(integral_constant<int, 1>{}, false_type{});
      else if (act == 5) finish(integral_constant<int, 5>{}, false_type{});
      else if (act == 3) finish(integral_constant<int, 3>{}, false_type{});
      else finish(integral_constant<int, 4>{}, false_type{});
    }
  }
}

// ---- relative-position self attention of one utterance, two heads at a time (waves 0-3 | 4-7), T <= 64:
// logits = (q k^T + shift(q E^T)) / sqrt(dh), key padding by `len` (aps/asr/transformer/impl.py:225-296).
// Round 6, second form: the three products (q k^T, q E^T, p v) on the f16 matrix pipe with the projections' two-plane
// arithmetic -- 48 MFMAs of 32 x 32 x 16 per wave and head instead of 128 exact-fp32 ones at a sixteenth of the rate
// (attention was 111 k of a layer's 800 k cycles, two thirds of it the fp32 MFMAs and their LDS operand reads):
//   * q / sqrt(dh), k and the table window E: planes [row][64 k] with a power of two per ROW (4 lanes per row, a quad
//     reduction); E's planes are staged once per phase, all heads read them;
//   * v row j carries its own power of two 2^ev[j]; the probabilities absorb it (p'[i][j] = p[i][j] 2^-ev[j], exact) and
//     take a power of two per query row: o = 2^-ep[i] (p'_h v'_h + 2^-11 cross);
//   * scores: fp32 [64][68] over the dead q planes; the shifted term is added in place (one owner per (i, j));
//     softmax with four lanes per row.
// Error: every product within 2^-21 of sum |a||b| like the projections (tests/test_gpu_mega.py holds the layer output).
// LDS (148 KB of the 160): E planes [2][128][144 B] | per group: R1 q planes (later the scores) | R2 k planes (later the
// probabilities' planes) | R3 v^T planes | exponents.
constexpr int APB = 64 * 2 + 16;                 // row pitch of a 64-wide plane (bytes): b128 reads conflict free
constexpr int APLANE64 = 64 * APB, AIMG64 = 2 * APLANE64;      // 9 216 / 18 432: a 64-row matrix' planes
constexpr int AE_BYTES = 2 * 128 * APB;                        // 36 864: the window's planes
constexpr int AG_BYTES = 3 * AIMG64 + 4 * 64 * 4;              // a group's regions + eq | ek | ev | ep
constexpr int ATT_BYTES = AE_BYTES + 128 * 4 + 2 * AG_BYTES;   // + ee[128]
static_assert(64 * kAttPitch * 4 <= AIMG64, "the scores fit the q planes' region");
static_assert(ATT_BYTES <= MAIN && IMG <= MAIN, "the regions fit the main LDS region");

// 16 consecutive floats of a row held by one lane (4 lanes per row) -> the row's power of two, both planes
__device__ __forceinline__ int32_t att_split16(const f32x4 (&v)[4], float post, unsigned char* dst_h, int plane_bytes) {
  float mx = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) mx = fmaxf(mx, fabsf(v[c][e] * post));
  mx = fmaxf(mx, dpp_move<0xB1>(mx));
  mx = fmaxf(mx, dpp_move<0x4E>(mx));
  const int32_t ex = scale_exponent(mx);
#pragma unroll
  for (int hv = 0; hv < 2; ++hv) {
    _Float16 hh[8], ll[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = v[hv * 2 + (e >> 2)][e & 3] * post;
      hh[e] = (_Float16)ldexpf(x, ex);
      ll[e] = (_Float16)fmaf((float)hh[e], -kLowUp, ldexpf(x, ex + 11));
    }
    u32x4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      h[e] = __builtin_bit_cast(uint32_t, f16x2{hh[2 * e], hh[2 * e + 1]});
      l[e] = __builtin_bit_cast(uint32_t, f16x2{ll[2 * e], ll[2 * e + 1]});
    }
    *reinterpret_cast<u32x4*>(dst_h + hv * 16) = h;
    *reinterpret_cast<u32x4*>(dst_h + plane_bytes + hv * 16) = l;
  }
  return ex;
}

__device__ __forceinline__ void attention_phase(const Smem& sm, const float* __restrict__ qkv, float* __restrict__ ctx,
                                                const float* __restrict__ rel, int64_t rel_zero, int64_t rel_len, int T,
                                                int len, int H, int D, float scale) {
  constexpr int DH = 64, VP = kAttPitch;
  const int tid_all = lane_id_here();
  const int tid = tid_all & 255, grp = __builtin_amdgcn_readfirstlane(tid_all >> 8);
  const int wv = tid >> 6, ln = tid & 63;
  const int wm = wv >> 1, wn = wv & 1;
  const int li = ln & 31, lk = ln >> 5;
  unsigned char* const e_pl = sm.main;                                        // [2][128][APB]
  int32_t* const ee = reinterpret_cast<int32_t*>(sm.main + AE_BYTES);         // [128]
  unsigned char* const gb = sm.main + AE_BYTES + 128 * 4 + grp * AG_BYTES;
  unsigned char* const r1 = gb;                  // q planes, later the scores
  unsigned char* const r2 = gb + AIMG64;         // k planes, later the probabilities' planes
  unsigned char* const r3 = gb + 2 * AIMG64;     // v^T planes [d][key]
  int32_t* const eq = reinterpret_cast<int32_t*>(gb + 3 * AIMG64);
  int32_t* const ek = eq + 64;
  int32_t* const ev = eq + 128;
  int32_t* const ep = eq + 192;
  float* const sc = reinterpret_cast<float*>(r1);
  const int D3 = 3 * D;
  // (descriptors over exactly T rows: the frames beyond read as zeros, their context rows are not stored)
  auto rsrc_q = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(qkv), 0, (uint32_t)(T * D3 * 4), 0x00020000);
  auto rsrc_c = __builtin_amdgcn_make_buffer_rsrc(ctx, 0, (uint32_t)(T * D * 4), 0x00020000);
  auto rsrc_e = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(rel), 0, (uint32_t)(rel_len * DH * 4), 0x00020000);
  // C[32 x 32] = A[rows a0 ..][64 k] B[rows b0 ..][64 k]^T on the planes: (main, cross) accumulators
  auto tile = [&](const unsigned char* A, int a_plane, const unsigned char* B, int b_plane, f32x16& acc, f32x16& accx) {
    const unsigned char* pa = A + li * APB + lk * 16;
    const unsigned char* pb = B + li * APB + lk * 16;
    u32x4 ah[4], al[4], bh[4], bl[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {   // k = 16 s4 ..
      ah[s4] = *reinterpret_cast<const u32x4*>(pa + s4 * 32);
      al[s4] = *reinterpret_cast<const u32x4*>(pa + a_plane + s4 * 32);
      bh[s4] = *reinterpret_cast<const u32x4*>(pb + s4 * 32);
      bl[s4] = *reinterpret_cast<const u32x4*>(pb + b_plane + s4 * 32);
    }
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      accx = mfma_f16(ah[s4], bl[s4], accx);
      acc = mfma_f16(ah[s4], bh[s4], acc);
      accx = mfma_f16(al[s4], bh[s4], accx);
    }
  };
  // ---- the table window, once: window row w <-> offset j - i = w - 63 <-> table row w - 63 + rel_zero
  {
    const int w = tid_all >> 2, part = tid_all & 3;   // 128 rows x 4 lanes
    const int64_t r = (int64_t)w - 63 + rel_zero;
    f32x4 v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
      v[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                           rsrc_e, (r >= 0 && r < rel_len) ? (uint32_t)((r * DH + part * 16 + c * 4) * 4) : 0x80000000u, 0, 0));
    const int32_t ex = att_split16(v, 1.0f, e_pl + w * APB + part * 32, 128 * APB);
    if (part == 0) ee[w] = ex;
  }
  for (int hp = 0; hp < H; hp += 2) {
    const int h = hp + grp;
    const bool live = h < H;   // (an odd head count: the second group idles through the barriers)
    __syncthreads();           // the group's regions are free (previous pair); E is staged (first pair)
    if (live) {
      const int row = tid >> 2, part = tid & 3;   // 64 rows x 4 lanes: q, k, v row `row`, 16 of its 64 values
      const uint32_t off = (uint32_t)((row * D3 + h * DH + part * 16) * 4);
      f32x4 q[4], k[4], v[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        q[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_q, off, c * 16, 0));
        k[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_q, off, D * 4 + c * 16, 0));
        v[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_q, off, 2 * D * 4 + c * 16, 0));
      }
      const int32_t xq = att_split16(q, scale, r1 + row * APB + part * 32, APLANE64);
      const int32_t xk = att_split16(k, 1.0f, r2 + row * APB + part * 32, APLANE64);
      // v row `row` (a key): its power of two, then the TRANSPOSED planes [d][key]
      float mx = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) mx = fmaxf(mx, fabsf(v[c][e]));
      mx = fmaxf(mx, dpp_move<0xB1>(mx));
      mx = fmaxf(mx, dpp_move<0x4E>(mx));
      const int32_t xv = scale_exponent(mx);
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x = v[c][e];
          const _Float16 hh = (_Float16)ldexpf(x, xv);
          const _Float16 ll = (_Float16)fmaf((float)hh, -kLowUp, ldexpf(x, xv + 11));
          unsigned char* dst = r3 + (part * 16 + c * 4 + e) * APB + row * 2;
          *reinterpret_cast<_Float16*>(dst) = hh;
          *reinterpret_cast<_Float16*>(dst + APLANE64) = ll;
        }
      if (part == 0) eq[row] = xq, ek[row] = xk, ev[row] = xv;
    }
    __syncthreads();
    f32x16 sacc, saccx, pacc[2], paccx[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) sacc[e] = saccx[e] = pacc[0][e] = paccx[0][e] = pacc[1][e] = paccx[1][e] = 0.f;
    int32_t exq[16];
    if (live) {
      // wave (wm, wn): query rows 32 wm .., keys 32 wn ..; window rows 64 wn + 32 t ..
      tile(r1 + wm * 32 * APB, APLANE64, r2 + wn * 32 * APB, APLANE64, sacc, saccx);
#pragma unroll
      for (int t = 0; t < 2; ++t) tile(r1 + wm * 32 * APB, APLANE64, e_pl + (wn * 64 + t * 32) * APB, 128 * APB, pacc[t], paccx[t]);
#pragma unroll
      for (int e = 0; e < 16; ++e) exq[e] = eq[wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk];
    }
    __syncthreads();   // the q and k planes are dead: the scores go to the q planes' region
    if (live) {
      const int j = wn * 32 + li;
      const int32_t xk = ek[j];
      // accumulator layout: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int i = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
        sc[i * VP + j] = (j < len) ? ldexpf(fmaf(saccx[e], kLowDown, sacc[e]), -(exq[e] + xk)) : -INFINITY;
      }
    }
    __syncthreads();
    if (live) {
      // the shifted term: element (i, w) of q E^T belongs to key j = w + i - 63 (each (i, j) has exactly one owner)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int w = wn * 64 + t * 32 + li;
        const int32_t xe = ee[w];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int i = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
          const int j = w + i - 63;
          if (j >= 0 && j < 64) sc[i * VP + j] += ldexpf(fmaf(paccx[t][e], kLowDown, pacc[t][e]), -(exq[e] + xe));
        }
      }
    }
    __syncthreads();
    if (live) {
      // row softmax, four lanes per row (16 keys each); the probabilities leave as planes with v's powers of two
      // folded in and a power of two per query row
      const int i = tid >> 2, part = tid & 3;
      const float* pr = sc + i * VP + part * 16;
      f32x4 v[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = *reinterpret_cast<const f32x4*>(pr + c * 4);
      float m = -INFINITY;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) m = fmaxf(m, v[c][e]);
      m = fmaxf(m, dpp_move<0xB1>(m));
      m = fmaxf(m, dpp_move<0x4E>(m));
      // a fully padded sequence (len = 0) is softmax over -inf only -> NaN in torch; zeros here
      float sum = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[c][e] = (m > -INFINITY) ? __expf(v[c][e] - m) : 0.f;
          sum += v[c][e];
        }
      sum += dpp_move<0xB1>(sum);
      sum += dpp_move<0x4E>(sum);
      const float inv = sum > 0.f ? 1.0f / sum : 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const i32x4 xv = *reinterpret_cast<const i32x4*>(&ev[part * 16 + c * 4]);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[c][e] = ldexpf(v[c][e] * inv, -xv[e]);
      }
      const int32_t xp = att_split16(v, 1.0f, r2 + i * APB + part * 32, APLANE64);
      if (part == 0) ep[i] = xp;
    }
    __syncthreads();
    if (live) {
      f32x16 oacc, oaccx;
#pragma unroll
      for (int e = 0; e < 16; ++e) oacc[e] = oaccx[e] = 0.f;
      tile(r2 + wm * 32 * APB, APLANE64, r3 + wn * 32 * APB, APLANE64, oacc, oaccx);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int i = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
        const int d = wn * 32 + li;
        const float o = ldexpf(fmaf(oaccx[e], kLowDown, oacc[e]), -ep[i]);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o), rsrc_c, (uint32_t)((i * D + h * DH + d) * 4), 0, 0);
      }
    }
  }
  __syncthreads();
}

// ---- GLU . depthwise conv (15 taps, zero outside [0, T)) . BatchNorm affine . activation: a thread owns a channel
// (impl.py:478-489; the arithmetic order of nn.hip's glu_dwconv_kernel).  x [T, 2 D] -> out [T, D]
__device__ __forceinline__ void glu_dwconv_phase(const float* __restrict__ x, float* __restrict__ out, const ConvParams& L,
                                                 int T, int D) {
  constexpr int K = 15, PAD = 7;
  auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (uint32_t)(T * 2 * D * 4), 0x00020000);
  auto rsrc_o = __builtin_amdgcn_make_buffer_rsrc(out, 0, (uint32_t)(T * D * 4), 0x00020000);
  for (int d = lane_id_here(); d < D; d += NT) {
    float gl[RT];
#pragma unroll
    for (int tb = 0; tb < RT; tb += 16) {   // 32 loads in flight, then their gates (frames beyond T read as zeros)
      float av[16], bv_[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const uint32_t off = (uint32_t)(((tb + i) * 2 * D + d) * 4);
        av[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc_x, off, 0, 0));
        bv_[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc_x, off, D * 4, 0));
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) gl[tb + i] = av[i] * __builtin_amdgcn_rcpf(1.0f + __expf(-bv_[i]));
      __builtin_amdgcn_sched_barrier(0);
    }
    float wr[K];
#pragma unroll
    for (int k = 0; k < K; ++k) wr[k] = L.dw_w[(int64_t)d * K + k];
    const float bv = L.dw_b ? L.dw_b[d] : 0.f, sc = L.bn_scale ? L.bn_scale[d] : 1.f, sh = L.bn_shift ? L.bn_shift[d] : 0.f;
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      float acc = bv;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int s = t + k - PAD;
        if (s >= 0 && s < RT) acc += wr[k] * gl[s];
      }
      acc = acc * sc + sh;
      if (L.conv_act == 1) acc = acc * __builtin_amdgcn_rcpf(1.0f + __expf(-acc));
      else if (L.conv_act == 2) acc = fmaxf(acc, 0.f);
      else if (L.conv_act == 3) acc = 0.5f * acc * (1.0f + erff(acc * 0.70710678118654752f));
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc), rsrc_o, (uint32_t)((t * D + d) * 4), 0, 0);
    }
  }
}

__global__ __launch_bounds__(NT, 2) void conformer_stack_kernel(Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
  Smem sm;
  sm.main = s_dyn;
  sm.exps = reinterpret_cast<int32_t*>(s_dyn + MAIN);
  sm.stat = reinterpret_cast<float2*>(s_dyn + MAIN + RT * 4);
  sm.flags = reinterpret_cast<int32_t*>(s_dyn + MAIN + RT * 4 + RT * 8);
  const int64_t n = blockIdx.x;
  const int T = a.T, D = a.D, FF = a.FF;
  const int len = (int)(a.lens ? min((int64_t)T, max((int64_t)0, a.lens[n])) : T);
  float* const X = a.x + n * (int64_t)T * D;          // the residual stream, row pitch D
  float* const Hb = a.scratch + n * a.scratch_floats;  // [64][FF] hidden of the FFNs / the GLU input (2 D <= FF)
  float* const Q = Hb + (int64_t)RT * FF;              // [64][3 D]
  float* const C = Q + (int64_t)RT * 3 * D;            // [64][D] attention / convolution output

  // The twelve phases of a layer, ONE instance of each phase's code inside a loop (the dependency chain of the
  // layer: a barrier per phase instead of a launch):
  //   0 ff1_up X -> H | 1, 2 ff1_dn halves H -> X | 3 qkv X -> Q | 4 attention Q -> C | 5 out C -> X |
  //   6 pw1 X -> H | 7 GLU . dwconv . BN . act H -> C | 8 pw2 C -> X | 9 ff2_up | 10, 11 ff2_dn halves
  const int phases = 12 * a.num_layers;
  unsigned long long k0 = 0, r0 = 0, tprev = 0;
  if (a.trace && blockIdx.x == 0 && threadIdx.x == 0) {
    k0 = __builtin_amdgcn_s_memtime();       // shader-clock ticks
    r0 = __builtin_amdgcn_s_memrealtime();   // the constant 100 MHz counter: their ratio is the clock the CU ran at
  }
#pragma unroll 1
  for (int ph = 0; ph < phases; ++ph) {
    const int l = ph / 12, p = ph - 12 * l;
    const Layer& L = a.layers[l];
    __syncthreads();   // the previous phase's rows are written (and visible), its reads of the LDS regions are over
    unsigned long long t0 = 0;
    if (a.trace && blockIdx.x == 0 && threadIdx.x == 0) {
      t0 = __builtin_amdgcn_s_memtime();
      if (ph > 0) a.trace[16 + (p + 11) % 12] += t0 - tprev;   // the previous phase barrier to barrier (its slowest wave)
      tprev = t0;
    }
    if (p == 4) {
      attention_phase(sm, Q, C, a.rel, a.rel_zero, a.rel_len, T, len, a.heads, D, a.att_scale);
    } else if (p == 7) {
      ConvParams cp = uniform(*reinterpret_cast<const ConvParams*>(&L.dw_w));
      glu_dwconv_phase(Hb, C, cp, T, D);
    } else {
      // gemm index in the layer's table, where its rows come from and go to
      const int gi = p < 4 ? p : (p < 7 ? p - 1 : p - 2);
      const Gemm g = uniform(reinterpret_cast<const Gemm*>(&L)[gi]);
      const int kind = gi >= 7 ? gi - 7 : gi;          // 0 up | 1, 2 down halves | 3 qkv | 4 out | 5 pw1 | 6 pw2
      const float* src;
      float* dst;
      const float* res = nullptr;
      int ld_src, ld_dst;
      if (kind == 0 || kind == 5) {            // X -> H
        src = X, ld_src = D, dst = Hb, ld_dst = FF;
      } else if (kind == 1 || kind == 2) {     // H halves -> X (+ X)
        src = Hb + (kind == 2 ? KP : 0), ld_src = FF, dst = X, ld_dst = D, res = X;
      } else if (kind == 3) {                  // X -> Q
        src = X, ld_src = D, dst = Q, ld_dst = 3 * D;
      } else {                                 // C -> X (+ X)
        src = C, ld_src = D, dst = X, ld_dst = D, res = X;
      }
      const bool lane_wide = g.colsum ? stage_rows<true>(sm, src, ld_src, T, g.ln_eps)
                                      : stage_rows<false>(sm, src, ld_src, T, g.ln_eps);
      // (the barrier that publishes the image also tells every wave whether ANY staged element misses its row's scale)
      const bool wide_a = __syncthreads_or(lane_wide) != 0;
      if (a.trace && blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        a.trace[12] += t1 - t0;   // (staging share of the projection phases)
      }
      gemm_phase(sm, g, src, ld_src, res, D, dst, ld_dst, T, a.wide_count, wide_a);
    }
    if (a.trace && blockIdx.x == 0 && threadIdx.x == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      a.trace[p] += __builtin_amdgcn_s_memtime() - t0;   // (thread 0's view: its own wave's share of the phase)
    }
  }
  if (a.trace && blockIdx.x == 0 && threadIdx.x == 0) {
    a.trace[13] += __builtin_amdgcn_s_memtime() - k0;
    a.trace[14] += __builtin_amdgcn_s_memrealtime() - r0;
  }
}

}  // namespace mega
}  // namespace aps

using namespace aps;

static_assert(sizeof(mega::Gemm) == sizeof(ApsMegaGemm), "ApsMegaGemm mirrors mega::Gemm");
static_assert(sizeof(mega::Layer) == sizeof(ApsMegaLayer), "ApsMegaLayer mirrors mega::Layer");

static unsigned long long* g_mega_trace = nullptr;
// (experiments; not in include/aps_amd.h) the 32 counters of APS_MEGA_TRACE=1 (0-11 wave 0's share of each phase kind, 12 staging, 13 / 14 the kernel in shader-clock / 100 MHz ticks, 16-27 each phase kind barrier to barrier), cleared by the call
extern "C" int aps_debug_conformer_trace(unsigned long long* host32) {
  if (!g_mega_trace) return APS_ERR_INVALID;
  if (hipDeviceSynchronize() != hipSuccess ||
      hipMemcpy(host32, g_mega_trace, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemset(g_mega_trace, 0, 32 * sizeof(unsigned long long)) != hipSuccess)
    return APS_ERR_LAUNCH;
  return APS_OK;
}

extern "C" int64_t aps_conformer_stack_scratch(int64_t D, int64_t FF) {
  // floats per utterance: hidden [64][FF] | qkv [64][3 D] | attention / convolution output [64][D]
  return (int64_t)mega::RT * (FF + 4 * D);
}

extern "C" int aps_conformer_stack(float* x, const int64_t* lens, const ApsMegaLayer* layers, int32_t num_layers,
                                   const float* rel, int64_t rel_zero, int64_t rel_len, int64_t N, int64_t T, int64_t D,
                                   int64_t FF, int64_t heads, float* scratch, int32_t* wide_count, void* stream) {
  APS_CHECK_ARG(x && layers && scratch && rel && rel_len > 0 && num_layers > 0 && N > 0 && N <= 65535 && T > 0);
  // the shapes this kernel is built for (everything else stays on the per-launch path)
  if (T > mega::RT || D != mega::KP || FF != 2 * mega::KP || heads <= 0 || D != heads * 64) return APS_ERR_UNSUPPORTED;
  static ApsPerDevice attr_set;
  if (!aps_lds_opt_in(attr_set, reinterpret_cast<const void*>(&mega::conformer_stack_kernel), mega::LDS_BYTES))
    return APS_ERR_LAUNCH;
  // experiments: APS_MEGA_TRACE=1 -> cycles per phase kind of workgroup 0, summed over layers and launches
  // (aps_debug_conformer_trace reads and clears them)
  static const bool want_trace = [] { const char* e = getenv("APS_MEGA_TRACE"); return e && e[0] == '1'; }();
  if (want_trace && !g_mega_trace) {
    if (hipMalloc(&g_mega_trace, 32 * sizeof(unsigned long long)) != hipSuccess ||
        hipMemset(g_mega_trace, 0, 32 * sizeof(unsigned long long)) != hipSuccess)
      return APS_ERR_LAUNCH;
  }
  mega::Args a{x, lens, reinterpret_cast<const mega::Layer*>(layers), scratch, wide_count, rel, rel_zero, rel_len,
               aps_conformer_stack_scratch(D, FF), num_layers, (int32_t)T, (int32_t)D, (int32_t)FF, (int32_t)heads, 0,
               1.0f / sqrtf(64.0f), g_mega_trace};
  hipLaunchKernelGGL(mega::conformer_stack_kernel, dim3((unsigned)N), dim3(mega::NT), mega::LDS_BYTES,
                     static_cast<hipStream_t>(stream), a);
  return aps_launch_status();
}
