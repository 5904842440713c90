// Persistent LSTM layer for gfx950: the recurrence of the RNN mask estimator
// (PyTorchRNNEncoder -> nn.LSTM, aps/asr/base/encoder.py:87-184, aps/asr/base/component.py:26-55,
// 145-190) as ONE launch per layer and direction instead of a GEMM + pointwise launch per step.
//
//   gates_t = pre_t + h_{t-1} W_hh^T + b_hh,   pre = x W_ih^T + b_ih  (one batched GEMM, aps_linear)
//   i, f, g, o = sigmoid, sigmoid, tanh, sigmoid  (torch gate order)
//   c_t = f c_{t-1} + i g,  h_t = o tanh(c_t)
//
// Decomposition: a workgroup owns 4 UT hidden units (UT "unit tiles": 16 UT rows of W_hh, 4 gates x
// 4 UT units) of 16 MT utterances (MT "row tiles") for the whole sequence.  Its slice of W_hh lives
// in VGPRs (4 waves split K = H, each lane UT H/16 values: the B operands of
// v_mfma_f32_16x16x4_f32, exact fp32), the cell state c in the registers of its gate threads.  Per
// step a workgroup needs h_{t-1} of ITS utterances ([16 MT, H], gathered from the layer output y
// itself) and produces 4 UT columns of h_t for them.  The gather crosses XCDs, i.e. it is served by
// the fabric, and its volume per step is (H / 4 UT) N H 4 bytes whatever the batch split: wide unit
// blocks (UT) cut it, splitting the batch over workgroups (B = ceil(N / 16 MT)) restores the
// parallelism.  (Measured: UT = 1 moves 24-32 MB per step on the benchmark shapes, ~5 TB/s: the
// step time WAS that transfer.)
//
// Inter-workgroup hand-off per step ((H / 4 UT) B workgroups per direction, all resident):
// the layer output y doubles as the exchange buffer and every cell of it is written exactly once
// per call, so "written" is one bit per word: the call pre-fills y with the sentinel 0xFFFFFFFF (a
// NaN pattern no finite h has) by a memset node ahead of the launch,
//   producer: h_t chunk -> y with one 16-byte write-through (sc1) store per utterance,
//   consumer: gathers h_{t-1} with 16-byte sc1 loads (L1-bypassing, served from the coherence
//             point) and re-loads the chunks that still hold a sentinel word,
// i.e. no flag, no drain, no separate poll round trip: the data is the flag (word-granular, so a
// torn 16-byte store is harmless).  Spins are bounded (a timeout word in the workspace turns a
// lost workgroup into a reported error, not a hang).  Both directions of a bidirectional layer run
// in the same launch (blockIdx / G); the second group may instead be an independent forward LSTM on
// the same input (the real / imaginary pair of DCCRN's complex LSTM).
//
// Sequence lengths follow the packed-sequence semantics of the reference: outputs at t >= len are
// zero; the reverse direction of a bidirectional layer starts at each utterance's own last frame.
//
// The recurrent product runs on the f16 matrix pipe as THREE products of two-plane operands
// (v_mfma_f32_16x16x32_f16, ~17 cycles per 16 x 16 x 32 against 8 x 32 cycles for the same block on
// v_mfma_f32_16x16x4_f32: the MFMA part of a step was 1.7 us of 5.9 at H = 512, 32 rows):
//   h = h_hi + h_lo,  h_hi = rtz_f16(h), h_lo = rtz_f16(h - h_hi)   (|h| < 1 by construction: no
//       scale; the gathering thread splits the 4 values it fetched on their way into LDS),
//   W_hh row r scaled by 2^e_r to [2^10, 2^11), w' = w_hi + w_lo (round to nearest), resident in the
//       same registers the fp32 slice took; 2^-e_r is applied to the finished row sum,
//   h W^T ~ h_hi w_hi + h_hi w_lo + h_lo w_hi, fp32 accumulation.
// What is dropped is h_lo w_lo and the planes' own rounding: per element |dh| <= max(2^-21 |h|,
// 2^-24), |dw| <= 2^-22 max_k|w_rk|, i.e. an ABSOLUTE error of the gate pre-activation below
// 2^-20 sum_k |w_rk| -- the pre-activation feeds sigmoid / tanh, whose slope is <= 1, and the fp32
// kernel's own v_exp / v_rcp forms are 1e-7 away from the IEEE functions already.  Hidden sizes that
// are not a multiple of 128 (a wave's K quarter must hold whole 32-wide blocks) keep the fp32 form.
#include <stdio.h>
#include <stdlib.h>

#include <atomic>

#include <type_traits>

#include "common.h"

namespace aps {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

#ifndef APS_LSTM_F16
#define APS_LSTM_F16 1  // 0: the exact-fp32 recurrent product everywhere (A/B builds)
#endif

constexpr int kLstmUnits = 4;            // hidden units per workgroup
constexpr int kLstmRows = 4 * kLstmUnits;  // W_hh rows per workgroup (the MFMA N dimension)
constexpr unsigned kSpinLimit = 1u << 20;
constexpr unsigned kSentinel = 0xffffffffu;

struct LstmArgs {
  const float* pre[2];   // per direction [N, T, 4H]
  const float* w_hh[2];  // [4H, H]
  const float* b_hh[2];  // [4H] or null
  const int64_t* lens;   // [N] or null
  float* y;              // [N, T, ldy]; direction d owns columns d H .. d H + H - 1
  unsigned* tmo;         // timeout word
  int32_t N, T, H, ldy;
  int32_t bsplit;          // B: batch splits (workgroups along the utterance axis)
  int32_t second_reverse;  // 1: group 1 is the backward direction; 0: a second forward LSTM
  int32_t debug;  // timing probes only (APS_LSTM_DEBUG): 1 = no gather, 2 = gather without waiting, 3 = no gather and no store
  int32_t groups;    // directions / paired LSTMs in this launch
  unsigned* team;    // team form: [8][32] placement table (zeroed ahead of the launch)
};

// v_rcp_f32 / v_exp_f32 forms (1 ulp): the kernel is instruction-latency bound (one wave per SIMD),
// so the IEEE division expansions (10 instructions each, 5 per cell) are worth removing
__device__ __forceinline__ float sigmoid_f(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}
__device__ __forceinline__ float tanh_f(float x) {
  // 1 - 2 / (e^{2x} + 1): saturates cleanly (e^{2x} = inf -> 1, 0 -> -1), abs error ~1e-7
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f);
}
__device__ __forceinline__ bool has_sentinel(u32x4 v) {
  return max(max(v.x, v.y), max(v.z, v.w)) == kSentinel;
}

#ifdef APS_LSTM_TRACE
// experiments only (scripts/lstm_trace.py, a library built with -DAPS_LSTM_TRACE): s_memtime stamps of
// lane 0 of every wave of two workgroups, summed over the steps: [workgroup slot 2][group 2][wave 4][16]
__device__ unsigned long long g_lstm_trace[2 * 2 * 4 * 16];
#define APS_LSTM_STAMP(k) stamp[k] = __builtin_amdgcn_s_memtime();
#define APS_LSTM_PIN(x) asm volatile("" ::"v"(x));
#else
#define APS_LSTM_STAMP(k)
#define APS_LSTM_PIN(x)
#endif

// ---- two-plane f16 operands of the recurrent product (see the header) ---------------------------
// 4 gathered fp32 values -> 4 + 4 f16 (hi | lo), truncating conversions (v_cvt_pkrtz_f16_f32: the
// residual is exact in fp32 and keeps the sign of h)
__device__ __forceinline__ void lstm_split4(u32x4 v, u32x2& hi, u32x2& lo) {
  const float x0 = __uint_as_float(v.x), x1 = __uint_as_float(v.y);
  const float x2 = __uint_as_float(v.z), x3 = __uint_as_float(v.w);
  const auto h01 = __builtin_amdgcn_cvt_pkrtz(x0, x1), h23 = __builtin_amdgcn_cvt_pkrtz(x2, x3);
  // x - hi in ONE instruction: v_fma_mix_f32 converts the f16 operand on the way in (the compiler
  // emits v_cvt_f32_f16 + v_sub_f32 for the C form, and this pass is VALU bound)
  const unsigned p01 = __builtin_bit_cast(unsigned, h01), p23 = __builtin_bit_cast(unsigned, h23);
  float r0, r1, r2, r3;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(p01), "v"(x0));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(p01), "v"(x1));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r2) : "v"(p23), "v"(x2));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r3) : "v"(p23), "v"(x3));
  const auto l01 = __builtin_amdgcn_cvt_pkrtz(r0, r1), l23 = __builtin_amdgcn_cvt_pkrtz(r2, r3);
  hi = u32x2{__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23)};
  lo = u32x2{__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23)};
}
// exponent e with max 2^e in [2^10, 2^11) (0 for an all-zero or non-finite row)
__device__ __forceinline__ int lstm_row_exponent(float row_max) {
  if (!(row_max > 0.f) || !(row_max < 3.0e38f)) return 0;
  return 11 - __builtin_amdgcn_frexp_expf(row_max);
}
// 8 consecutive weights of a row, scaled, as the two planes of one MFMA B operand
__device__ __forceinline__ void lstm_weight_planes(const float* wp, int e, f16x8& hi, f16x8& lo) {
  const float4 t0 = *reinterpret_cast<const float4*>(wp), t1 = *reinterpret_cast<const float4*>(wp + 4);
  const float w[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float x = ldexpf(w[i], e);
    const _Float16 h = (_Float16)x;
    hi[i] = h;
    lo[i] = (_Float16)(x - (float)h);
  }
}
__device__ __forceinline__ float lstm_absmax8(const float* wp) {
  const float4 t0 = *reinterpret_cast<const float4*>(wp), t1 = *reinterpret_cast<const float4*>(wp + 4);
  return fmaxf(fmaxf(fmaxf(fabsf(t0.x), fabsf(t0.y)), fmaxf(fabsf(t0.z), fabsf(t0.w))),
               fmaxf(fmaxf(fabsf(t1.x), fabsf(t1.y)), fmaxf(fabsf(t1.z), fabsf(t1.w))));
}
__device__ __forceinline__ f32x4 lstm_mfma16(f16x8 a, f16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// OCC: waves per SIMD the register allocation must leave room for (2: a second launch of another
// stream shares every CU -- the `share` = 2 case of the team form, whose grid is the whole chip)
template <int KREGS, int MT, int UT, bool TEAM = false, int OCC = 1>
__global__ __launch_bounds__(256, OCC) void lstm_layer_kernel(LstmArgs a) {
  constexpr int H = 16 * KREGS;
  constexpr bool F16 = APS_LSTM_F16 != 0 && KREGS % 8 == 0;  // two-plane f16 product (header)
  constexpr int PITCH = H + 4;  // 16-byte aligned rows; 4 r mod 64 banks: b128 fetches conflict free
  constexpr int PH = H + 8;     // f16 row pitch of a plane: the same 16-byte shift per row
  constexpr int ROWS = 16 * MT;
  constexpr int UNITS = kLstmUnits * UT;  // hidden units per workgroup
  constexpr int GR = kLstmRows * UT;      // gate rows per workgroup
  constexpr int G = H / UNITS;
  static_assert(!TEAM || G == 32, "team form: one unit block per CU of an XCD");
  // The batch is processed as NH interleaved groups of 16 MTH utterances (utterances are
  // independent): while group g's MFMAs / gates run, the gather of the NEXT group's h_{t-1} --
  // published half a step ago -- is already in flight.
  // (two groups in flight hold 2 x NL x 4 gather registers: only while that fits the budget)
  constexpr int NH = (MT % 2 == 0 && (MT / 2) * H / 64 <= 16) ? 2 : 1;
  constexpr int MTH = MT / NH;       // M tiles per group
  constexpr int RH = 16 * MTH;       // utterance rows per group
  constexpr int CH = H / 4;          // float4 chunks per row
  constexpr int NL = RH * CH / 256;  // gather loads per thread and group (H % 64 == 0: exact)
  static_assert(RH * UNITS <= 256, "one gate thread per (utterance of a group, unit)");
  extern __shared__ __attribute__((aligned(16))) float s_dyn[];
  float* s_h = s_dyn;                // fp32: [ROWS][PITCH]; f16: per group [hi | lo][RH][PH] halves
  constexpr int RP = 4 * GR + 4;     // floats per utterance row of the reduction buffer
  float* s_red = s_dyn + ROWS * PH;  // [RH][gate row GR][wave 4] (+ 4 per row), shared by the groups
  const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
  const int per_grp = G * a.bsplit;
  const int dbg = a.debug & 3;
  int grp, b, row0;
  bool plain_publish = false;  // team form on one XCD: h stays in that XCD's L2
  if constexpr (TEAM) {
    // Team form (see launch_lstm_team): the workgroups that exchange h -- all G = 32 unit blocks of
    // one (group, batch split) -- are the blocks with the same id % 8, which the dispatcher is
    // OBSERVED to place on one XCD.  Where that holds the hand-off needs no trip through the fabric:
    // plain stores leave the line in the XCD's L2, the sc1 gathers (L1 bypassing) are served from
    // it.  Placement is not a contract, so every workgroup publishes the XCC id it runs on and the
    // team takes the short path only if all 32 agree; otherwise it runs the placement independent
    // protocol (sc1 stores).  All members read the same 32 words: the decision is the same in all.
    const int team = (int)blockIdx.x & 7;
    if (team >= a.groups * a.bsplit) return;
    b = (int)blockIdx.x >> 3;
    grp = team / a.bsplit;
    row0 = (team % a.bsplit) * ROWS;
    const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) + 1u;  // XCC_ID[3:0] + 1
    unsigned* tab = a.team + team * 32;
    if (tid == 0) __hip_atomic_store(tab + b, xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int* s_flag = reinterpret_cast<int*>(s_dyn);
    if (tid < 64) {
      unsigned seen = xcc;
      if (tid < 32) {
        unsigned spins = 0;
        do {
          seen = __hip_atomic_load(tab + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (seen == 0) __builtin_amdgcn_s_sleep(1);
        } while (seen == 0 && ++spins < kSpinLimit);  // (a missing member: the main loop reports it)
      }
      const bool all_here = __ballot(seen != xcc) == 0;
      if (tid == 0) {
        s_flag[0] = all_here ? 1 : 0;
        if (a.debug & 32)  // experiment: word 1 counts the decisions (low half: short path), word 2 = an XCC id
          __hip_atomic_fetch_add(a.tmo + 1, all_here ? 1u : 0x10000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((a.debug & 32) && blockIdx.x == 9) a.tmo[2] = xcc | (seen << 8);
      }
    }
    __syncthreads();
    plain_publish = (s_flag[0] != 0 || (a.debug & 8)) && !(a.debug & 16);
    __syncthreads();
  } else {
    // experiment (APS_LSTM_DEBUG bit 2): the grid is 8 x as large and only the blocks with id % 8 == 0
    // (observed: XCD 0) take part -- every workgroup of the launch on one XCD; bit 3: plain stores
    // (the line stays in that XCD's L2) instead of sc1 ones
    if ((a.debug & 4) && (blockIdx.x & 7) != 0) return;
    const int bid = (a.debug & 4) ? (int)blockIdx.x >> 3 : (int)blockIdx.x;
    grp = bid / per_grp;                // group: direction or paired LSTM
    b = (bid % per_grp) % G;            // unit block
    row0 = ((bid % per_grp) / G) * ROWS;  // first utterance of this batch split
    plain_publish = (a.debug & 8) != 0;
  }
  const int dir = grp & a.second_reverse;              // 1: this group runs backward in time
  const int u0 = b * UNITS;
  const int N = a.N, T = a.T;
  const float* pre = a.pre[grp];
  const float* w_hh = a.w_hh[grp];
  const float* b_hh = a.b_hh[grp];
  const int col0 = grp * H;  // this group's first column of y

  // ---- resident W_hh slice.  K order inside a wave's quarter is permuted so that one b128 LDS
  // fetch feeds 4 MFMAs: MFMA (j, e) contracts k = wv H/4 + 16 j + 4 (ln >> 4) + e on both operands.
  float wreg[F16 ? 1 : UT][F16 ? 1 : KREGS];
  f16x8 whi[F16 ? UT : 1][F16 ? KREGS / 8 : 1], wlo[F16 ? UT : 1][F16 ? KREGS / 8 : 1];
  float wscale[4] = {1.f, 1.f, 1.f, 1.f};  // f16: 2^-e of this gate thread's 4 gate rows
  if constexpr (F16) {
    // MFMA m4 contracts k = wv H/4 + 32 m4 + 8 (ln >> 4) + e, e = 0..7, on both operands.  Row
    // exponents first: the row's largest magnitude over ALL of K (lanes ln >> 4, then the 4 waves)
    int* s_exp = reinterpret_cast<int*>(s_red);  // [4][UT][16] partial maxima, then [UT][16] exponents
#pragma unroll
    for (int ut = 0; ut < UT; ++ut) {
      const int j = ln & 15;
      const int row = (j >> 2) * H + u0 + 4 * ut + (j & 3);
      const float* wp = w_hh + (int64_t)row * H + wv * (H / 4) + 8 * (ln >> 4);
      float mx = 0.f;
#pragma unroll
      for (int m4 = 0; m4 < KREGS / 8; ++m4) mx = fmaxf(mx, lstm_absmax8(wp + 32 * m4));
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      if (ln < 16) s_red[(wv * UT + ut) * 16 + ln] = mx;
    }
    __syncthreads();
    float mrow = 0.f;
    if (tid < UT * 16)
      mrow = fmaxf(fmaxf(s_red[tid], s_red[UT * 16 + tid]),
                   fmaxf(s_red[2 * UT * 16 + tid], s_red[3 * UT * 16 + tid]));
    __syncthreads();
    if (tid < UT * 16) s_exp[tid] = lstm_row_exponent(mrow);
    __syncthreads();
#pragma unroll
    for (int ut = 0; ut < UT; ++ut) {
      const int j = ln & 15;
      const int row = (j >> 2) * H + u0 + 4 * ut + (j & 3);
      const float* wp = w_hh + (int64_t)row * H + wv * (H / 4) + 8 * (ln >> 4);
      const int e = s_exp[ut * 16 + j];
#pragma unroll
      for (int m4 = 0; m4 < KREGS / 8; ++m4)
        lstm_weight_planes(wp + 32 * m4, e, whi[ut][m4], wlo[ut][m4]);
    }
    {
      const int gu_ = tid % UNITS;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        wscale[q] = ldexpf(1.0f, -s_exp[16 * (gu_ >> 2) + q * 4 + (gu_ & 3)]);
    }
    __syncthreads();  // s_red is the reduction buffer from here on
  } else {
#pragma unroll
    for (int ut = 0; ut < UT; ++ut) {
      const int j = ln & 15;
      const int row = (j >> 2) * H + u0 + 4 * ut + (j & 3);
      const float* wp = w_hh + (int64_t)row * H + wv * (H / 4) + 4 * (ln >> 4);
#pragma unroll
      for (int q = 0; q < KREGS / 4; ++q) {
        const float4 t = *reinterpret_cast<const float4*>(wp + 16 * q);
        wreg[ut][4 * q + 0] = t.x, wreg[ut][4 * q + 1] = t.y;
        wreg[ut][4 * q + 2] = t.z, wreg[ut][4 * q + 3] = t.w;
      }
    }
  }
  // ---- gate role: thread (utterance row0 + g RH + gl of EACH group g, unit u0 + gu); the cell
  // states of its NH utterances live in its registers
  const int gl = tid / UNITS, gu = tid % UNITS;
  const bool gate_lane = gl < RH;
  bool gvalid[NH];
  int gn_c[NH], glen[NH];
  float c[NH];
#pragma unroll
  for (int g = 0; g < NH; ++g) {
    const int gn = row0 + g * RH + gl;
    gvalid[g] = gate_lane && gn < N;
    gn_c[g] = min(gn, N - 1);
    glen[g] = gvalid[g] ? (a.lens ? (int)min((int64_t)T, max((int64_t)0, a.lens[gn_c[g]])) : T) : 0;
    c[g] = 0.f;
  }
  float bias[4] = {0.f, 0.f, 0.f, 0.f};
  if (gate_lane && b_hh) {
#pragma unroll
    for (int g = 0; g < 4; ++g) bias[g] = b_hh[g * H + u0 + gu];
  }

  // buffer descriptor over y for the sc1 (write-through / L1-bypassing) 16-byte accesses; offsets
  // outside [0, y_bytes) read as zero instead of faulting (used for dead rows, see below)
  const uint32_t y_bytes = (uint32_t)((int64_t)N * T * a.ldy * 4);
  auto rsrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, y_bytes, 0x00020000);

  // Hot path = straight-line code, as few instructions as possible (one wave per SIMD: every
  // instruction's latency is exposed).  Gather address of (row r, chunk q) at step s:
  //   forward  y[r, s - 1]      -> voff[i] + (s - 1) ldy 4   (the step part is a scalar offset)
  //   backward y[r, len_r - s]  -> voff[i] - s ldy 4         (voff includes len_r ldy 4)
  // Rows past their length / past N are "dead": whatever they load only reaches their own (unused)
  // gate rows, so they are neither waited for nor zeroed; their offsets may leave the buffer.
  int voff[NH][NL];
  // row length (0 for rows >= N) of chunk (g, i): it is waited for while s < that.  Only the slow
  // path asks, so it is recomputed there instead of living in 2 x NL registers
  auto live_until = [&](int g, int i) -> unsigned {
    const int r = row0 + g * RH + (tid + 256 * i) / CH;
    if (r >= N) return 0u;
    return a.lens ? (unsigned)min((int64_t)T, max((int64_t)0, a.lens[r])) : (unsigned)T;
  };
#pragma unroll
  for (int g = 0; g < NH; ++g)
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int idx = tid + 256 * i;
      const int r = row0 + g * RH + idx / CH, q = idx % CH;
      const int rc = min(r, N - 1);
      const int rl = a.lens ? (int)min((int64_t)T, max((int64_t)0, a.lens[rc])) : T;
      voff[g][i] = (int)((((int64_t)rc * T + (dir ? rl : 0)) * a.ldy + col0 + 4 * q) * 4);
    }
  const int step_bytes = a.ldy * 4;
  u32x4 v[NH][NL];
  bool timed_out = false;

  auto issue = [&](auto gc, int s) {
    constexpr int g = decltype(gc)::value;
    if (dbg == 1 || dbg == 3) {  // timing probes (wave-uniform)
#pragma unroll
      for (int i = 0; i < NL; ++i) v[g][i] = u32x4{0u, 0u, 0u, 0u};
      return;
    }
    if (dir == 0) {
      const int soff = (s - 1) * step_bytes;
#pragma unroll
      for (int i = 0; i < NL; ++i)
        v[g][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[g][i], soff, 16);
    } else {
      const int back = s * step_bytes;
#pragma unroll
      for (int i = 0; i < NL; ++i)
        v[g][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[g][i] - back, 0, 16);
    }
  };
  // a sentinel word = not written yet: re-load those chunks until they are
  auto finish = [&](auto gc, int s) {
    constexpr int g = decltype(gc)::value;
    // fast path (2 instructions per chunk): no sentinel word anywhere in this lane's chunks
    unsigned top = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i)
      top = max(top, max(max(v[g][i].x, v[g][i].y), max(v[g][i].z, v[g][i].w)));
    if (top != kSentinel || dbg) return;
    unsigned live = 0, bad = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) live |= ((unsigned)s < live_until(g, i)) ? (1u << i) : 0u;
#pragma unroll
    for (int i = 0; i < NL; ++i) bad |= (has_sentinel(v[g][i]) && ((live >> i) & 1u)) ? (1u << i) : 0u;
    unsigned spins = 0;
    // slow path: a producer is behind.  The WHOLE gather is requested again, all chunks in flight
    // together (the data is write-once: a chunk that had arrived reads the same words), one round
    // trip per attempt -- re-requesting only the missing chunks one by one made an attempt as many
    // dependent round trips as there were chunks (3.6k cycles per step where every step takes this
    // path: the single-group forms)
    while (bad != 0 && !timed_out) {
      if (++spins > kSpinLimit) {
        timed_out = true;
        __hip_atomic_fetch_add(a.tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      issue(gc, s);
      bad = 0;
#pragma unroll
      for (int i = 0; i < NL; ++i)
        bad |= (has_sentinel(v[g][i]) && ((live >> i) & 1u)) ? (1u << i) : 0u;
    }
  };
  using G0 = std::integral_constant<int, 0>;
  using G1 = std::integral_constant<int, NH - 1>;
  int row_bytes0[NH];
#pragma unroll
  for (int g = 0; g < NH; ++g)
    row_bytes0[g] = (int)((((int64_t)gn_c[g] * T) * a.ldy + col0 + u0 + gu) * 4);

#ifdef APS_LSTM_TRACE
  const int tslot = blockIdx.x == 0 ? 0 : (blockIdx.x == 19 * 8 + 3 ? 1 : -1);
  unsigned long long stamp[10], tsum[NH][9];
#pragma unroll
  for (int g = 0; g < NH; ++g)
#pragma unroll
    for (int q = 0; q < 9; ++q) tsum[g][q] = 0;
#endif
  // one group's step: recurrent product (s > 0), gates, cell update, publish
  auto step = [&](auto gc, auto first, int s, const float (&p)[4]) {
    constexpr int g = decltype(gc)::value;
    constexpr bool FIRST = decltype(first)::value;  // s == 0: no recurrent term
    APS_LSTM_STAMP(0)
    float part[4] = {0.f, 0.f, 0.f, 0.f};
    const bool mine = gvalid[g];
    const int len = glen[g];
    if (!FIRST) {
      finish(gc, s);
      APS_LSTM_STAMP(1)
      float* sh = s_h + g * RH * (F16 ? PH : PITCH);
      _Float16* shh = reinterpret_cast<_Float16*>(sh);  // f16: [hi | lo][RH][PH]
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const int idx = tid + 256 * i;
        if constexpr (F16) {
          u32x2 hi, lo;
          lstm_split4(v[g][i], hi, lo);
          _Float16* dst = shh + (idx / CH) * PH + 4 * (idx % CH);
          *reinterpret_cast<u32x2*>(dst) = hi;
          *reinterpret_cast<u32x2*>(dst + RH * PH) = lo;
        } else {
          *reinterpret_cast<u32x4*>(sh + (idx / CH) * PITCH + 4 * (idx % CH)) = v[g][i];
        }
      }
      APS_LSTM_STAMP(2)
      __syncthreads();
      APS_LSTM_STAMP(3)
      // the next group's (or next step's first group's) gather flies during this group's compute
      // (a single group: nobody has published step s yet, the request goes out behind the publish)
      if (g + 1 < NH) {
        issue(G1{}, s);
      } else if (NH == 2 && s + 1 < T) {
        issue(G0{}, s + 1);
      }
      APS_LSTM_STAMP(4)
      // ---- partial products over this wave's K quarter (two accumulators per tile: consecutive
      // MFMAs never depend on each other)
      // (f16 form: MFMAs on the same accumulator are MTH UT apart; from 4 on -- 68 cycles against a
      // dependent latency of 40 -- one accumulator per tile is enough and frees 16 registers)
      constexpr bool DUAL = !F16 || MTH * UT < 4;
      f32x4 acc[MTH][UT], acc2[DUAL ? MTH : 1][DUAL ? UT : 1];
#pragma unroll
      for (int m = 0; m < MTH; ++m)
#pragma unroll
        for (int ut = 0; ut < UT; ++ut) {
          acc[m][ut] = f32x4{0.f, 0.f, 0.f, 0.f};
          if constexpr (DUAL) acc2[m][ut] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      if constexpr (F16) {
        const _Float16* hp = shh + (ln & 15) * PH + wv * (H / 4) + 8 * (ln >> 4);
#pragma unroll
        for (int m4 = 0; m4 < KREGS / 8; ++m4) {
#pragma unroll
          for (int m = 0; m < MTH; ++m) {
            const f16x8 ahi = *reinterpret_cast<const f16x8*>(hp + m * 16 * PH + 32 * m4);
            const f16x8 alo = *reinterpret_cast<const f16x8*>(hp + RH * PH + m * 16 * PH + 32 * m4);
#pragma unroll
            for (int ut = 0; ut < UT; ++ut) acc[m][ut] = lstm_mfma16(ahi, whi[ut][m4], acc[m][ut]);
#pragma unroll
            for (int ut = 0; ut < UT; ++ut) {
              if constexpr (DUAL) acc2[m][ut] = lstm_mfma16(ahi, wlo[ut][m4], acc2[m][ut]);
              else acc[m][ut] = lstm_mfma16(ahi, wlo[ut][m4], acc[m][ut]);
            }
#pragma unroll
            for (int ut = 0; ut < UT; ++ut) {
              if constexpr (DUAL) acc2[m][ut] = lstm_mfma16(alo, whi[ut][m4], acc2[m][ut]);
              else acc[m][ut] = lstm_mfma16(alo, whi[ut][m4], acc[m][ut]);
            }
          }
        }
      } else {
        const float* hp = sh + (ln & 15) * PITCH + wv * (H / 4) + 4 * (ln >> 4);
#pragma unroll
        for (int q = 0; q < KREGS / 4; ++q) {
#pragma unroll
          for (int m = 0; m < MTH; ++m) {
            const float4 t = *reinterpret_cast<const float4*>(hp + m * 16 * PITCH + 16 * q);
#pragma unroll
            for (int ut = 0; ut < UT; ++ut) {
              acc[m][ut] = __builtin_amdgcn_mfma_f32_16x16x4f32(t.x, wreg[ut][4 * q + 0], acc[m][ut], 0, 0, 0);
              acc2[m][ut] = __builtin_amdgcn_mfma_f32_16x16x4f32(t.y, wreg[ut][4 * q + 1], acc2[m][ut], 0, 0, 0);
              acc[m][ut] = __builtin_amdgcn_mfma_f32_16x16x4f32(t.z, wreg[ut][4 * q + 2], acc[m][ut], 0, 0, 0);
              acc2[m][ut] = __builtin_amdgcn_mfma_f32_16x16x4f32(t.w, wreg[ut][4 * q + 3], acc2[m][ut], 0, 0, 0);
            }
          }
        }
      }
      // D layout: lane l, register r -> (batch row 4 (l >> 4) + r, gate row l & 15 of unit tile ut)
#pragma unroll
      for (int m = 0; m < MTH; ++m)
#pragma unroll
        for (int ut = 0; ut < UT; ++ut) {
          if constexpr (DUAL) acc[m][ut] += acc2[m][ut];
#pragma unroll
          for (int r = 0; r < 4; ++r)
            s_red[(m * 16 + 4 * (ln >> 4) + r) * RP + (16 * ut + (ln & 15)) * 4 + wv] = acc[m][ut][r];
        }
      APS_LSTM_STAMP(5)
      __syncthreads();
      APS_LSTM_STAMP(6)
      if (mine) {
        // the 4 waves' partial sums of a gate row sit side by side: one 16-byte fetch per gate
        const f32x4* rp = reinterpret_cast<const f32x4*>(s_red + gl * RP + (16 * (gu >> 2) + (gu & 3)) * 4);
        const f32x4 t0 = rp[0], t1 = rp[4], t2 = rp[8], t3 = rp[12];
        part[0] = (t0.x + t0.y) + (t0.z + t0.w), part[1] = (t1.x + t1.y) + (t1.z + t1.w);
        part[2] = (t2.x + t2.y) + (t2.z + t2.w), part[3] = (t3.x + t3.y) + (t3.z + t3.w);
        if (F16) {
#pragma unroll
          for (int q = 0; q < 4; ++q) part[q] *= wscale[q];
        }
      }
      APS_LSTM_PIN(part[0]) APS_LSTM_PIN(part[1]) APS_LSTM_PIN(part[2]) APS_LSTM_PIN(part[3])
      APS_LSTM_STAMP(7)
    }
    float h = 0.f;
    if (mine && s < len) {
      const float gi = sigmoid_f(p[0] + part[0] + bias[0]);
      const float gf = sigmoid_f(p[1] + part[1] + bias[1]);
      const float gg = tanh_f(p[2] + part[2] + bias[2]);
      const float go = sigmoid_f(p[3] + part[3] + bias[3]);
      c[g] = gf * c[g] + gi * gg;
      h = go * tanh_f(c[g]);
    }
    APS_LSTM_PIN(h)
    APS_LSTM_STAMP(8)
    // the 4 units of an utterance sit in 4 adjacent lanes: lane gu == 0 stores all 16 bytes
    const float h1 = __shfl_down(h, 1, 64), h2 = __shfl_down(h, 2, 64), h3 = __shfl_down(h, 3, 64);
    // Every lane executes the store (no exec-masked branch: the compiler can then count it in its
    // vmcnt bookkeeping and later waits do not collapse to vmcnt(0), i.e. to waiting for the
    // write-through ack); lanes that have nothing to publish aim outside the buffer, where stores
    // are dropped by the range check.
    {
      const int t_out = (s < len) ? (dir ? len - 1 - s : s) : s;  // padded frames: zeros in place
      const bool pub = mine && (gu & 3) == 0 && dbg != 3;
      const uint32_t off = pub ? (uint32_t)(row_bytes0[g] + t_out * step_bytes) : 0xfffffff0u;
      u32x4 o = {__float_as_uint(h), __float_as_uint(h1), __float_as_uint(h2), __float_as_uint(h3)};
      if (plain_publish)
        __builtin_amdgcn_raw_buffer_store_b128(o, rsrc, off, 0, 0);
      else
        __builtin_amdgcn_raw_buffer_store_b128(o, rsrc, off, 0, 16);
    }
    if (NH == 1 && s + 1 < T) issue(G0{}, s + 1);
#ifdef APS_LSTM_TRACE
    APS_LSTM_STAMP(9)
    if (!FIRST) {
#pragma unroll
      for (int q = 0; q < 9; ++q) tsum[g][q] += stamp[q + 1] - stamp[q];
    }
#endif
    // s_h (per group) / s_red are rewritten only behind later barriers that every wave reaches
    // after it has finished reading them: no extra barrier here
  };

  // input pre-activations are fetched one step ahead (HBM latency off the per-step critical path);
  // the address is clamped, the value unused when s >= len
  auto load_pre = [&](auto gc, int s, float (&p)[4]) {
    constexpr int g = decltype(gc)::value;
    int t_cur = dir ? glen[g] - 1 - s : s;
    t_cur = min(max(t_cur, 0), T - 1);
    const float* pp = pre + ((int64_t)gn_c[g] * T + t_cur) * 4 * H + u0 + (gate_lane ? gu : 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) p[q] = pp[q * H];
  };
  float p[NH][4], pn[NH][4];
  load_pre(G0{}, 0, p[0]);
  load_pre(G0{}, 1, pn[0]);
  if (NH == 2) {
    load_pre(G1{}, 0, p[NH - 1]);
    load_pre(G1{}, 1, pn[NH - 1]);
  }
  step(G0{}, std::true_type{}, 0, p[0]);
  if (NH == 2) step(G1{}, std::true_type{}, 0, p[NH - 1]);
  if (NH == 2 && T > 1) issue(G0{}, 1);
  for (int s = 1; s < T; ++s) {  // uniform body: the vmcnt bookkeeping stays exact across iterations
#pragma unroll
    for (int g = 0; g < NH; ++g)
#pragma unroll
      for (int q = 0; q < 4; ++q) p[g][q] = pn[g][q];
    load_pre(G0{}, s + 1, pn[0]);
    if (NH == 2) load_pre(G1{}, s + 1, pn[NH - 1]);
    step(G0{}, std::false_type{}, s, p[0]);
    if (NH == 2) step(G1{}, std::false_type{}, s, p[NH - 1]);
  }
#ifdef APS_LSTM_TRACE
  if (tslot >= 0 && ln == 0) {
#pragma unroll
    for (int g = 0; g < NH; ++g)
#pragma unroll
      for (int q = 0; q < 9; ++q) g_lstm_trace[((tslot * 2 + g) * 4 + wv) * 16 + q] = tsum[g][q];
    g_lstm_trace[((tslot * 2) * 4 + wv) * 16 + 15] = (unsigned long long)(T - 1);
  }
#endif
}

// ------------------------------------------------------------------------------------------------
// Layer-pipelined stack (unidirectional nn.LSTM, num_layers > 1) in ONE launch: workgroups
// [l G, (l + 1) G) run layer l.  Layer 0 is the kernel above (pre-activations from one batched GEMM);
// a layer l >= 1 consumes the layer below LIVE -- x_t = y_{l-1}[:, t, :] is gathered with the same
// sentinel protocol as its own h_{t-1} = y_l[:, t-1, :], its slices of W_ih and W_hh both live in
// registers and the step contracts K = 2H -- so layer l trails layer l-1 by about one step instead
// of a whole sequence, and the input GEMMs of the upper layers disappear.
// ------------------------------------------------------------------------------------------------
constexpr int kLstmMaxLayers = 4;

struct LstmStackArgs {
  const float* pre0;                    // layer 0: x W_ih^T + b_ih, [N, T, 4H]
  const float* w_ih[kLstmMaxLayers];    // [4H, H] (layers >= 1; entry 0 unused)
  const float* w_hh[kLstmMaxLayers];    // [4H, H]
  const float* b_ih[kLstmMaxLayers];    // [4H] or null (entry 0 unused: inside pre0)
  const float* b_hh[kLstmMaxLayers];    // [4H] or null
  float* y[kLstmMaxLayers];             // [N, T, H] per layer, sentinel filled
  const int64_t* lens;
  unsigned* tmo;
  int32_t N, T, H, L;
  int32_t bsplit;  // batch splits (workgroups along the utterance axis)
};

template <int KREGS, int MT, int UT, bool UPPER>
__device__ __forceinline__ void lstm_stack_body(const LstmStackArgs& a, int layer, int b, int row0,
                                                float* s_dyn) {
  constexpr int H = 16 * KREGS;
  constexpr int UNITS = kLstmUnits * UT, GR = kLstmRows * UT;
  static_assert(16 * MT * UNITS <= 256, "one gate thread per (utterance, unit)");
  constexpr int NSRC = UPPER ? 2 : 1;          // operand = [x_t | h_{t-1}] or [h_{t-1}]
  constexpr bool F16 = APS_LSTM_F16 != 0 && KREGS % 8 == 0;  // (see lstm_layer_kernel)
  constexpr int PITCH = NSRC * H + 4;
  constexpr int PH = NSRC * H + 8;  // f16 row pitch of a plane
  constexpr int NH = (MT % 2 == 0) ? 2 : 1;
  constexpr int MTH = MT / NH, RH = 16 * MTH, CH = H / 4, NL = RH * CH / 256;
  float* s_h = s_dyn;               // [RH][PITCH] (f16: [hi | lo][RH][PH] halves), shared by the groups
  constexpr int RP = 4 * GR + 4;    // floats per utterance row of the reduction buffer
  float* s_red = s_dyn + RH * PH;   // [RH][gate row GR][wave 4] (+ 4 per row)
  const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
  const int u0 = b * UNITS;
  const int N = a.N, T = a.T;

  // resident weight slices (K order permuted per quarter as in lstm_layer_kernel)
  float wreg[F16 ? 1 : NSRC][F16 ? 1 : UT][F16 ? 1 : KREGS];
  f16x8 whi[F16 ? NSRC : 1][F16 ? UT : 1][F16 ? KREGS / 8 : 1];
  f16x8 wlo[F16 ? NSRC : 1][F16 ? UT : 1][F16 ? KREGS / 8 : 1];
  float wscale[4] = {1.f, 1.f, 1.f, 1.f};
  if constexpr (F16) {
    // one exponent per gate row for the row of [W_ih | W_hh] (the step contracts both in one sum)
    int* s_exp = reinterpret_cast<int*>(s_red);
#pragma unroll
    for (int ut = 0; ut < UT; ++ut) {
      const int j = ln & 15;
      const int row = (j >> 2) * H + u0 + 4 * ut + (j & 3);
      float mx = 0.f;
#pragma unroll
      for (int src = 0; src < NSRC; ++src) {
        const float* w = (UPPER && src == 0) ? a.w_ih[layer] : a.w_hh[layer];
        const float* wp = w + (int64_t)row * H + wv * (H / 4) + 8 * (ln >> 4);
#pragma unroll
        for (int m4 = 0; m4 < KREGS / 8; ++m4) mx = fmaxf(mx, lstm_absmax8(wp + 32 * m4));
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      if (ln < 16) s_red[(wv * UT + ut) * 16 + ln] = mx;
    }
    __syncthreads();
    float mrow = 0.f;
    if (tid < UT * 16)
      mrow = fmaxf(fmaxf(s_red[tid], s_red[UT * 16 + tid]),
                   fmaxf(s_red[2 * UT * 16 + tid], s_red[3 * UT * 16 + tid]));
    __syncthreads();
    if (tid < UT * 16) s_exp[tid] = lstm_row_exponent(mrow);
    __syncthreads();
#pragma unroll
    for (int ut = 0; ut < UT; ++ut) {
      const int j = ln & 15;
      const int row = (j >> 2) * H + u0 + 4 * ut + (j & 3);
      const int e = s_exp[ut * 16 + j];
#pragma unroll
      for (int src = 0; src < NSRC; ++src) {
        const float* w = (UPPER && src == 0) ? a.w_ih[layer] : a.w_hh[layer];
        const float* wp = w + (int64_t)row * H + wv * (H / 4) + 8 * (ln >> 4);
#pragma unroll
        for (int m4 = 0; m4 < KREGS / 8; ++m4)
          lstm_weight_planes(wp + 32 * m4, e, whi[src][ut][m4], wlo[src][ut][m4]);
      }
    }
    {
      const int gu_ = tid % UNITS;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        wscale[q] = ldexpf(1.0f, -s_exp[16 * (gu_ >> 2) + q * 4 + (gu_ & 3)]);
    }
    __syncthreads();
  } else {
#pragma unroll
    for (int ut = 0; ut < UT; ++ut) {
      const int j = ln & 15;
      const int row = (j >> 2) * H + u0 + 4 * ut + (j & 3);
#pragma unroll
      for (int src = 0; src < NSRC; ++src) {
        const float* w = (UPPER && src == 0) ? a.w_ih[layer] : a.w_hh[layer];
        const float* wp = w + (int64_t)row * H + wv * (H / 4) + 4 * (ln >> 4);
#pragma unroll
        for (int q = 0; q < KREGS / 4; ++q) {
          const float4 t = *reinterpret_cast<const float4*>(wp + 16 * q);
          wreg[src][ut][4 * q + 0] = t.x, wreg[src][ut][4 * q + 1] = t.y;
          wreg[src][ut][4 * q + 2] = t.z, wreg[src][ut][4 * q + 3] = t.w;
        }
      }
    }
  }
  const int gl = tid / UNITS, gu = tid % UNITS;
  const int gn = row0 + gl;
  const bool gate_thread = gl < 16 * MT && gn < N;
  const int gn_c = min(gn, N - 1);
  const int len = gate_thread ? (a.lens ? (int)min((int64_t)T, max((int64_t)0, a.lens[gn])) : T) : 0;
  float bias[4] = {0.f, 0.f, 0.f, 0.f};
  if (gate_thread) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (a.b_hh[layer]) bias[g] += a.b_hh[layer][g * H + u0 + gu];
      if (UPPER && a.b_ih[layer]) bias[g] += a.b_ih[layer][g * H + u0 + gu];
    }
  }
  float c = 0.f;
  const uint32_t y_bytes = (uint32_t)((int64_t)N * T * H * 4);
  auto rsrc_h = __builtin_amdgcn_make_buffer_rsrc(a.y[layer], 0, y_bytes, 0x00020000);
  auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc(a.y[UPPER ? layer - 1 : layer], 0, y_bytes,
                                                  0x00020000);
  int voff[NH][NL];
  unsigned live_until[NH][NL];
#pragma unroll
  for (int g = 0; g < NH; ++g)
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int idx = tid + 256 * i;
      const int r = row0 + g * RH + idx / CH, q = idx % CH;
      const int rc = min(r, N - 1);
      const int rl = a.lens ? (int)min((int64_t)T, max((int64_t)0, a.lens[rc])) : T;
      live_until[g][i] = (r < N) ? (unsigned)rl : 0u;
      voff[g][i] = (int)((((int64_t)rc * T) * H + 4 * q) * 4);
    }
  const int step_bytes = H * 4;
  // ONE register set for the gathers of both interleaved groups: group g's words are dead once they are
  // split into LDS (before the barrier of its step), and the other group's request goes out behind that
  // barrier -- the two never live at the same time.  As arrays indexed by the group they were allocated
  // side by side (299 VGPRs: beside this kernel a CU then holds ONE wave per SIMD of a 128-register kernel
  // of another stream -- the two-batches-in-flight mode's GEMMs; 240 now: two)
  u32x4 vx[UPPER ? NL : 1], vh[NL];
  bool timed_out = false;

  // requests for step s of group g: x_s (upper layers) and h_{s-1} (s > 0)
  auto issue = [&](auto gc, auto first, int s) {
    constexpr int g = decltype(gc)::value;
    constexpr bool FIRST = decltype(first)::value;
    if (UPPER) {
      const int soff = s * step_bytes;
#pragma unroll
      for (int i = 0; i < NL; ++i)
        vx[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, voff[g][i], soff, 16);
    }
    if (FIRST) {
#pragma unroll
      for (int i = 0; i < NL; ++i) vh[i] = u32x4{0u, 0u, 0u, 0u};
    } else {
      const int soff = (s - 1) * step_bytes;
#pragma unroll
      for (int i = 0; i < NL; ++i)
        vh[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_h, voff[g][i], soff, 16);
    }
  };
  auto finish = [&](auto gc, auto first, int s) {
    constexpr int g = decltype(gc)::value;
    constexpr bool FIRST = decltype(first)::value;
    // fast path (2 instructions per chunk): no sentinel word anywhere in this lane's chunks
    unsigned top = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      if (UPPER)
        top = max(top, max(max(vx[i].x, vx[i].y), max(vx[i].z, vx[i].w)));
      if (!FIRST)
        top = max(top, max(max(vh[i].x, vh[i].y), max(vh[i].z, vh[i].w)));
    }
    if (top != kSentinel) return;
    unsigned live = 0, bad = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) live |= ((unsigned)s < live_until[g][i]) ? (1u << i) : 0u;
    auto missing = [&]() {
      unsigned m = 0;
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        bool hole = false;
        if (UPPER) hole |= has_sentinel(vx[i]);
        if (!FIRST) hole |= has_sentinel(vh[i]);
        m |= (hole && ((live >> i) & 1u)) ? (1u << i) : 0u;
      }
      return m;
    };
    bad = missing();
    unsigned spins = 0;
    // slow path: a producer is behind -- the whole request again, all chunks in flight together
    // (see lstm_layer_kernel)
    while (bad != 0 && !timed_out) {
      if (++spins > kSpinLimit) {
        timed_out = true;
        __hip_atomic_fetch_add(a.tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      issue(gc, first, s);
      bad = missing();
    }
  };
  using G0 = std::integral_constant<int, 0>;
  using G1 = std::integral_constant<int, NH - 1>;
  const int my_group = gl / RH;
  const int row_bytes0 = (int)((((int64_t)gn_c * T) * H + u0 + gu) * 4);

  auto step = [&](auto gc, auto first, int s, const float (&p)[4]) {
    constexpr int g = decltype(gc)::value;
    constexpr bool FIRST = decltype(first)::value;
    float part[4] = {0.f, 0.f, 0.f, 0.f};
    const bool mine = gate_thread && my_group == g;
    if (UPPER || !FIRST) {
      finish(gc, first, s);
      // ONE operand buffer serves both groups: the previous group's MFMAs finished reading it
      // before the barrier in front of its gate phase, which every wave has passed by now
      _Float16* shh = reinterpret_cast<_Float16*>(s_h);  // f16: [hi | lo][RH][PH]
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const int idx = tid + 256 * i;
        if constexpr (F16) {
          _Float16* dst = shh + (idx / CH) * PH + 4 * (idx % CH);
          u32x2 hi, lo;
          if (UPPER) {
            lstm_split4(vx[i], hi, lo);
            *reinterpret_cast<u32x2*>(dst) = hi;
            *reinterpret_cast<u32x2*>(dst + RH * PH) = lo;
          }
          lstm_split4(vh[i], hi, lo);
          *reinterpret_cast<u32x2*>(dst + (NSRC - 1) * H) = hi;
          *reinterpret_cast<u32x2*>(dst + (NSRC - 1) * H + RH * PH) = lo;
        } else {
          float* dst = s_h + (idx / CH) * PITCH + 4 * (idx % CH);
          if (UPPER) *reinterpret_cast<u32x4*>(dst) = vx[i];
          *reinterpret_cast<u32x4*>(dst + (NSRC - 1) * H) = vh[i];
        }
      }
      __syncthreads();
      if (g + 1 < NH) {
        issue(G1{}, first, s);
      } else if (s + 1 < T) {
        issue(G0{}, std::false_type{}, s + 1);
      }
      f32x4 acc[MTH][UT], acc2[MTH][UT];
#pragma unroll
      for (int m = 0; m < MTH; ++m)
#pragma unroll
        for (int ut = 0; ut < UT; ++ut) acc[m][ut] = acc2[m][ut] = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (F16) {
#pragma unroll
        for (int src = 0; src < NSRC; ++src) {
          const _Float16* hp = shh + (ln & 15) * PH + src * H + wv * (H / 4) + 8 * (ln >> 4);
#pragma unroll
          for (int m4 = 0; m4 < KREGS / 8; ++m4) {
#pragma unroll
            for (int m = 0; m < MTH; ++m) {
              const f16x8 ahi = *reinterpret_cast<const f16x8*>(hp + m * 16 * PH + 32 * m4);
              const f16x8 alo = *reinterpret_cast<const f16x8*>(hp + RH * PH + m * 16 * PH + 32 * m4);
#pragma unroll
              for (int ut = 0; ut < UT; ++ut)
                acc[m][ut] = lstm_mfma16(ahi, whi[src][ut][m4], acc[m][ut]);
#pragma unroll
              for (int ut = 0; ut < UT; ++ut)
                acc2[m][ut] = lstm_mfma16(ahi, wlo[src][ut][m4], acc2[m][ut]);
#pragma unroll
              for (int ut = 0; ut < UT; ++ut)
                acc2[m][ut] = lstm_mfma16(alo, whi[src][ut][m4], acc2[m][ut]);
            }
          }
        }
      } else {
#pragma unroll
      for (int src = 0; src < NSRC; ++src) {
        const float* hp = s_h + (ln & 15) * PITCH + src * H + wv * (H / 4) + 4 * (ln >> 4);
#pragma unroll
        for (int q = 0; q < KREGS / 4; ++q) {
#pragma unroll
          for (int m = 0; m < MTH; ++m) {
            const float4 t = *reinterpret_cast<const float4*>(hp + m * 16 * PITCH + 16 * q);
#pragma unroll
            for (int ut = 0; ut < UT; ++ut) {
              acc[m][ut] = __builtin_amdgcn_mfma_f32_16x16x4f32(t.x, wreg[src][ut][4 * q + 0], acc[m][ut], 0, 0, 0);
              acc2[m][ut] = __builtin_amdgcn_mfma_f32_16x16x4f32(t.y, wreg[src][ut][4 * q + 1], acc2[m][ut], 0, 0, 0);
              acc[m][ut] = __builtin_amdgcn_mfma_f32_16x16x4f32(t.z, wreg[src][ut][4 * q + 2], acc[m][ut], 0, 0, 0);
              acc2[m][ut] = __builtin_amdgcn_mfma_f32_16x16x4f32(t.w, wreg[src][ut][4 * q + 3], acc2[m][ut], 0, 0, 0);
            }
          }
        }
      }
      }
#pragma unroll
      for (int m = 0; m < MTH; ++m)
#pragma unroll
        for (int ut = 0; ut < UT; ++ut) {
          acc[m][ut] += acc2[m][ut];
#pragma unroll
          for (int r = 0; r < 4; ++r)
            s_red[(m * 16 + 4 * (ln >> 4) + r) * RP + (16 * ut + (ln & 15)) * 4 + wv] = acc[m][ut][r];
        }
      __syncthreads();
      if (mine) {
        // the 4 waves' partial sums of a gate row sit side by side: one 16-byte fetch per gate
        const f32x4* rp = reinterpret_cast<const f32x4*>(s_red + (gl - g * RH) * RP + (16 * (gu >> 2) + (gu & 3)) * 4);
        const f32x4 t0 = rp[0], t1 = rp[4], t2 = rp[8], t3 = rp[12];
        part[0] = (t0.x + t0.y) + (t0.z + t0.w), part[1] = (t1.x + t1.y) + (t1.z + t1.w);
        part[2] = (t2.x + t2.y) + (t2.z + t2.w), part[3] = (t3.x + t3.y) + (t3.z + t3.w);
        if (F16) {
#pragma unroll
          for (int q = 0; q < 4; ++q) part[q] *= wscale[q];
        }
      }
    }
    float h = 0.f;
    if (mine && s < len) {
      const float gi = sigmoid_f(p[0] + part[0] + bias[0]);
      const float gf = sigmoid_f(p[1] + part[1] + bias[1]);
      const float gg = tanh_f(p[2] + part[2] + bias[2]);
      const float go = sigmoid_f(p[3] + part[3] + bias[3]);
      c = gf * c + gi * gg;
      h = go * tanh_f(c);
    }
    const float h1 = __shfl_down(h, 1, 64), h2 = __shfl_down(h, 2, 64), h3 = __shfl_down(h, 3, 64);
    {
      const bool pub = mine && (gu & 3) == 0;
      const uint32_t off = pub ? (uint32_t)(row_bytes0 + s * step_bytes) : 0xfffffff0u;
      u32x4 o = {__float_as_uint(h), __float_as_uint(h1), __float_as_uint(h2), __float_as_uint(h3)};
      __builtin_amdgcn_raw_buffer_store_b128(o, rsrc_h, off, 0, 16);
    }
  };

  auto load_pre = [&](int s, float (&p)[4]) {
    if (UPPER) {
#pragma unroll
      for (int g = 0; g < 4; ++g) p[g] = 0.f;
    } else {
      const int t_cur = min(s, T - 1);
      const float* pp = a.pre0 + ((int64_t)gn_c * T + t_cur) * 4 * H + u0 + gu;
#pragma unroll
      for (int g = 0; g < 4; ++g) p[g] = pp[g * H];
    }
  };
  float p[4], pn[4];
  load_pre(0, p);
  load_pre(1, pn);
  if (UPPER) issue(G0{}, std::true_type{}, 0);
  step(G0{}, std::true_type{}, 0, p);
  if (NH == 2) step(G1{}, std::true_type{}, 0, p);
  if (!UPPER && T > 1) issue(G0{}, std::false_type{}, 1);
  for (int s = 1; s < T; ++s) {
#pragma unroll
    for (int g = 0; g < 4; ++g) p[g] = pn[g];
    load_pre(s + 1, pn);
    step(G0{}, std::false_type{}, s, p);
    if (NH == 2) step(G1{}, std::false_type{}, s, p);
  }
}

struct LstmShape {
  int mt, ut;
};

// Launches that may run at the same time (graph replicas on separate streams, aps_amd/replicas.py)
// share the chip: the caller passes `share` = how many such launches can be in flight at once, and
// each sizes its grid for 1 / share of the resident slots.  The grid it leads to is what a stream
// capture records.
//
// The hand-off protocol needs every workgroup of a launch resident at once -- and those of every
// launch running beside it: two half-resident grids would wait on each other's missing workgroups.
// The residency capacity of a kernel is a property of the device: cached per device.
template <typename K>
static bool lstm_fits(K kernel, int grid, size_t lds, int share, ApsPerDevice& capacity) {
  const int dev = aps_current_device();
  if (dev < 0) return false;
  int cap = capacity.get(dev);
  if (cap == 0) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, lds) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      return false;
    cap = per_cu * cus;
    if (cap <= 0) return false;
    capacity.set(dev, cap);
  }
  return (int64_t)grid * share <= cap;
}

// compute units of the current device (the "one workgroup per CU" target of the shape rule)
static int lstm_device_cus() {
  static ApsPerDevice cus;
  const int dev = aps_current_device();
  if (dev < 0) return 256;
  int n = cus.get(dev);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      return 256;
    cus.set(dev, n);
  }
  return n;
}

template <int KREGS, int MT, int UT>
__global__ __launch_bounds__(256) void lstm_stack_kernel(LstmStackArgs a) {
  extern __shared__ __attribute__((aligned(16))) float s_stack[];
  constexpr int G = 16 * KREGS / (kLstmUnits * UT);
  const int per_layer = G * a.bsplit;
  const int layer = blockIdx.x / per_layer, rem = blockIdx.x % per_layer;
  const int b = rem % G, row0 = (rem / G) * 16 * MT;
  if (layer == 0)
    lstm_stack_body<KREGS, MT, UT, false>(a, 0, b, row0, s_stack);
  else
    lstm_stack_body<KREGS, MT, UT, true>(a, layer, b, row0, s_stack);
}

template <int KREGS, int MT, int UT>
static int launch_lstm_stack(LstmStackArgs a, int share, hipStream_t st) {
  constexpr int H = 16 * KREGS;
  constexpr int NH = (MT % 2 == 0) ? 2 : 1;
  constexpr int RH = 16 * MT / NH;
  a.bsplit = (a.N + 16 * MT - 1) / (16 * MT);
  const int grid = a.L * (H / (kLstmUnits * UT)) * a.bsplit;
  const size_t lds = (size_t)RH * (2 * H + 8 + 4 * (kLstmRows * UT + 1)) * sizeof(float);
  if (lds > 160 * 1024) return APS_ERR_UNSUPPORTED;
  // once per device: not legal inside a stream capture (the first call must be an eager one)
  static ApsPerDevice attr_set, capacity;
  if (lds > 64 * 1024 &&
      !aps_lds_opt_in(attr_set, reinterpret_cast<const void*>(&lstm_stack_kernel<KREGS, MT, UT>),
                      160 * 1024))
    return APS_ERR_LAUNCH;
  if (!lstm_fits(lstm_stack_kernel<KREGS, MT, UT>, grid, lds, share, capacity))
    return APS_ERR_UNSUPPORTED;
  // sentinel fill: one memset when the layers' outputs are one allocation (as the binding makes them)
  const size_t layer_bytes = (size_t)a.N * a.T * H * sizeof(float);
  bool one_block = true;
  for (int l = 1; l < a.L; ++l)
    one_block &= reinterpret_cast<char*>(a.y[l]) == reinterpret_cast<char*>(a.y[l - 1]) + layer_bytes;
  // (a fill KERNEL, not hipMemsetAsync: memset nodes are unreliable under graph replay, common.h)
  if (one_block) {
    if (aps_fill_u32(a.y[0], kSentinel, layer_bytes * a.L / 4, st) != APS_OK) return APS_ERR_LAUNCH;
  } else {
    for (int l = 0; l < a.L; ++l)
      if (aps_fill_u32(a.y[l], kSentinel, layer_bytes / 4, st) != APS_OK) return APS_ERR_LAUNCH;
  }
  hipLaunchKernelGGL((lstm_stack_kernel<KREGS, MT, UT>), dim3(grid), dim3(256), lds, st, a);
  return aps_launch_status();
}

// Shape of the decomposition for (H, N, groups).  Measured on MI355X (scripts/lstm_shape_probe2.py,
// H = 512, two groups, us per step): N = 64: (MT, UT) = (4,1) 7.9, (2,1) 9.3, (2,2) 5.6, (1,4) 5.7;
// N = 32: (2,1) 4.8, (1,2) 5.1, (2,2) 5.7, (1,4) 5.9.  Rule that reproduces the winners: two row
// tiles per workgroup when there are two (their hand-off waits interleave), then the widest unit
// block that still yields one workgroup per CU.  APS_LSTM_SHAPE="MT,UT" overrides (tuning).
static LstmShape pick_lstm_shape(int H, int N, int groups, int k_factor, int max_regs,
                                 bool shared_gate_threads, int share) {
  auto legal = [&](int mt, int ut) {
    if (mt < 1 || mt > 4 || (ut != 1 && ut != 2 && ut != 4)) return false;
    // gate threads: one per (utterance, unit) of a workgroup -- of ONE of the two interleaved row
    // groups where the kernel shares them between the groups (lstm_layer_kernel)
    const int nh = (shared_gate_threads && mt % 2 == 0 && (mt / 2) * H / 64 <= 16) ? 2 : 1;
    return (mt / nh) * ut <= 4 && H % (4 * ut) == 0 && k_factor * ut * (H / 16) <= max_regs;
  };
  if (const char* e = getenv("APS_LSTM_SHAPE")) {
    int mt = 0, ut = 0;
    if (sscanf(e, "%d,%d", &mt, &ut) == 2 && legal(mt, ut)) return {mt, ut};
  }
  const int tiles = (N + 15) / 16;
  auto wgs = [&](int mt, int ut) { return groups * (H / (4 * ut)) * ((tiles + mt - 1) / mt); };
  const int slots = max(1, lstm_device_cus() / share);
  int mt = tiles >= 2 ? 2 : 1, ut = 1;
  for (int u = 4; u >= 2; u >>= 1)
    if (legal(mt, u) && wgs(mt, u) >= slots) {
      ut = u;
      break;
    }
  // too many workgroups for one per CU: more rows per workgroup
  while (wgs(mt, ut) > slots && legal(mt + 1, ut) && mt + 1 <= tiles) ++mt;
  if (!legal(mt, ut)) return {0, 0};
  return {mt, ut};
}

template <int KREGS, int MT, int UT>
static int launch_lstm_shape(LstmArgs a, int dirs, int share, hipStream_t st) {
  constexpr int H = 16 * KREGS;
  a.bsplit = (a.N + 16 * MT - 1) / (16 * MT);
  const int grid = dirs * (H / (kLstmUnits * UT)) * a.bsplit;
  const size_t lds = (size_t)(16 * MT) * (H + 8 + 4 * (kLstmRows * UT + 1)) * sizeof(float);
  if (lds > 160 * 1024) return APS_ERR_UNSUPPORTED;
  // once per device: not legal inside a stream capture (the first call must be an eager one)
  static ApsPerDevice attr_set, capacity;
  if (lds > 64 * 1024 &&
      !aps_lds_opt_in(attr_set, reinterpret_cast<const void*>(&lstm_layer_kernel<KREGS, MT, UT>),
                      160 * 1024))
    return APS_ERR_LAUNCH;
  if (!lstm_fits(lstm_layer_kernel<KREGS, MT, UT>, grid, lds, share, capacity))
    return APS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((lstm_layer_kernel<KREGS, MT, UT>), dim3((a.debug & 4) ? 8 * grid : grid), dim3(256),
                     lds, st, a);
  return aps_launch_status();
}

constexpr int kLstmMaxWeightRegs = 128;  // resident weight values per lane
constexpr int kLstmTeamSlots = 1024;     // placement tables in the workspace (one per launch, cycled)
constexpr int kLstmTeamWords = 8 * 32;

// Team form of the layer kernel (lstm_layer_kernel<.., TEAM = true>): 8 teams of 32 workgroups, team
// = block id % 8 = (group, batch split of 16 MT utterances), member = unit block of H / 32 units.
// Every launch gets its own zeroed placement table (slots are cycled: two launches in flight would
// have to be kLstmTeamSlots launches apart to share one, and then only mis-judge their placement,
// which costs the short path or a reported timeout, never a wrong value).
template <int KREGS, int MT, int OCC>
static int launch_lstm_team(LstmArgs a, int dirs, int share, hipStream_t st) {
  constexpr int H = 16 * KREGS, UT = KREGS / 8;
  a.bsplit = (a.N + 16 * MT - 1) / (16 * MT);
  a.groups = dirs;
  if (dirs * a.bsplit > 8) return APS_ERR_UNSUPPORTED;
  const int grid = 8 * 32;
  const size_t lds = (size_t)(16 * MT) * (H + 8 + 4 * (kLstmRows * UT + 1)) * sizeof(float);
  static ApsPerDevice attr_set, capacity;
  if (lds > 64 * 1024 &&
      !aps_lds_opt_in(attr_set, reinterpret_cast<const void*>(&lstm_layer_kernel<KREGS, MT, UT, true, OCC>),
                      160 * 1024))
    return APS_ERR_LAUNCH;
  if (!lstm_fits(lstm_layer_kernel<KREGS, MT, UT, true, OCC>, grid, lds, share, capacity))
    return APS_ERR_UNSUPPORTED;
  static std::atomic<unsigned> next_slot{0};
  a.team = a.tmo + 4 + (size_t)(next_slot.fetch_add(1) % kLstmTeamSlots) * kLstmTeamWords;
  if (aps_fill_u32(a.team, 0u, kLstmTeamWords, st) != APS_OK) return APS_ERR_LAUNCH;
  hipLaunchKernelGGL((lstm_layer_kernel<KREGS, MT, UT, true, OCC>), dim3(grid), dim3(256), lds, st, a);
  return aps_launch_status();
}

template <int KREGS>
static int launch_lstm(const LstmArgs& a, int dirs, int share, hipStream_t st) {
  constexpr int H = 16 * KREGS;
  const LstmShape sh = pick_lstm_shape(H, a.N, dirs, 1, kLstmMaxWeightRegs, true, share);
  if (sh.mt == 0) return APS_ERR_UNSUPPORTED;
  // every word of y = sentinel ("not written yet")
  // (a fill KERNEL, not hipMemsetAsync: memset nodes are unreliable under graph replay, common.h)
  if (aps_fill_u32(a.y, kSentinel, (size_t)a.N * a.T * a.ldy, st) != APS_OK) return APS_ERR_LAUNCH;
  // the team form where the geometry has one (H = 128 / 256 / 512, at most 8 (group, 16- or 32-row
  // batch split) pairs) and the launch fits beside the `share` - 1 others; APS_LSTM_TEAM=0: off
  if constexpr (KREGS == 8 || KREGS == 16 || KREGS == 32) {
    static const bool team_on = [] {
      const char* e = getenv("APS_LSTM_TEAM");
      return !(e && e[0] == '0');
    }();
    if (team_on && !getenv("APS_LSTM_SHAPE")) {
      const int tiles = (a.N + 15) / 16;
      int rc = APS_ERR_UNSUPPORTED;
      // (one launch of the team form covers the chip: `share` of them need `share` per CU)
      if (share == 1) {
        if (dirs * tiles <= 8) rc = launch_lstm_team<KREGS, 1, 1>(a, dirs, share, st);
        if (rc == APS_ERR_UNSUPPORTED && dirs * ((tiles + 1) / 2) <= 8)
          rc = launch_lstm_team<KREGS, 2, 1>(a, dirs, share, st);
      } else if (share == 2) {
        // (two per CU.  At H = 512 the 16-row team kernel held to half a SIMD's registers spills 28
        // of them and still beats the spread form -- joint 17.6 - 18.6k against 16.8 - 16.9k utt/s,
        // one stream 8.9 against 9.7 ms --; the 32-row one would spill 100+ and is left out)
        if (dirs * tiles <= 8) rc = launch_lstm_team<KREGS, 1, 2>(a, dirs, share, st);
        if constexpr (KREGS < 32) {
          if (rc == APS_ERR_UNSUPPORTED && dirs * ((tiles + 1) / 2) <= 8)
            rc = launch_lstm_team<KREGS, 2, 2>(a, dirs, share, st);
        }
      }
      if (rc != APS_ERR_UNSUPPORTED) return rc;
    }
  }
  if (sh.ut == 1) {
    switch (sh.mt) {
      case 1: return launch_lstm_shape<KREGS, 1, 1>(a, dirs, share, st);
      case 2: return launch_lstm_shape<KREGS, 2, 1>(a, dirs, share, st);
      case 3: return launch_lstm_shape<KREGS, 3, 1>(a, dirs, share, st);
      default: return launch_lstm_shape<KREGS, 4, 1>(a, dirs, share, st);
    }
  }
  constexpr bool kPairs = 2 * (16 * KREGS) / 64 <= 16;  // MT = 4 runs as two interleaved pairs
  if constexpr (2 * KREGS <= kLstmMaxWeightRegs) {
    if (sh.ut == 2 && sh.mt == 1) return launch_lstm_shape<KREGS, 1, 2>(a, dirs, share, st);
    if (sh.ut == 2 && sh.mt == 2) return launch_lstm_shape<KREGS, 2, 2>(a, dirs, share, st);
    if constexpr (kPairs) {
      if (sh.ut == 2 && sh.mt == 4) return launch_lstm_shape<KREGS, 4, 2>(a, dirs, share, st);
    }
  }
  if constexpr (4 * KREGS <= kLstmMaxWeightRegs) {
    if (sh.ut == 4 && sh.mt == 1) return launch_lstm_shape<KREGS, 1, 4>(a, dirs, share, st);
    if (sh.ut == 4 && sh.mt == 2) return launch_lstm_shape<KREGS, 2, 4>(a, dirs, share, st);
  }
  return APS_ERR_UNSUPPORTED;
}

}  // namespace aps

using namespace aps;

extern "C" int64_t aps_lstm_workspace(int64_t H) {
  if (H <= 0 || H % 64) return -1;
  return 16 + (int64_t)kLstmTeamSlots * kLstmTeamWords * 4;  // status words + placement tables
}

extern "C" int aps_lstm_layer(const float* pre_fwd, const float* pre_bwd, const float* w_hh_fwd,
                              const float* w_hh_bwd, const float* b_hh_fwd, const float* b_hh_bwd,
                              const int64_t* lens, float* y, int64_t N, int64_t T, int64_t H,
                              int32_t second_reverse, int32_t share, void* workspace,
                              void* stream) {
  APS_CHECK_ARG(pre_fwd && w_hh_fwd && y && workspace && N > 0 && T > 0 && H > 0 && share >= 1);
  APS_CHECK_ARG((pre_bwd == nullptr) == (w_hh_bwd == nullptr));
  APS_CHECK_ARG(((uintptr_t)y & 15) == 0);
  const int dirs = pre_bwd ? 2 : 1;
  const int64_t ldy = dirs * H;
  if (N > 128 || N * T * ldy * 4 >= ((int64_t)1 << 31)) return APS_ERR_UNSUPPORTED;
  LstmArgs a{{pre_fwd, pre_bwd}, {w_hh_fwd, w_hh_bwd}, {b_hh_fwd, b_hh_bwd}, lens, y,
             static_cast<unsigned*>(workspace), (int32_t)N, (int32_t)T, (int32_t)H, (int32_t)ldy, 1,
             second_reverse ? 1 : 0, 0, dirs, nullptr};
  if (const char* e = getenv("APS_LSTM_DEBUG")) a.debug = atoi(e);
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (H) {
    case 64: return launch_lstm<4>(a, dirs, share, st);
    case 128: return launch_lstm<8>(a, dirs, share, st);
    case 256: return launch_lstm<16>(a, dirs, share, st);
    case 320: return launch_lstm<20>(a, dirs, share, st);
    case 384: return launch_lstm<24>(a, dirs, share, st);
    case 512: return launch_lstm<32>(a, dirs, share, st);
    case 640: return launch_lstm<40>(a, dirs, share, st);
    case 768: return launch_lstm<48>(a, dirs, share, st);
    case 1024: return launch_lstm<64>(a, dirs, share, st);
    default: return APS_ERR_UNSUPPORTED;
  }
}

// 1 when a bounded spin of ANY launch that used this workspace since the caller zeroed it expired
// (a workgroup was not resident, or an input NaN reproduced the sentinel); that launch's output
// holds NaNs.  Reads the sticky counter with a blocking copy.
extern "C" int aps_lstm_timed_out(const void* workspace, void* stream) {
  if (!workspace) return APS_ERR_INVALID;
  unsigned v = 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (hipMemcpyAsync(&v, workspace, sizeof(v), hipMemcpyDeviceToHost, st) != hipSuccess)
    return APS_ERR_LAUNCH;
  if (hipStreamSynchronize(st) != hipSuccess) return APS_ERR_LAUNCH;
  return v != 0;
}

// Unidirectional multi-layer stack in one launch (layers pipelined, see lstm_stack_kernel).
// Restricted to the register / residency budget it was sized for: H in {128, 256, 512}, N <= 32,
// 2 <= L <= 4, L * H / 4 workgroups resident; otherwise APS_ERR_UNSUPPORTED and the caller runs the
// layers one launch at a time.
extern "C" int aps_lstm_stack(const float* pre0, const float* const* w_ih, const float* const* w_hh,
                              const float* const* b_ih, const float* const* b_hh,
                              const int64_t* lens, float* const* y, int64_t N, int64_t T, int64_t H,
                              int64_t L, int32_t share, void* workspace, void* stream) {
  APS_CHECK_ARG(pre0 && w_ih && w_hh && b_ih && b_hh && y && workspace && N > 0 && T > 0 &&
                share >= 1);
  if (L < 2 || L > kLstmMaxLayers || N > 64 || N * T * H * 4 >= ((int64_t)1 << 31))
    return APS_ERR_UNSUPPORTED;
  LstmStackArgs a{};
  a.pre0 = pre0;
  for (int l = 0; l < L; ++l) {
    APS_CHECK_ARG(w_hh[l] && y[l] && (l == 0 || w_ih[l]) && ((uintptr_t)y[l] & 15) == 0);
    a.w_ih[l] = w_ih[l];
    a.w_hh[l] = w_hh[l];
    a.b_ih[l] = b_ih[l];
    a.b_hh[l] = b_hh[l];
    a.y[l] = y[l];
  }
  a.lens = lens;
  a.tmo = static_cast<unsigned*>(workspace);
  a.N = (int32_t)N, a.T = (int32_t)T, a.H = (int32_t)H, a.L = (int32_t)L;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // upper layers hold W_ih and W_hh slices: 2 UT H/16 values per lane
  const LstmShape sh = pick_lstm_shape((int)H, (int)N, (int)L, 2, kLstmMaxWeightRegs, false, share);
#define APS_STACK_CASE(KR)                                                                  \
  case 16 * KR:                                                                             \
    if (sh.ut == 1) {                                                                       \
      if (sh.mt == 1) return launch_lstm_stack<KR, 1, 1>(a, share, st);                            \
      if (sh.mt == 2) return launch_lstm_stack<KR, 2, 1>(a, share, st);                            \
    }                                                                                       \
    if constexpr (4 * KR <= kLstmMaxWeightRegs) {                                           \
      if (sh.ut == 2 && sh.mt == 1) return launch_lstm_stack<KR, 1, 2>(a, share, st);              \
      if (sh.ut == 2 && sh.mt == 2) return launch_lstm_stack<KR, 2, 2>(a, share, st);              \
    }                                                                                       \
    if constexpr (8 * KR <= kLstmMaxWeightRegs) {                                           \
      if (sh.ut == 4 && sh.mt == 1) return launch_lstm_stack<KR, 1, 4>(a, share, st);              \
    }                                                                                       \
    return APS_ERR_UNSUPPORTED;
  switch (H) {
    APS_STACK_CASE(4)
    APS_STACK_CASE(8)
    APS_STACK_CASE(16)
    APS_STACK_CASE(32)
    default:
      return APS_ERR_UNSUPPORTED;
  }
#undef APS_STACK_CASE
}

#ifdef APS_LSTM_TRACE
extern "C" int aps_debug_lstm_trace(void* host, int64_t bytes) {
  if (bytes > (int64_t)sizeof(aps::g_lstm_trace)) bytes = sizeof(aps::g_lstm_trace);
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(aps::g_lstm_trace), (size_t)bytes) == hipSuccess
             ? APS_OK : APS_ERR_LAUNCH;
}
#endif
