// Encoder kernels for gfx950: the dense contractions of the transformer encoder on fp32 MFMA
// (exact f32 products / accumulation -- the 1e-4 parity bar rules out bf16) with the bias / ReLU /
// residual epilogues fused in, plus the row-wise LayerNorm, the sinusoid position add and the
// scaled-dot-product attention core.
//
// Replaces the torch ops behind aps/asr/transformer/impl.py:147-185, 377-429, 718-756,
// aps/asr/transformer/pose.py:29-118 and the Linear layers of aps/asr/base/encoder.py:367-441.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace aps {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------
// C[M, N] = act(A[M, K] . W[N, K]^T + bias[N]) * alpha + residual[M, N]
//
// TM x TN output tile per workgroup, 4 wavefronts in a 2 x 2 grid, each owning (TM/2) x (TN/2) as
// 32 x 32 v_mfma_f32_32x32x2_f32 tiles (exact fp32).  K is consumed BK = 32 at a time through LDS.
// Both operands are K-contiguous in HBM (activations [M, K], nn.Linear weights [N, K]) and are kept
// row-major in LDS ([rows][BK + 4]): staging is a straight 16-byte copy (8 lanes cover one 128-byte
// row segment), and the MFMA operands are fetched 4 k at a time with one ds_read_b128 -- lanes
// 0-31 take k = 8 g .. 8 g + 3 of their row, lanes 32-63 k = 8 g + 4 .. 8 g + 7, i.e. the k order
// inside a group of 8 is permuted identically for A and B, which a sum over k does not see.
// Pipeline: LDS double buffered, global loads register-staged TWO tiles ahead (tile s + 2 is
// requested before the MFMAs of tile s), so ~2 compute phases cover the L2/HBM latency even with
// one workgroup per CU -- the regime of the encoder GEMMs (M = batch x frames ~ 2 k rows).
// The 64 x 64 tile (94 VGPRs, 37 KB LDS: up to 4-5 workgroups per CU) is the production shape --
// occupancy hides what the barriers expose; with a grid that is a multiple of 8 the linear block id
// is remapped so that each XCD (block id mod 8) owns a contiguous range of row panels (A read by
// one L2 only).
// ------------------------------------------------------------------------------------------

struct GemmArgs {
  const float* A;
  const float* W;
  const float* bias;      // [N] or null
  const float* residual;  // [M, N] (ldc) or null
  float* C;
  int64_t M, N, K;
  int64_t lda, ldw, ldc;
  int32_t act;  // 0 none, 1 relu, 2 swish (x * sigmoid(x)), 3 sigmoid, 4 tanh, 5 gelu (erf form)
  float alpha;  // C = act(A W^T + bias) * alpha + residual
  int32_t tiles_n, remap;
  // LayerNorm of the A rows folded into this GEMM (LN template flag):
  //   LN(x) W^T + b = rstd_r (x W'^T - mean_r cs) + b',  W' = W diag(gamma), cs[n] = sum_k W'[n,k],
  //   b' = b + W beta  (W', cs, b' prepared on the host: W = W', bias = b', ln_cs = cs)
  // the MFMA loop runs on the raw rows; mean_r / rstd_r come from the row sums the staging threads
  // accumulate while the tiles pass through their registers (no extra pass over x, no LN launch).
  const float* ln_cs;
  float ln_eps;
};

// SWP (64 x 64 tile): the operand fetches are software pipelined by hand -- the LDS reads of one
// half of a K tile are issued a whole group of 8 MFMAs ahead of their use (two operand register
// sets), the LDS writes of the next tile before the first MFMA group, the barrier between the two
// groups: no LDS latency and no write drain is left in front of an MFMA.
template <int TM, int TN, int kBK, int WAVES_PER_SIMD, bool LN = false, bool SWP = false>
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void gemm_f32_kernel(GemmArgs g) {
  constexpr int kPitch = kBK + 4;   // 16-byte aligned rows, 4 r mod 64 banks
  constexpr int kRowF4 = kBK / 4;   // float4 per tile row
  constexpr int kRPP = 256 / kRowF4;  // rows staged per pass of the 256 threads
  constexpr int WM = TM / 2, WN = TN / 2, SM = WM / 32, SN = WN / 32;
  constexpr int LA = TM / kRPP, LB = TN / kRPP;  // float4 per thread and tile
  constexpr int kBufFloats = (TM + TN) * kPitch;
  extern __shared__ __attribute__((aligned(16))) float s_gemm[];  // [2][TM + TN][kPitch]
  const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
  const int wm = wv >> 1, wn = wv & 1;
  int64_t lin = blockIdx.x;
  if (g.remap) {  // XCD x (= lin % 8) takes the x-th contiguous eighth of the tile list
    const int64_t per = gridDim.x / 8;
    lin = (lin & 7) * per + (lin >> 3);
  }
  const int64_t m0 = (lin / g.tiles_n) * TM, n0 = (lin % g.tiles_n) * TN;
  // staging role: float4 sc / 4 of row sr (+ kRPP i); consecutive lanes cover a row's BK floats
  const int sr = tid / kRowF4, sc = (tid % kRowF4) * 4;

  // two accumulators per tile when the wave owns a single tile: consecutive MFMAs then never
  // depend on each other (a 16-long dependent chain costs ~25 % per K step otherwise)
  constexpr int NACC = (SM * SN == 1) ? 2 : 1;
  f32x16 acc[SM][SN][NACC];
#pragma unroll
  for (int i = 0; i < SM; ++i)
#pragma unroll
    for (int j = 0; j < SN; ++j)
#pragma unroll
      for (int q = 0; q < NACC; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][q][e] = 0.f;

  u32x4 ra[2][LA], rb[2][LB];
  const int nfull = (int)(g.K / kBK);  // full K tiles; a remainder (< BK) is handled after the loop
  // Loads go through buffer descriptors: the per-thread part of the address (row, float4 column)
  // is a 32-bit VGPR offset computed once, the K position of the tile is the instruction's scalar
  // offset -- no per-step vector address arithmetic, and the steady-state loop carries no branch
  // (requests past the last full tile are clamped to it and never consumed), so the compiler's
  // vmcnt bookkeeping keeps the two-tiles-ahead loads in flight instead of draining them.
  // Rows past the edge are clamped to the last valid one (their products are never stored).
  auto rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0,
                                                  (uint32_t)(g.M * g.lda * 4), 0x00020000);
  auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.W), 0,
                                                  (uint32_t)(g.N * g.ldw * 4), 0x00020000);
  int32_t va[LA], vb[LB];  // byte offset of (row, this thread's float4 column)
#pragma unroll
  for (int i = 0; i < LA; ++i)
    va[i] = (int32_t)(min(m0 + sr + kRPP * i, g.M - 1) * g.lda * 4) + sc * 4;
#pragma unroll
  for (int i = 0; i < LB; ++i)
    vb[i] = (int32_t)(min(n0 + sr + kRPP * i, g.N - 1) * g.ldw * 4) + sc * 4;

  auto gload = [&](auto stage, int step) {
    constexpr int P = decltype(stage)::value;
    const int32_t soff = min(step, nfull - 1) * (kBK * 4);
#pragma unroll
    for (int i = 0; i < LA; ++i) ra[P][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, va[i], soff, 0);
#pragma unroll
    for (int i = 0; i < LB; ++i) rb[P][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, vb[i], soff, 0);
  };
  // K remainder: the column is clamped inside the row (lda, ldw are multiples of 4 >= K) and the
  // components with k >= K are zeroed by selects
  auto tail4 = [&](u32x4 v, int64_t k) {
    v.x = (k + 0 < g.K) ? v.x : 0u;
    v.y = (k + 1 < g.K) ? v.y : 0u;
    v.z = (k + 2 < g.K) ? v.z : 0u;
    v.w = (k + 3 < g.K) ? v.w : 0u;
    return v;
  };
  auto gload_tail = [&](auto stage) {
    constexpr int P = decltype(stage)::value;
    const int64_t k = (int64_t)nfull * kBK + sc;
    const int32_t ca = (int32_t)(min(k, g.lda - 4) - sc) * 4, cw = (int32_t)(min(k, g.ldw - 4) - sc) * 4;
#pragma unroll
    for (int i = 0; i < LA; ++i)
      ra[P][i] = tail4(__builtin_amdgcn_raw_buffer_load_b128(rsrc_a, va[i] + ca, 0, 0), k);
#pragma unroll
    for (int i = 0; i < LB; ++i)
      rb[P][i] = tail4(__builtin_amdgcn_raw_buffer_load_b128(rsrc_w, vb[i] + cw, 0, 0), k);
  };
  // LN: this thread's share of sum x / sum x^2 of its staged rows, as two-lane partials (packed
  // fp32 instructions: 5 per float4, branch free -- `keep` = 0 for the clamped re-request of the
  // last tile, so the K loop carries no branch and the arithmetic stays inside an MFMA's shadow)
  f32x2 ln_s1[LA], ln_s2[LA];
#pragma unroll
  for (int i = 0; i < LA; ++i) ln_s1[i] = ln_s2[i] = f32x2{0.f, 0.f};
  auto ln_accumulate = [&](int i, u32x4 v, bool fresh) {
    const float keep = fresh ? 1.0f : 0.0f;
    const f32x2 k2 = {keep, keep};
    const f32x2 a = k2 * f32x2{__uint_as_float(v.x), __uint_as_float(v.y)};
    const f32x2 b = k2 * f32x2{__uint_as_float(v.z), __uint_as_float(v.w)};
    ln_s1[i] = (a + b) + ln_s1[i];
    ln_s2[i] = __builtin_elementwise_fma(a, a, ln_s2[i]);
    ln_s2[i] = __builtin_elementwise_fma(b, b, ln_s2[i]);
  };
  auto sstore = [&](auto stage, int buf, bool fresh = true) {
    constexpr int P = decltype(stage)::value;
    float* sa = s_gemm + buf * kBufFloats;
    float* sb = sa + TM * kPitch;
    if (LN) {  // `fresh`: not the clamped re-request of the last tile
#pragma unroll
      for (int i = 0; i < LA; ++i) ln_accumulate(i, ra[P][i], fresh);
    }
#pragma unroll
    for (int i = 0; i < LA; ++i)
      *reinterpret_cast<u32x4*>(sa + (sr + kRPP * i) * kPitch + sc) = ra[P][i];
#pragma unroll
    for (int i = 0; i < LB; ++i)
      *reinterpret_cast<u32x4*>(sb + (sr + kRPP * i) * kPitch + sc) = rb[P][i];
  };
  const int frow = ln & 31, fk = (ln >> 5) * 4;
  auto compute = [&](int buf) {
    const float* sa = s_gemm + buf * kBufFloats + (wm * WM + frow) * kPitch + fk;
    const float* sb = s_gemm + buf * kBufFloats + (TM + wn * WN + frow) * kPitch + fk;
#pragma unroll
    for (int kg = 0; kg < kBK / 8; ++kg) {
      float a[SM][4], b[SN][4];
#pragma unroll
      for (int i = 0; i < SM; ++i) {
        const float4 t = *reinterpret_cast<const float4*>(sa + i * 32 * kPitch + kg * 8);
        a[i][0] = t.x, a[i][1] = t.y, a[i][2] = t.z, a[i][3] = t.w;
      }
#pragma unroll
      for (int j = 0; j < SN; ++j) {
        const float4 t = *reinterpret_cast<const float4*>(sb + j * 32 * kPitch + kg * 8);
        b[j][0] = t.x, b[j][1] = t.y, b[j][2] = t.z, b[j][3] = t.w;
      }
      // k outermost: consecutive MFMAs hit different accumulators
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < SM; ++i)
#pragma unroll
          for (int j = 0; j < SN; ++j)
            acc[i][j][e % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e],
                                                                       acc[i][j][e % NACC], 0, 0, 0);
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  // epilogue operands (bias, LN column sums, residual) are requested before the K loop so that
  // their latency is covered by it instead of extending the kernel's tail
  const int li = ln & 31, lk = ln >> 5;
  float e_bias[SM][SN], e_cs[SM][SN], e_res[SM][SN][16];
#pragma unroll
  for (int i = 0; i < SM; ++i)
#pragma unroll
    for (int j = 0; j < SN; ++j) {
      const int64_t col = min(n0 + wn * WN + j * 32 + li, g.N - 1);
      e_bias[i][j] = g.bias ? g.bias[col] : 0.f;
      e_cs[i][j] = LN ? g.ln_cs[col] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t row = min(m0 + wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk, g.M - 1);
        e_res[i][j][e] = g.residual ? g.residual[row * g.ldc + col] : 0.f;
      }
    }

  if constexpr (SWP) {
    static_assert(SM == 1 && SN == 1 && kBK == 32, "software pipelined loop: 64 x 64 x 32 tile");
    if (nfull > 0) {
      // operand sets: k groups 0-1 (X) and 2-3 (Y) of a tile; element q = 0, 1: A rows, 2, 3: B rows
      f32x4 xo[4], yo[4];
      const float* fa = s_gemm + (wm * WM + frow) * kPitch + fk;
      const float* fb = s_gemm + (TM + wn * WN + frow) * kPitch + fk;
      auto read1 = [&](f32x4 (&o)[4], int q, int buf, int koff) {  // one b128 operand fetch
        const float* p = ((q < 2) ? fa : fb) + buf * kBufFloats + koff + (q & 1) * 8;
        o[q] = *reinterpret_cast<const f32x4*>(p);
      };
      auto mfma1 = [&](const f32x4 (&o)[4], int i) {  // MFMA i of the 8 of an operand set
        const int q = i >> 2, e = i & 3;
        acc[0][0][e & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(o[q][e], o[2 + q][e],
                                                                acc[0][0][e & 1], 0, 0, 0);
      };
      auto store1 = [&](auto stage, int q, int buf, bool fresh) {  // one b128 of the staged tile
        constexpr int P = decltype(stage)::value;
        float* base = s_gemm + buf * kBufFloats;
        if (q < LA) {
          if (LN) ln_accumulate(q, ra[P][q], fresh);
          *reinterpret_cast<u32x4*>(base + (sr + kRPP * q) * kPitch + sc) = ra[P][q];
        } else {
          *reinterpret_cast<u32x4*>(base + (TM + sr + kRPP * (q - LA)) * kPitch + sc) = rb[P][q - LA];
        }
      };
      auto load1 = [&](auto stage, int q, int step) {  // one b128 global request of tile `step`
        constexpr int P = decltype(stage)::value;
        const int32_t soff = min(step, nfull - 1) * (kBK * 4);
        if (q < LA)
          ra[P][q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, va[q], soff, 0);
        else
          rb[P][q - LA] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, vb[q - LA], soff, 0);
      };
      static_assert(LA == 2 && LB == 2, "64 x 64 x 32 staging: 4 float4 per thread and tile");
      // One memory instruction behind every MFMA: its issue cycles (12-28 per LDS / buffer
      // instruction, measured in scripts/micro/mfma_lds.hip) then fall into the 64-cycle shadow of
      // that MFMA instead of between two MFMA groups.  sched_barrier(0) after every pair pins the
      // order (the scheduler otherwise clumps the memory instructions and lines up dependent MFMAs).
      // first half of a step: MFMAs on X; the next tile goes to LDS behind the first four (its
      // writes have landed by the barrier), Y of the same tile is fetched behind the last four
      // (order A0, B0, A1, B1: the first MFMA of the second half needs the first two only), and
      // the global requests of tile + 3 trail the writes by one slot
      auto half_a = [&](auto stage, int cur, int nxt, bool fresh, int step) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          mfma1(xo, i);
          if (i < 4) store1(stage, i, nxt, fresh);
          // the stage just written is refilled one slot behind its write (two tiles ahead of its
          // next use): requests as early as the registers allow
          if (i >= 1 && i <= 4) load1(stage, i - 1, step);
          if (i >= 4) read1(yo, ((i - 4) & 1) * 2 + ((i - 4) >> 1), cur, 16);
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      // workgroup barrier that only waits for the LDS WRITES of this wave: LDS operations return in
      // order, the 4 fetches issued behind the writes may stay in flight across the barrier (the
      // compiler's own wait insertion still guards their results)
      auto barrier_after_writes = [&]() {
        asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      };
      // second half: MFMAs on Y, fetch X of the next tile (behind the barrier, same A0, B0, A1, B1
      // order)
      auto half_b = [&](int nxt) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          mfma1(yo, i);
          if ((i & 1) == 0) read1(xo, ((i >> 1) & 1) * 2 + (i >> 2), nxt, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      };
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
      for (; s + 1 < nfull; s += 2) {
        half_a(S1{}, 0, 1, s + 1 < nfull, s + 3);  // tile s (buffer 0); tile s + 1: stage 1 -> buffer 1
        barrier_after_writes();
        half_b(1);
        half_a(S0{}, 1, 0, s + 2 < nfull, s + 4);  // tile s + 1 (buffer 1); tile s + 2: stage 0 -> buffer 0
        barrier_after_writes();
        half_b(0);
      }
      if (s < nfull) {  // last tile of an odd count: in buffer 0, X already fetched
#pragma unroll
        for (int q = 0; q < 4; ++q) read1(yo, q, 0, 16);
#pragma unroll
        for (int i = 0; i < 8; ++i) mfma1(xo, i);
#pragma unroll
        for (int i = 0; i < 8; ++i) mfma1(yo, i);
      }
      __syncthreads();  // the K remainder / LN statistics reuse the buffers
    }
  } else if (nfull > 0) {
    gload(S0{}, 0);
    gload(S1{}, 1);
    sstore(S0{}, 0);
    __syncthreads();
    // tile s lives in LDS buffer s & 1; register stage s & 1 is free once tile s is in LDS and is
    // refilled with tile s + 2 while the MFMAs of tile s run
    int s = 0;
    for (; s + 1 < nfull; s += 2) {
      gload(S0{}, s + 2);
      __builtin_amdgcn_sched_barrier(0);  // keep the requests AHEAD of the MFMAs (the scheduler
      compute(0);                         // otherwise sinks them behind the tile they should overlap)
      sstore(S1{}, 1);
      __syncthreads();
      gload(S1{}, s + 3);
      __builtin_amdgcn_sched_barrier(0);
      compute(1);
      sstore(S0{}, 0, s + 2 < nfull);
      __syncthreads();
    }
    if (s < nfull) compute(0);
  }
  if ((int64_t)nfull * kBK < g.K) {  // K remainder, not pipelined (odd feature sizes only)
    gload_tail(S0{});
    __syncthreads();  // every wave is done reading buffer 0
    sstore(S0{}, 0);
    __syncthreads();
    compute(0);
  }
  if (NACC == 2) {
#pragma unroll
    for (int i = 0; i < SM; ++i)
#pragma unroll
      for (int j = 0; j < SN; ++j) acc[i][j][0] += acc[i][j][1];
  }

  // LN: fold the row sums (kRowF4 staging lanes per row) and publish mean / rstd per tile row
  float* s_stat = s_gemm;  // [TM][2], the tile buffers are free now
  if (LN) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      float a = ln_s1[i].x + ln_s1[i].y, b = ln_s2[i].x + ln_s2[i].y;
#pragma unroll
      for (int o = 1; o < kRowF4; o <<= 1) {
        a += __shfl_xor(a, o, 64);
        b += __shfl_xor(b, o, 64);
      }
      if ((tid % kRowF4) == 0) {
        const float mean = a / (float)g.K;
        const float var = fmaxf(b / (float)g.K - mean * mean, 0.f);
        s_stat[(sr + kRPP * i) * 2 + 0] = mean;
        s_stat[(sr + kRPP * i) * 2 + 1] = 1.0f / sqrtf(var + g.ln_eps);
      }
    }
    __syncthreads();
  }

  // epilogue: C/D layout col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int i = 0; i < SM; ++i)
#pragma unroll
    for (int j = 0; j < SN; ++j) {
      const int64_t col = n0 + wn * WN + j * 32 + li;
      if (col >= g.N) continue;
      const float bv = e_bias[i][j];
      const float cs = e_cs[i][j];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int trow = wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
        const int64_t row = m0 + trow;
        if (row >= g.M) continue;
        float v = acc[i][j][0][e];
        if (LN) v = s_stat[trow * 2 + 1] * (v - s_stat[trow * 2] * cs);
        v += bv;
        if (g.act == 1) v = fmaxf(v, 0.f);
        if (g.act == 2) v = v / (1.0f + __expf(-v));
        if (g.act == 3) v = 1.0f / (1.0f + __expf(-v));
        if (g.act == 4) v = tanhf(v);
        if (g.act == 5) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));  // nn.GELU (exact)
        v = v * g.alpha + e_res[i][j][e];
        g.C[row * g.ldc + col] = v;
      }
    }
}

template <int TM, int TN, int kBK, int WPS, bool LN = false, bool SWP = false>
static int launch_gemm(GemmArgs g, hipStream_t st) {
  constexpr int kPitch = kBK + 4;
  const int64_t tiles_m = (g.M + TM - 1) / TM, tiles_n = (g.N + TN - 1) / TN;
  const int64_t total = tiles_m * tiles_n;
  if (total > 0x7fffffff || tiles_n > 0x7fffffff) return APS_ERR_INVALID;
  g.tiles_n = (int32_t)tiles_n;
  g.remap = (total % 8 == 0) ? 1 : 0;
  constexpr size_t lds = 2 * (size_t)(TM + TN) * kPitch * sizeof(float);
  static ApsPerDevice attr_set;  // > 64 KB of dynamic LDS needs the opt-in once per device
  if (lds > 64 * 1024 &&
      !aps_lds_opt_in(attr_set,
                      reinterpret_cast<const void*>(&gemm_f32_kernel<TM, TN, kBK, WPS, LN, SWP>),
                      (int)lds))
    return APS_ERR_LAUNCH;
  hipLaunchKernelGGL((gemm_f32_kernel<TM, TN, kBK, WPS, LN, SWP>), dim3((unsigned)total), dim3(256), lds,
                     st, g);
  return aps_launch_status();
}

// ------------------------------------------------------------------------------------------
// out[r, :] = LayerNorm(x[r, :] (+ res[r, :])) * gamma + beta     (one wavefront per row)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ res,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        float* __restrict__ out, int64_t rows,
                                                        int D, float eps) {
  const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + wv;
  if (row >= rows) return;
  const float* xr = x + row * D;
  const float* rr = res ? res + row * D : nullptr;
  float* orow = out + row * D;
  constexpr int MAXV = 16;  // register-resident up to D = 1024
  float v[MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int d = ln + 64 * i;
    float t = 0.f;
    if (d < D) {
      t = xr[d];
      if (rr) t += rr[d];
    }
    v[i] = t;
    s += t;
  }
  for (int d = ln + 64 * MAXV; d < D; d += 64) s += xr[d] + (rr ? rr[d] : 0.f);
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int d = ln + 64 * i;
    const float c = v[i] - mean;
    if (d < D) q += c * c;
  }
  for (int d = ln + 64 * MAXV; d < D; d += 64) {
    const float c = xr[d] + (rr ? rr[d] : 0.f) - mean;
    q += c * c;
  }
  const float inv = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int d = ln + 64 * i;
    if (d < D) orow[d] = (v[i] - mean) * inv * gamma[d] + beta[d];
  }
  for (int d = ln + 64 * MAXV; d < D; d += 64) {
    const float t = xr[d] + (rr ? rr[d] : 0.f);
    orow[d] = (t - mean) * inv * gamma[d] + beta[d];
  }
}

// ------------------------------------------------------------------------------------------
// out[n, t, d] = x[n, t, d] * factor + pe(t0 + t, d),  pe = interleaved (sin, cos)(pos * div[d/2])
// (InputSinPosEncoding, pose.py:93-118; the transpose to T x N x D is a view concern)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void posenc_kernel(const float* __restrict__ x,
                                                     const float* __restrict__ div,
                                                     float* __restrict__ out, int64_t total,
                                                     int64_t T, int D, float factor, int t0) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * 256) {
    const int d = (int)(i % D);
    const int64_t t = (i / D) % T;
    const float ang = (float)(t0 + t) * div[d >> 1];
    const float pe = (d & 1) ? cosf(ang) : sinf(ang);
    out[i] = x[i] * factor + pe;
  }
}

// token embedding + position encoding of the transformer decoder (decoder.py:150-153):
// out[n, t, :] = table[ids[n, t], :] * factor + pe(t0 + t, :); ids outside [0, V) raise the flag
__global__ __launch_bounds__(256) void embed_posenc_kernel(const float* __restrict__ table,
                                                           const int64_t* __restrict__ ids,
                                                           const float* __restrict__ div,
                                                           float* __restrict__ out, int64_t total,
                                                           int64_t T, int D, int64_t V,
                                                           float factor, int t0,
                                                           int32_t* __restrict__ bad) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * 256) {
    const int d = (int)(i % D);
    const int64_t row = i / D, t = row % T;
    const int64_t id = ids[row];
    float x = 0.f;
    if (id >= 0 && id < V) {
      x = table[id * D + d];
    } else if (bad != nullptr && d == 0) {
      atomicAdd(bad, 1);
    }
    const float ang = (float)(t0 + t) * div[d >> 1];
    out[i] = x * factor + ((d & 1) ? cosf(ang) : sinf(ang));
  }
}

// ------------------------------------------------------------------------------------------
// Scaled dot-product attention core:
//   ctx[i, :] = softmax_j((q_i . k_j [+ q_i . E[j - i + zero]]) / sqrt(dh) + pad_mask_j) V
// (impl.py:90-114; the bracketed term is RelMultiheadAttention's term_b, impl.py:258-292, whose
// digit_shift of q E^T (utils.py:14-39) is exactly the gather E[j - i + T - 1] done here in place).
// qkv: [N, T, 3, H, dh] (the fused in-projection output), ctx: [N, T, H, dh], E: [rel_len, dh].
// Workgroup = (head, utterance, tile of 16 queries); a wavefront owns 4 queries whose running
// max / sum / context live in registers.  Keys are streamed in blocks of KB through LDS (K and V
// of the head, plus the KB + 15 rows of E the 16 x KB (query, key) offsets touch) with the usual
// rescaling, so any sequence length works; lanes run along the keys for the scores and the softmax
// (wave reductions), then along dh for the context.
// (VALU form: ~2 % of the encoder's flops; the GEMMs carry the MFMA work.)
// ------------------------------------------------------------------------------------------
// Optional pieces shared by both attention kernels:
//   * rel table per head (head_stride = rel_len * dh) or shared by all heads (head_stride = 0)
//   * Transformer-XL biases (XlMultiheadAttention.dot_att, impl.py:322-343):
//       score = (q + u_h) . k_j + (q + v_h) . R_h[j - i + zero]
//     and its "query" is the VALUE projection (impl.py:366 passes `value` -- kept as is): qslot = 2
//   * context window of prep_context_mask (transformer/utils.py:60-98): key j is visible to query i
//     iff max((i / chunk - lctx) chunk, 0) <= j < (i / chunk + rctx + 1) chunk; lctx / rctx < 0: open
struct AttExtra {
  const float* rel_u;  // [H, dh] or null
  const float* rel_v;  // [H, dh] or null
  int64_t rel_head_stride;
  int32_t qslot;       // 0: query projection, 2: value projection
  int32_t chunk, lctx, rctx;
  // cross attention (streaming kernel only): keys / values of another sequence, kv [N, Tk, 2 H dh]
  // (k | v); null = self attention on the packed qkv rows
  const float* kv;
  int64_t Tk;
  // arbitrary additive mask [T, Tk] (0 / -inf or any bias) of the layers' `src_mask` argument
  // (streaming kernel only); null = none
  const float* add_mask;
};

__device__ __forceinline__ bool ctx_visible(const AttExtra& x, int64_t i, int64_t j) {
  const int64_t cf = i / x.chunk;
  if (x.rctx >= 0 && j >= (cf + x.rctx + 1) * x.chunk) return false;
  if (x.lctx >= 0 && j < (cf - x.lctx) * x.chunk) return false;
  return true;
}

constexpr int kAttQ = 4;             // queries per wavefront
constexpr int kAttQB = 4 * kAttQ;    // queries per workgroup

template <int DH, int KB, bool REL>
__global__ __launch_bounds__(256) void attention_core_kernel(const float* __restrict__ qkv,
                                                             const int64_t* __restrict__ lens,
                                                             const float* __restrict__ rel,
                                                             int64_t rel_zero, int64_t rel_len,
                                                             float* __restrict__ ctx, int64_t T,
                                                             int H, float scale, AttExtra ex) {
  constexpr int ER = REL ? KB + kAttQB - 1 : 1;
  __shared__ float s_k[KB][DH + 1];
  __shared__ float s_v[KB][DH + 1];
  __shared__ float s_e[ER][DH + 1];
  __shared__ float s_q[4][kAttQ][DH];                 // (q + u) / sqrt(dh): the key term
  __shared__ float s_q2[REL ? 4 : 1][kAttQ][DH];      // (q + v) / sqrt(dh): the relative term
  __shared__ float s_p[4][KB];
  const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
  const int h = blockIdx.x;
  const int64_t n = blockIdx.y;
  const int64_t qb = (int64_t)blockIdx.z * kAttQB;
  const int64_t q0 = qb + wv * kAttQ;
  // rows: self attention reads q | k | v from one [N, T, 3 H dh] tensor; cross attention reads
  // queries from [N, T, H dh] and keys | values from ex.kv [N, Tk, 2 H dh]
  const int64_t HD = (int64_t)H * DH;
  const int64_t Tk = ex.kv ? ex.Tk : T;
  const int64_t q_row = ex.kv ? HD : 3 * HD, k_row = ex.kv ? 2 * HD : 3 * HD;
  const float* qbase = qkv + n * T * q_row + (int64_t)h * DH + (ex.kv ? 0 : (int64_t)ex.qslot * HD);
  const float* kbase = ex.kv ? ex.kv + n * Tk * k_row + (int64_t)h * DH
                             : qkv + n * T * k_row + (int64_t)h * DH + HD;
  const float* vbase = kbase + HD;
  const int64_t len = lens ? min(Tk, max((int64_t)0, lens[n])) : Tk;
  constexpr int NV = (DH + 63) / 64;  // context elements per lane

  for (int qi = 0; qi < kAttQ; ++qi) {
    const int64_t i = q0 + qi;
    for (int d = ln; d < DH; d += 64) {
      const float q = (i < T) ? qbase[i * q_row + d] : 0.f;
      s_q[wv][qi][d] = (q + (ex.rel_u ? ex.rel_u[h * DH + d] : 0.f)) * scale;
      if (REL) s_q2[wv][qi][d] = (q + (ex.rel_v ? ex.rel_v[h * DH + d] : 0.f)) * scale;
    }
  }
  if (REL) rel += (int64_t)h * ex.rel_head_stride;
  float run_max[kAttQ], run_sum[kAttQ], acc[kAttQ][NV];
#pragma unroll
  for (int qi = 0; qi < kAttQ; ++qi) {
    run_max[qi] = -INFINITY;
    run_sum[qi] = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[qi][v] = 0.f;
  }

  for (int64_t j0 = 0; j0 < Tk; j0 += KB) {
    __syncthreads();  // previous key block fully consumed (and s_q visible)
    const int64_t nk = min((int64_t)KB, Tk - j0);
    for (int64_t e = tid; e < nk * DH; e += 256) {
      const int64_t j = e / DH;
      const int d = (int)(e % DH);
      s_k[j][d] = kbase[(j0 + j) * k_row + d];
      s_v[j][d] = vbase[(j0 + j) * k_row + d];
    }
    if (REL) {
      // window row w <-> table row j0 + w - (kAttQB - 1) - qb + rel_zero, i.e. offset (j - i) of
      // key j = j0 + jl and query i = qb + il sits at w = jl - il + kAttQB - 1
      const int64_t r0 = j0 - (kAttQB - 1) - qb + rel_zero;
      for (int e = tid; e < ER * DH; e += 256) {
        const int w = e / DH, d = e % DH;
        const int64_t r = r0 + w;
        s_e[w][d] = (r >= 0 && r < rel_len) ? rel[r * DH + d] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int qi = 0; qi < kAttQ; ++qi) {
      if (q0 + qi >= T) break;  // wave-uniform
      float sc[KB / 64];
      float blk_max = -INFINITY;
#pragma unroll
      for (int c = 0; c < KB / 64; ++c) {
        const int64_t j = ln + 64 * c;
        float dot = -INFINITY;
        if (j < nk && j0 + j < len && ctx_visible(ex, q0 + qi, j0 + j)) {
          dot = 0.f;
          if (REL) {
            const int w = (int)j - (wv * kAttQ + qi) + kAttQB - 1;
#pragma unroll 8
            for (int d = 0; d < DH; ++d)
              dot += s_q[wv][qi][d] * s_k[j][d] + s_q2[wv][qi][d] * s_e[w][d];
          } else {
#pragma unroll 8
            for (int d = 0; d < DH; ++d) dot += s_q[wv][qi][d] * s_k[j][d];
          }
          if (ex.add_mask) dot += ex.add_mask[(q0 + qi) * Tk + j0 + j];
        }
        sc[c] = dot;
        blk_max = fmaxf(blk_max, dot);
      }
      blk_max = wave_max(blk_max);
      const float new_max = fmaxf(run_max[qi], blk_max);
      // every key masked so far: leave the state untouched (exp(-inf - -inf) would be NaN)
      if (new_max > -INFINITY) {
        const float corr = (run_max[qi] > -INFINITY) ? expf(run_max[qi] - new_max) : 0.f;
        float psum = 0.f;
#pragma unroll
        for (int c = 0; c < KB / 64; ++c) {
          const float p = (sc[c] > -INFINITY) ? expf(sc[c] - new_max) : 0.f;
          s_p[wv][ln + 64 * c] = p;
          psum += p;
        }
        run_sum[qi] = run_sum[qi] * corr + wave_sum(psum);
        run_max[qi] = new_max;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const int d = ln + 64 * v;
          float a = acc[qi][v] * corr;
          if (d < DH) {
            for (int64_t j = 0; j < nk; ++j) a += s_p[wv][j] * s_v[j][d];
          }
          acc[qi][v] = a;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
  }
#pragma unroll
  for (int qi = 0; qi < kAttQ; ++qi) {
    const int64_t i = q0 + qi;
    if (i >= T) break;
    // a fully padded sequence (len = 0) is softmax over -inf only -> NaN in torch; zeros here
    const float inv = run_sum[qi] > 0.f ? 1.0f / run_sum[qi] : 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int d = ln + 64 * v;
      if (d < DH) ctx[(n * T + i) * (int64_t)H * DH + (int64_t)h * DH + d] = acc[qi][v] * inv;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Short-sequence attention core on fp32 MFMA: T <= 64, head_dim 64 (the joint front end: 249
// STFT frames -> 63 encoder frames).  One workgroup per (utterance, head) holds Q, K, V^T and the
// 2T-1 window of the relative table in LDS (120 KB) and does
//   S = Q K^T (64 x 64 x 64),  P = Q E^T (64 x 128 x 64),  S[i][j] += P[i][j - i + 63]
//   (the reference's digit_shift gather, done on the LDS copy of P), masked row softmax,
//   O = softmax(S) V (64 x 64 x 64)
// as 32 x 32 x 2 MFMA tiles with the same b128 operand fetch as the GEMM (rows K-contiguous in LDS,
// pitch 68).  ~8 us per launch against ~40 us for the streaming VALU kernel at this size.
// ------------------------------------------------------------------------------------------
constexpr int kSmallT = 64, kSmallPitch = 68;

// KT = 64: the whole sequence in one tile;
// KT = 128: 64 < T <= 128 (BASELINE config 4: 100 encoder frames), one workgroup per 64-query tile
// against all 128 keys.  Relative / XL terms in both: the table window of a 64-query tile spans the
// offsets j - i of its 64 x KT (query, key) pairs, KT + 63 rows (held as KT + 64).
template <int KT, bool REL>
__global__ __launch_bounds__(256) void attention_small_kernel(const float* __restrict__ qkv,
                                                              const int64_t* __restrict__ lens,
                                                              const float* __restrict__ rel,
                                                              int64_t rel_zero, int64_t rel_len,
                                                              float* __restrict__ ctx, int64_t T,
                                                              int H, float scale, AttExtra ex) {
  constexpr int DH = 64, PT = kSmallPitch, VP = KT + 4, CT = KT / 64;
  constexpr int WIN = KT + 64, PP = WIN + 1, PTL = WIN / 64;  // window rows, pitch of P, P tiles per wave
  // KT = 128 keeps V in registers until Q is dead and stages V^T 64 keys at a time into Q's region:
  // 52 KB of LDS instead of 86 KB, so three workgroups share a CU and one's staging / softmax
  // phases hide behind another's MFMAs
  // The relative form keeps V in registers too and lets the shifted scores P overwrite the table
  // window E they were computed from: 70 KB of LDS instead of 137 KB (87 KB with the Transformer-XL
  // biases, whose second query copy needs its own 17 KB), so two workgroups share a CU.
  constexpr bool LATE_V = (KT == 128) || REL;
  constexpr int VH = KT / 64;  // 64-key halves of V held in registers (LATE_V)
  extern __shared__ __attribute__((aligned(16))) float s_att[];
  float* s_q = s_att;                 // [64][68]  (q + u) / sqrt(dh)   (LATE_V: later V^T halves)
  float* s_k = s_q + 64 * PT;         // [KT][68]  (later: scores / probabilities [64][KT + 4])
  float* s_vt = s_k + KT * PT;        // [64 d][KT + 4]   (!LATE_V)
  float* s_e = LATE_V ? s_vt : s_vt + 64 * VP;  // [WIN][68]   (REL)
  float* s_p = s_e;                   // [64][WIN + 1]   (REL): E is dead when P is written
  const bool two_q = REL && (ex.rel_u != nullptr || ex.rel_v != nullptr);
  float* s_q2 = two_q ? s_e + WIN * PT : s_q;  // [64][68]  (XL: (q + v) / sqrt(dh))
  const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
  const int wm = wv >> 1, wn = wv & 1;
  const int h = blockIdx.x;
  const int64_t n = blockIdx.y;
  const int q0 = blockIdx.z * 64;
  const int64_t D3 = (int64_t)3 * H * DH;
  const float* base = qkv + n * T * D3 + (int64_t)h * DH;
  const int len = (int)(lens ? min(T, max((int64_t)0, lens[n])) : T);

  // ---- stage Q tile (scaled), K, V^T, E window: float4 global loads, rows beyond T are zero
  for (int e = tid; e < 64 * 16; e += 256) {
    const int r = e >> 4, c4 = (e & 15) * 4;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q0 + r < T)
      q = *reinterpret_cast<const float4*>(base + (int64_t)(q0 + r) * D3 +
                                           (int64_t)ex.qslot * H * DH + c4);
    float4 qa = q, qb = q;
    if (ex.rel_u) {
      const float4 u = *reinterpret_cast<const float4*>(ex.rel_u + h * DH + c4);
      qa = make_float4(q.x + u.x, q.y + u.y, q.z + u.z, q.w + u.w);
    }
    if (ex.rel_v) {
      const float4 u = *reinterpret_cast<const float4*>(ex.rel_v + h * DH + c4);
      qb = make_float4(q.x + u.x, q.y + u.y, q.z + u.z, q.w + u.w);
    }
    qa.x *= scale, qa.y *= scale, qa.z *= scale, qa.w *= scale;
    qb.x *= scale, qb.y *= scale, qb.z *= scale, qb.w *= scale;
    *reinterpret_cast<float4*>(s_q + r * PT + c4) = qa;
    if (REL && two_q) *reinterpret_cast<float4*>(s_q2 + r * PT + c4) = qb;
  }
  // LATE_V: V fragment (it, half): key 64 half + 16 it + (ln >> 2), columns 16 wv + 4 (ln & 3) ..
  // (a wave's 64 lanes write 4 x 16 distinct LDS banks when the fragment goes down transposed)
  float4 vreg[LATE_V ? VH : 1][4];
  if constexpr (LATE_V) {
    float4 kreg[KT * 16 / 256];
#pragma unroll
    for (int it = 0; it < KT * 16 / 256; ++it) {
      const int e = tid + 256 * it, r = e >> 4, c4 = (e & 15) * 4;
      kreg[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < T) kreg[it] = *reinterpret_cast<const float4*>(base + (int64_t)r * D3 + c4 + (int64_t)H * DH);
    }
#pragma unroll
    for (int hf = 0; hf < VH; ++hf)
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int r = 64 * hf + 16 * it + (ln >> 2), c4 = 16 * wv + 4 * (ln & 3);
        vreg[hf][it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < T)
          vreg[hf][it] =
              *reinterpret_cast<const float4*>(base + (int64_t)r * D3 + c4 + (int64_t)2 * H * DH);
      }
#pragma unroll
    for (int it = 0; it < KT * 16 / 256; ++it) {
      const int e = tid + 256 * it, r = e >> 4, c4 = (e & 15) * 4;
      *reinterpret_cast<float4*>(s_k + r * PT + c4) = kreg[it];
    }
  } else {
    for (int e = tid; e < KT * 16; e += 256) {
      const int r = e >> 4, c4 = (e & 15) * 4;
      float4 k = make_float4(0.f, 0.f, 0.f, 0.f), v = k;
      if (r < T) {
        const float* p = base + (int64_t)r * D3 + c4;
        k = *reinterpret_cast<const float4*>(p + (int64_t)H * DH);
        v = *reinterpret_cast<const float4*>(p + (int64_t)2 * H * DH);
      }
      *reinterpret_cast<float4*>(s_k + r * PT + c4) = k;
      s_vt[(c4 + 0) * VP + r] = v.x;
      s_vt[(c4 + 1) * VP + r] = v.y;
      s_vt[(c4 + 2) * VP + r] = v.z;
      s_vt[(c4 + 3) * VP + r] = v.w;
    }
  }
  if (REL) {
    rel += (int64_t)h * ex.rel_head_stride;
    // window row w <-> offset j - (q0 + i) = w - 63 - q0 <-> table row w - 63 - q0 + rel_zero
    for (int e = tid; e < WIN * 16; e += 256) {
      const int w = e >> 4, c4 = (e & 15) * 4;
      const int64_t r = (int64_t)w - 63 - q0 + rel_zero;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r >= 0 && r < rel_len) v = *reinterpret_cast<const float4*>(rel + r * DH + c4);
      *reinterpret_cast<float4*>(s_e + w * PT + c4) = v;
    }
  }
  __syncthreads();

  const int frow = ln & 31, fk = (ln >> 5) * 4;
  // C[32 x 32] += A[32 rows][k] . B[32 rows][k]^T over k = 0 .. 8 KG - 1 (both K-contiguous)
  auto tile = [&](const float* A, int pa_, const float* B, int pb_, int KG, f32x16& acc) {
    const float* pa = A + frow * pa_ + fk;
    const float* pb = B + frow * pb_ + fk;
    for (int kg = 0; kg < KG; ++kg) {
      const float4 a = *reinterpret_cast<const float4*>(pa + kg * 8);
      const float4 b = *reinterpret_cast<const float4*>(pb + kg * 8);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
    }
  };
  f32x16 sacc[CT], pacc[REL ? PTL : 1];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
#pragma unroll
    for (int c = 0; c < CT; ++c) sacc[c][e] = 0.f;
#pragma unroll
    for (int t = 0; t < (REL ? PTL : 1); ++t) pacc[t][e] = 0.f;
  }
  // wave (wm, wn): query rows 32 wm .., key columns CT x 32 starting at 32 CT wn
#pragma unroll
  for (int c = 0; c < CT; ++c)
    tile(s_q + wm * 32 * PT, PT, s_k + (wn * 32 * CT + c * 32) * PT, PT, 8, sacc[c]);
  if constexpr (REL) {
    // wave (wm, wn): window columns (WIN / 2) wn .. in PTL tiles of 32
#pragma unroll
    for (int t = 0; t < PTL; ++t)
      tile(s_q2 + wm * 32 * PT, PT, s_e + (wn * (WIN / 2) + t * 32) * PT, PT, 8, pacc[t]);
    __syncthreads();  // every wave is done with E: P goes into its place
    // accumulator layout: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int t = 0; t < PTL; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int i = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * (ln >> 5);
        s_p[i * PP + wn * (WIN / 2) + t * 32 + (ln & 31)] = pacc[t][e];
      }
  }
  __syncthreads();  // K no longer needed: its region becomes the score matrix [64][VP]
  auto put_v_half = [&](int hf) {  // V^T of keys 64 hf .. 64 hf + 63 -> s_q as [64 d][68]
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int r = 16 * it + (ln >> 2), c4 = 16 * wv + 4 * (ln & 3);
      const float4 v = vreg[LATE_V ? hf : 0][it];
      s_q[(c4 + 0) * PT + r] = v.x;
      s_q[(c4 + 1) * PT + r] = v.y;
      s_q[(c4 + 2) * PT + r] = v.z;
      s_q[(c4 + 3) * PT + r] = v.w;
    }
  };
  if constexpr (LATE_V) put_v_half(0);  // Q is dead as well
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int i = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * (ln >> 5);
      const int j = wn * 32 * CT + c * 32 + (ln & 31);
      float v = sacc[c][e];
      if (REL) v += s_p[i * PP + j - i + 63];
      s_k[i * VP + j] = (j < len && ctx_visible(ex, q0 + i, j)) ? v : -INFINITY;
    }
  __syncthreads();
  // ---- row softmax: wave w owns rows 16 w .. 16 w + 15, lanes = keys (CT per lane)
#pragma unroll 4
  for (int r = 0; r < 16; ++r) {
    const int i = wv * 16 + r;
    float v[CT], m = -INFINITY;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      v[c] = s_k[i * VP + ln + 64 * c];
      m = fmaxf(m, v[c]);
    }
    m = wave_max(m);
    // a fully padded sequence (len = 0) is softmax over -inf only -> NaN in torch; zeros here
    float p[CT], sum = 0.f;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      p[c] = (m > -INFINITY) ? __expf(v[c] - m) : 0.f;
      sum += p[c];
    }
    sum = wave_sum(sum);
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;
#pragma unroll
    for (int c = 0; c < CT; ++c) s_k[i * VP + ln + 64 * c] = p[c] * inv;
  }
  __syncthreads();
  f32x16 oacc;
#pragma unroll
  for (int e = 0; e < 16; ++e) oacc[e] = 0.f;
  if constexpr (LATE_V) {
    tile(s_k + wm * 32 * VP, VP, s_q + wn * 32 * PT, PT, 8, oacc);
    if constexpr (VH == 2) {
      __syncthreads();
      put_v_half(1);
      __syncthreads();
      tile(s_k + wm * 32 * VP + 64, VP, s_q + wn * 32 * PT, PT, 8, oacc);
    }
  } else {
    tile(s_k + wm * 32 * VP, VP, s_vt + wn * 32 * VP, VP, KT / 8, oacc);
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int i = q0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * (ln >> 5);
    const int d = wn * 32 + (ln & 31);
    if (i < T) ctx[(n * T + i) * (int64_t)H * DH + (int64_t)h * DH + d] = oacc[e];
  }
}

// ------------------------------------------------------------------------------------------
// Conformer convolution module between its two pointwise GEMMs (impl.py:478-489):
//   out[n, t, d] = swish(bn(sum_k w[d, k] * glu(x)[n, t + k - pad, d] + b[d]))
//   glu(x)[n, t, d] = x[n, t, d] * sigmoid(x[n, t, D + d]),  zero outside [0, T)
// x: [N, T, 2D] (output of the D -> 2D pointwise GEMM), out: [N, T, D]; bn is the eval-mode
// affine (scale, shift) folded on the host.  A workgroup owns 64 channels x 16 frames: its 4 waves
// stage the gated values of the 16 + K - 1 frames in LDS (each wave every 4th frame, the loads of
// a wave's frames issued together), then wave g slides the K taps over frames 4 g .. 4 g + 3.
// HBM: reads 2D (1 + (K-1)/16) (the halo re-reads hit L2), writes D.
// ------------------------------------------------------------------------------------------
constexpr int kConvTT = 16, kConvCh = 64, kConvMaxRows = kConvTT + 62;

__global__ __launch_bounds__(256) void glu_dwconv_kernel(const float* __restrict__ x,
                                                         const float* __restrict__ w,
                                                         const float* __restrict__ bias,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ shift,
                                                         float* __restrict__ out, int64_t T, int D,
                                                         int K, int swish, int causal,
                                                         const float* __restrict__ pad_bias) {
  __shared__ float s_g[kConvMaxRows][kConvCh];
  const int tid = threadIdx.x, c = tid & 63, rg = tid >> 6;
  const int d = blockIdx.x * kConvCh + c;
  const int dc = min(d, D - 1);  // clamped: lanes past D compute on valid data, store nothing
  const int64_t t0 = (int64_t)blockIdx.y * kConvTT;
  const int64_t n = blockIdx.z;
  // causal (casual_conv1d, impl.py:468-505): all K - 1 context frames on the left; the reference
  // pads the module INPUT, so a padded frame carries glu(pointwise bias), not zero
  const int pad = causal ? K - 1 : (K - 1) / 2;
  const int rows = kConvTT + K - 1;
  const float* xn = x + n * T * 2 * D;
  const float fill = (causal && pad_bias)
                         ? pad_bias[dc] * __builtin_amdgcn_rcpf(1.0f + __expf(-pad_bias[D + dc]))
                         : 0.f;
  // ---- stage: wave rg takes rows rg, rg + 4, ...; 8 independent row loads in flight per pass
  for (int r0 = rg; r0 < rows; r0 += 32) {
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int64_t t = min(max(t0 + r0 + 4 * i - pad, (int64_t)0), T - 1);
      a[i] = xn[t * 2 * D + dc];
      b[i] = xn[t * 2 * D + D + dc];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = r0 + 4 * i;
      const int64_t t = t0 + r - pad;
      if (r < rows)
        s_g[r][c] = (t >= 0 && t < T) ? a[i] * __builtin_amdgcn_rcpf(1.0f + __expf(-b[i]))
                                      : (t < 0 ? fill : 0.f);
    }
  }
  const float bv = bias ? bias[dc] : 0.f, sc = scale ? scale[dc] : 1.f, sh = shift ? shift[dc] : 0.f;
  // taps in registers (the first 32; K <= 63 keeps a rolled remainder)
  const float* wd = w + (int64_t)dc * K;
  float wr[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) wr[k] = (k < K) ? wd[k] : 0.f;
  __syncthreads();
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const int r = rg * 4 + f;
    const int64_t t = t0 + r;
    float acc = bv;
#pragma unroll
    for (int k = 0; k < 32; ++k)
      if (k < K) acc += wr[k] * s_g[r + k][c];
    for (int k = 32; k < K; ++k) acc += wd[k] * s_g[r + k][c];
    acc = acc * sc + sh;
    // activation after the BatchNorm affine: 1 swish, 2 relu, 3 gelu (erf form), 0 none
    if (swish == 1) acc = acc * __builtin_amdgcn_rcpf(1.0f + __expf(-acc));
    else if (swish == 2) acc = fmaxf(acc, 0.f);
    else if (swish == 3) acc = 0.5f * acc * (1.0f + erff(acc * 0.70710678118654752f));
    if (t < T && d < D) out[(n * T + t) * D + d] = acc;
  }
}

}  // namespace aps

using namespace aps;

static int run_linear(const float* A, const float* W, const float* bias, const float* residual,
                      float* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
                      int64_t ldc, int32_t act, float alpha, const float* ln_cs, float ln_eps,
                      void* stream) {
  APS_CHECK_ARG(A && W && C && M > 0 && N > 0 && K > 0);
  APS_CHECK_ARG(lda >= K && ldw >= K && ldc >= N);
  // 16-byte aligned row starts for the float4 tile loads
  APS_CHECK_ARG(lda % 4 == 0 && ldw % 4 == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0);
  APS_CHECK_ARG(act >= 0 && act <= 5);
  // 32-bit buffer offsets: operands up to 4 GB (2 GB for the signed per-thread part)
  if (M * lda * 4 >= ((int64_t)1 << 31) || N * ldw * 4 >= ((int64_t)1 << 31)) return APS_ERR_UNSUPPORTED;
  GemmArgs g{A, W, bias, residual, C, M, N, K, lda, ldw, ldc, act, alpha, 0, 0, ln_cs, ln_eps};
  hipStream_t st = static_cast<hipStream_t>(stream);
  // hand-scheduled loop for the latency-bound shapes.  Measured in situ on one box: joint step
  // 6 940 -> 7 150 utt/s with it on the M = 2016 GEMMs only (7 110 when the M = 7968 mask-net
  // GEMMs take it too); encoder workload (M = 12800, 6-25 tiles per CU) 9 050 -> 8 530 -> by M
  // (r02, batches of 128 utterances, M = 8064: 512 x 512 93 -> 104 TF, 512 x 1024 112 -> 117 TF with
  // the hand-scheduled loop, scripts/gemm_variants.py; at M = 12800 -- the encoder workload -- the
  // compiler-scheduled loop stays ahead, 9 760 against 9 150 utt/s, so the switch sits between)
  const bool swp = M <= 10240;
  // Measured and removed in round 3: a persistent form whose K-loop prefetch ran across
  // output tiles (168 VGPRs, two workgroups per CU: 43.1 against 41.1 us at 8064 x 512 x 512) and the
  // 128 x 128 / 128 x 64 tiles (the 64 x 64 tile wins at every shape of this path, 2016 x 512 x 512
  // 61 against 21 TF up to 4096^3 129 against 107 TF, scripts/gemm_sweep.py).
  if (ln_cs) return swp ? launch_gemm<64, 64, 32, 3, true, true>(g, st)
                        : launch_gemm<64, 64, 32, 3, true>(g, st);
  return swp ? launch_gemm<64, 64, 32, 3, false, true>(g, st) : launch_gemm<64, 64, 32, 3>(g, st);
}

extern "C" int aps_linear(const float* A, const float* W, const float* bias, const float* residual,
                          float* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
                          int64_t ldc, int32_t act, float alpha, void* stream) {
  return run_linear(A, W, bias, residual, C, M, N, K, lda, ldw, ldc, act, alpha, nullptr, 0.f,
                    stream);
}

extern "C" int aps_linear_layernorm(const float* A, const float* W_gamma, const float* bias_beta,
                                    const float* colsum, const float* residual, float* C,
                                    int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
                                    int64_t ldc, int32_t act, float alpha, float eps,
                                    void* stream) {
  APS_CHECK_ARG(colsum != nullptr);
  return run_linear(A, W_gamma, bias_beta, residual, C, M, N, K, lda, ldw, ldc, act, alpha, colsum,
                    eps, stream);
}

extern "C" int aps_layernorm(const float* x, const float* residual, const float* gamma,
                             const float* beta, float* out, int64_t rows, int64_t D, float eps,
                             void* stream) {
  APS_CHECK_ARG(x && gamma && beta && out && rows > 0 && D > 0 && D < (1 << 30));
  hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, residual, gamma, beta, out, rows, (int)D,
                     eps);
  return aps_launch_status();
}

extern "C" int aps_posenc_add(const float* x, const float* div_term, float* out, int64_t N,
                              int64_t T, int64_t D, float factor, int32_t t0, void* stream) {
  APS_CHECK_ARG(x && div_term && out && N > 0 && T > 0 && D > 0 && D % 2 == 0);
  const int64_t total = N * T * D;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(posenc_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, div_term, out, total, T, (int)D, factor,
                     (int)t0);
  return aps_launch_status();
}

extern "C" int aps_embedding_posenc(const float* table, const int64_t* ids, const float* div_term,
                                    float* out, int64_t N, int64_t T, int64_t D, int64_t V,
                                    float factor, int32_t t0, int32_t* bad_count, void* stream) {
  APS_CHECK_ARG(table && ids && div_term && out && N > 0 && T > 0 && D > 0 && D % 2 == 0 && V > 0);
  const int64_t total = N * T * D;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(embed_posenc_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     static_cast<hipStream_t>(stream), table, ids, div_term, out, total, T, (int)D,
                     V, factor, (int)t0, bad_count);
  return aps_launch_status();
}

template <int DH, int KB, bool REL>
static void launch_attention(dim3 grid, hipStream_t st, const float* qkv, const int64_t* lens,
                             const float* rel, int64_t rel_zero, int64_t rel_len, float* ctx,
                             int64_t T, int H, float scale, const AttExtra& ex) {
  hipLaunchKernelGGL((attention_core_kernel<DH, KB, REL>), grid, dim3(256), 0, st, qkv, lens, rel,
                     rel_zero, rel_len, ctx, T, H, scale, ex);
}

extern "C" int aps_attention_core(const float* qkv, const int64_t* lens, const float* rel,
                                  int64_t rel_zero, int64_t rel_len, int64_t rel_head_stride,
                                  const float* rel_u, const float* rel_v, int32_t query_slot,
                                  int32_t chunk, int32_t lctx, int32_t rctx, const float* add_mask,
                                  float* ctx, int64_t N, int64_t T, int64_t H, int64_t head_dim,
                                  void* stream) {
  APS_CHECK_ARG(qkv && ctx && N > 0 && N <= 65535 && T > 0 && H > 0 && H <= 65535);
  APS_CHECK_ARG(!rel || (rel_len > 0 && rel_zero >= 0 && rel_zero < rel_len));
  APS_CHECK_ARG(rel || (!rel_u && !rel_v));
  APS_CHECK_ARG((query_slot == 0 || query_slot == 2) && chunk >= 1 && rel_head_stride >= 0);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const float scale = 1.0f / sqrtf((float)head_dim);
  const AttExtra ex{rel_u, rel_v, rel_head_stride, query_slot, chunk, lctx, rctx, nullptr, 0, add_mask};
  if (head_dim == 64 && !add_mask && !getenv("APS_ATT_GENERIC") && T <= 128) {
    // once per device (not legal inside a stream capture: the first call must be an eager one)
    static ApsPerDevice attr_rel, attr_abs, attr_rel2;
    if (!aps_lds_opt_in(attr_rel, reinterpret_cast<const void*>(&attention_small_kernel<64, true>),
                        160 * 1024) ||
        !aps_lds_opt_in(attr_abs, reinterpret_cast<const void*>(&attention_small_kernel<128, false>),
                        160 * 1024) ||
        !aps_lds_opt_in(attr_rel2, reinterpret_cast<const void*>(&attention_small_kernel<128, true>),
                        160 * 1024))
      return APS_ERR_LAUNCH;
    if (T <= kSmallT) {
      // relative form: Q | K (later the scores) | E (later P) [| second query copy of the XL form]
      const size_t lds = (rel ? (size_t)(2 * 64 * kSmallPitch + 128 * kSmallPitch +
                                         ((rel_u || rel_v) ? 64 * kSmallPitch : 0))
                              : (size_t)(3 * 64 * kSmallPitch)) * sizeof(float);
      dim3 g2((unsigned)H, (unsigned)N, 1);
      if (rel)
        hipLaunchKernelGGL((attention_small_kernel<64, true>), g2, dim3(256), lds, st, qkv, lens, rel,
                           rel_zero, rel_len, ctx, T, (int)H, scale, ex);
      else
        hipLaunchKernelGGL((attention_small_kernel<64, false>), g2, dim3(256), lds, st, qkv, lens,
                           rel, rel_zero, rel_len, ctx, T, (int)H, scale, ex);
    } else {
      // Q | K (later the scores) [| the 192-row table window (later P) | second query copy (XL)]
      const size_t lds = (size_t)(64 * kSmallPitch + 128 * kSmallPitch +
                                  (rel ? 192 * kSmallPitch + ((rel_u || rel_v) ? 64 * kSmallPitch : 0)
                                       : 0)) * sizeof(float);
      dim3 g2((unsigned)H, (unsigned)N, (unsigned)((T + 63) / 64));
      if (rel)
        hipLaunchKernelGGL((attention_small_kernel<128, true>), g2, dim3(256), lds, st, qkv, lens, rel,
                           rel_zero, rel_len, ctx, T, (int)H, scale, ex);
      else
        hipLaunchKernelGGL((attention_small_kernel<128, false>), g2, dim3(256), lds, st, qkv, lens,
                           rel, rel_zero, rel_len, ctx, T, (int)H, scale, ex);
    }
    return aps_launch_status();
  }
  dim3 grid((unsigned)H, (unsigned)N, (unsigned)((T + kAttQB - 1) / kAttQB));
#define APS_ATT_CASE(DH)                                                                        \
  case DH:                                                                                      \
    if (rel)                                                                                    \
      launch_attention<DH, 64, true>(grid, st, qkv, lens, rel, rel_zero, rel_len, ctx, T, (int)H, \
                                     scale, ex);                                                \
    else                                                                                        \
      launch_attention<DH, 128, false>(grid, st, qkv, lens, nullptr, 0, 0, ctx, T, (int)H, scale, \
                                       ex);                                                     \
    break;
  switch (head_dim) {
    APS_ATT_CASE(32)
    APS_ATT_CASE(64)
    APS_ATT_CASE(128)
    default:
      return APS_ERR_UNSUPPORTED;
  }
#undef APS_ATT_CASE
  return aps_launch_status();
}

// Cross attention softmax(q k^T / sqrt(dh) + memory_mask + key padding) v of the transformer
// decoder (aps/asr/transformer/decoder.py:78-86): queries of one sequence against the keys / values
// of another, on the streaming kernel.  add_mask: the layer's memory_mask as an additive [Tq, Tk]
// matrix (0 / -inf or any bias), or NULL.
extern "C" int aps_attention_cross(const float* q, const float* kv, const int64_t* key_lens,
                                   const float* add_mask, float* ctx, int64_t N, int64_t Tq,
                                   int64_t Tk, int64_t H, int64_t head_dim, void* stream) {
  APS_CHECK_ARG(q && kv && ctx && N > 0 && N <= 65535 && Tq > 0 && Tk > 0 && H > 0 && H <= 65535);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const float scale = 1.0f / sqrtf((float)head_dim);
  const AttExtra ex{nullptr, nullptr, 0, 0, 1, -1, -1, kv, Tk, add_mask};
  dim3 grid((unsigned)H, (unsigned)N, (unsigned)((Tq + kAttQB - 1) / kAttQB));
  switch (head_dim) {
    case 32: launch_attention<32, 128, false>(grid, st, q, key_lens, nullptr, 0, 0, ctx, Tq, (int)H, scale, ex); break;
    case 64: launch_attention<64, 128, false>(grid, st, q, key_lens, nullptr, 0, 0, ctx, Tq, (int)H, scale, ex); break;
    case 128: launch_attention<128, 128, false>(grid, st, q, key_lens, nullptr, 0, 0, ctx, Tq, (int)H, scale, ex); break;
    default: return APS_ERR_UNSUPPORTED;
  }
  return aps_launch_status();
}

extern "C" int aps_glu_dwconv(const float* x, const float* weight, const float* bias,
                              const float* scale, const float* shift, float* out, int64_t N,
                              int64_t T, int64_t D, int64_t K, int32_t swish, int32_t causal,
                              const float* pad_bias, void* stream) {
  APS_CHECK_ARG(x && weight && out && N > 0 && N <= 65535 && T > 0 && D > 0 && D < (1 << 30));
  APS_CHECK_ARG(K > 0 && K % 2 == 1 && K <= 63);
  dim3 grid((unsigned)((D + kConvCh - 1) / kConvCh), (unsigned)((T + kConvTT - 1) / kConvTT),
            (unsigned)N);
  APS_CHECK_ARG(grid.y <= 65535);
  hipLaunchKernelGGL(glu_dwconv_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), x,
                     weight, bias, scale, shift, out, T, (int)D, (int)K, (int)swish, (int)causal,
                     pad_bias);
  return aps_launch_status();
}
