// Backward launches (section 8(f) row 1): one GPU thread per index functor of grad_core.h, the
// tiled transpose that feeds the backward GEMMs, and the reverse-time sweep of the LSTM.
//
//   linear backward   g_pre = g_y alpha act'(pre)            (act_backward)
//                     g_x = g_pre W        = aps_linear(g_pre, W^T)        W^T by aps_transpose
//                     g_W = g_pre^T x      = aps_linear(g_pre^T, x^T)      both by aps_transpose
//                     g_b = colsum(g_pre)                     (colreduce)
//   LSTM backward     gates / cells recomputed from the saved layer outputs (one batched GEMM + one
//                     scan), then T reverse steps of [g_h = g_pre_{t+1} W_hh: aps_linear] + [gate
//                     adjoint: lstm_backward_step], then the weight gradients as batched GEMMs.
// Everything the backward contracts runs on the forward's fp32 MFMA GEMM (nn.hip, aps_linear).
#include <stdlib.h>

#include "common.h"
#include "grad_core.h"

namespace aps {

template <class Op>
__global__ __launch_bounds__(256) void each_kernel(Op op, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) op(i);
}

template <class Op>
static int launch_each(const Op& op, int64_t n, void* stream) {
  if (n <= 0) return APS_OK;
  const int64_t blocks = (n + 255) / 256;
  if (blocks > 0x7fffffff) return APS_ERR_INVALID;
  hipLaunchKernelGGL((each_kernel<Op>), dim3((unsigned)blocks), dim3(256), 0,
                     static_cast<hipStream_t>(stream), op, n);
  return aps_launch_status();
}

// out[c, r] = in[r, c]: 32 x 32 tiles through LDS (pitch 33: conflict-free columns), both sides
// coalesced
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in,
                                                        float* __restrict__ out, int64_t rows,
                                                        int64_t cols, int64_t ld_in, int64_t ld_out) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t r = r0 + ty + 8 * k, c = c0 + tx;
    tile[ty + 8 * k][tx] = (r < rows && c < cols) ? in[r * ld_in + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t c = c0 + ty + 8 * k, r = r0 + tx;
    if (c < cols && r < rows) out[c * ld_out + r] = tile[tx][ty + 8 * k];
  }
}

// LayerNorm backward, one wavefront per row (grad_core.h: LayerNormBackward is the reference form)
__global__ __launch_bounds__(256) void layernorm_backward_kernel(
    const float* __restrict__ x, const float* __restrict__ residual, const float* __restrict__ gamma,
    const float* __restrict__ g_y, float* __restrict__ g_x, float* __restrict__ t, int64_t rows,
    int64_t D, float eps) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int ln = threadIdx.x & 63;
  if (r >= rows) return;
  const float* xr = x + r * D;
  const float* rr = residual ? residual + r * D : nullptr;
  const float* gr = g_y + r * D;
  float s = 0.f;
  for (int64_t d = ln; d < D; d += 64) s += xr[d] + (rr ? rr[d] : 0.f);
  const float mu = wave_sum(s) / (float)D;
  float v = 0.f;
  for (int64_t d = ln; d < D; d += 64) {
    const float c = xr[d] + (rr ? rr[d] : 0.f) - mu;
    v += c * c;
  }
  const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)D + eps);
  float m1 = 0.f, m2 = 0.f;
  for (int64_t d = ln; d < D; d += 64) {
    const float xh = (xr[d] + (rr ? rr[d] : 0.f) - mu) * rstd;
    const float gh = gr[d] * (gamma ? gamma[d] : 1.f);
    m1 += gh;
    m2 += gh * xh;
  }
  m1 = wave_sum(m1) / (float)D;
  m2 = wave_sum(m2) / (float)D;
  for (int64_t d = ln; d < D; d += 64) {
    const float xh = (xr[d] + (rr ? rr[d] : 0.f) - mu) * rstd;
    const float gh = gr[d] * (gamma ? gamma[d] : 1.f);
    g_x[r * D + d] = rstd * (gh - m1 - xh * m2);
    if (t) t[r * D + d] = gr[d] * xh;
  }
}

// the same for a few very long rows (the whole-utterance GroupNorm(1, D) of the linear / conv1d
// projections, component.py:85-114: one row = T x D values): a 1024-thread workgroup per row, the four
// row sums through LDS
__device__ __forceinline__ float layernorm_block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();  // (red is reused from the previous sum)
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) s += red[w];
  return s;
}
__global__ __launch_bounds__(1024) void layernorm_backward_wide_kernel(
    const float* __restrict__ x, const float* __restrict__ residual, const float* __restrict__ gamma,
    const float* __restrict__ g_y, float* __restrict__ g_x, float* __restrict__ t, int64_t D,
    float eps) {
  __shared__ float red[16];
  const int64_t r = blockIdx.x;
  const float* xr = x + r * D;
  const float* rr = residual ? residual + r * D : nullptr;
  const float* gr = g_y + r * D;
  float s = 0.f;
  for (int64_t d = threadIdx.x; d < D; d += 1024) s += xr[d] + (rr ? rr[d] : 0.f);
  const float mu = layernorm_block_sum(s, red) / (float)D;
  float v = 0.f;
  for (int64_t d = threadIdx.x; d < D; d += 1024) {
    const float c = xr[d] + (rr ? rr[d] : 0.f) - mu;
    v += c * c;
  }
  const float rstd = 1.0f / sqrtf(layernorm_block_sum(v, red) / (float)D + eps);
  float m1 = 0.f, m2 = 0.f;
  for (int64_t d = threadIdx.x; d < D; d += 1024) {
    const float xh = (xr[d] + (rr ? rr[d] : 0.f) - mu) * rstd;
    const float gh = gr[d] * (gamma ? gamma[d] : 1.f);
    m1 += gh;
    m2 += gh * xh;
  }
  m1 = layernorm_block_sum(m1, red) / (float)D;
  m2 = layernorm_block_sum(m2, red) / (float)D;
  for (int64_t d = threadIdx.x; d < D; d += 1024) {
    const float xh = (xr[d] + (rr ? rr[d] : 0.f) - mu) * rstd;
    const float gh = gr[d] * (gamma ? gamma[d] : 1.f);
    g_x[r * D + d] = rstd * (gh - m1 - xh * m2);
    if (t) t[r * D + d] = gr[d] * xh;
  }
}

static int launch_layernorm_backward(const float* x, const float* residual, const float* gamma,
                                     const float* g_y, float* g_x, float* t, int64_t rows, int64_t D,
                                     float eps, void* stream) {
  if (D >= 8192 && rows <= 0x7fffffff) {
    hipLaunchKernelGGL(layernorm_backward_wide_kernel, dim3((unsigned)rows), dim3(1024), 0,
                       static_cast<hipStream_t>(stream), x, residual, gamma, g_y, g_x, t, D, eps);
    return aps_launch_status();
  }
  const int64_t blocks = (rows + 3) / 4;
  if (blocks > 0x7fffffff) return APS_ERR_INVALID;
  hipLaunchKernelGGL(layernorm_backward_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, residual, gamma, g_y, g_x, t, rows, D, eps);
  return aps_launch_status();
}


// One reverse time step of an LSTM layer in ONE launch: g_h_rec = g_pre[:, t + 1, :] W_hh (K = 4H) on
// v_mfma_f32_16x16x4_f32 (exact fp32 products) fused with the gate adjoint of LstmBackwardStep
// (grad_core.h: the same arithmetic, index (n, u)).  Workgroup = (16 hidden units, 16 utterances) of
// 16 waves; wave w contracts over its sixteenth of K straight from HBM / L2 (both operands are
// K-contiguous rows: 16-byte loads feed 4 MFMAs each, the k order inside a group of 16 permuted
// identically on both sides), the partial sums meet in LDS, thread (row, unit) of the first four
// waves does the gate arithmetic.  Against [aps_linear on an M = N skinny GEMM (8 tiles on the chip,
// ~25 us) + a step launch] per time step this is one launch: the sweep was 22 of the 81 ms of the
// joint training step.
typedef float f32x4_g __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(1024) void lstm_backward_fused_kernel(
    const float* __restrict__ gates, const float* __restrict__ c, const float* __restrict__ g_y,
    const float* __restrict__ w_hh_t, const int64_t* __restrict__ lens, float* __restrict__ g_c,
    float* __restrict__ g_pre, int64_t N, int64_t T, int64_t H, int64_t t, int has_rec) {
  // 16 waves, each a sixteenth of K = 4H: at H = 512 eight 16-byte requests per operand and lane, all
  // in flight together (4 waves with 32 dependent batches of 4 took 12.9 us per step)
  __shared__ float s_red[16][16][17];
  const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
  const int64_t u0 = (int64_t)blockIdx.x * 16, n0 = (int64_t)blockIdx.y * 16;
  // the gate threads' operands are requested FIRST: their round trip runs beside the product's
  const int r = (tid >> 4) & 15, uu = tid & 15;
  const int64_t n = n0 + r, u = u0 + uu;
  const bool gate = tid < 256 && n < N;
  int64_t len = T;
  float gi = 0.f, gf = 0.f, gg = 0.f, g_o = 0.f, ct = 0.f, cp = 0.f, gy = 0.f, gc_in = 0.f;
  const int64_t base = ((gate ? n : 0) * T + t) * 4 * H + u;
  if (gate) {
    if (lens) len = lens[n] < 0 ? 0 : (lens[n] > T ? T : lens[n]);
    gi = gates[base], gf = gates[base + H], gg = gates[base + 2 * H], g_o = gates[base + 3 * H];
    ct = c[(n * T + t) * H + u];
    cp = t > 0 ? c[(n * T + t - 1) * H + u] : 0.f;
    gy = g_y[(n * T + t) * H + u];
    gc_in = g_c[n * H + u];
  }
  if (has_rec) {
    const int64_t row = min(n0 + (ln & 15), N - 1);
    const int64_t kw = H / 4;  // this wave's share of K
    const float* ap = g_pre + (row * T + t + 1) * 4 * H + wv * kw + 4 * (ln >> 4);
    const float* bp = w_hh_t + (u0 + (ln & 15)) * 4 * H + wv * kw + 4 * (ln >> 4);
    f32x4_g acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int64_t j = 0; j < kw; j += 16) {
      const float4 a = *reinterpret_cast<const float4*>(ap + j);
      const float4 b = *reinterpret_cast<const float4*>(bp + j);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc2, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc2, 0, 0, 0);
    }
    // D: lane l, register r -> (row 4 (l >> 4) + r, unit l & 15)
#pragma unroll
    for (int q = 0; q < 4; ++q) s_red[wv][4 * (ln >> 4) + q][ln & 15] = acc[q] + acc2[q];
  }
  __syncthreads();
  if (!gate) return;
  if (t >= len) {
    g_pre[base] = g_pre[base + H] = g_pre[base + 2 * H] = g_pre[base + 3 * H] = 0.f;
    return;
  }
  float gh = gy;
  if (has_rec && t + 1 < len) {
    float rec = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) rec += s_red[w][r][uu];
    gh += rec;
  }
  const float tc = tanhf(ct);
  const float gc = (t + 1 < len ? gc_in : 0.f) + gh * g_o * (1.f - tc * tc);
  g_pre[base] = gc * gg * gi * (1.f - gi);
  g_pre[base + H] = gc * cp * gf * (1.f - gf);
  g_pre[base + 2 * H] = gc * gi * (1.f - gg * gg);
  g_pre[base + 3 * H] = gh * tc * g_o * (1.f - g_o);
  g_c[n * H + u] = gc * gf;
}

// The reverse sweep of an LSTM layer as ONE persistent launch (H = 512, N <= 128): the team form of
// the forward recurrence (lstm.hip) turned around.  A team = the 32 workgroups with equal block id % 8
// (observed: one XCD, verified at run time as in lstm.hip) owns 16 utterances, a workgroup 16 hidden
// units and 16 waves.  g_pre [N, T, 4H] itself is the exchange buffer (sentinel-filled ahead of the
// launch, every cell written exactly once): step t gathers g_pre[:, t + 1, :] of its 16 utterances
// straight into MFMA operand registers (a lane needs exactly the words it requests: no LDS staging),
// contracts them with its resident slice of W_hh^T on v_mfma_f32_16x16x4_f32 (exact fp32: gradient
// magnitudes are not bounded like h), reduces the 16 waves' partial sums in LDS, and the 256 gate
// threads (cell gradient in a register for the whole sweep) publish the 4 gate gradients of their
// (utterance, unit).  One launch per time step cost 9.8 us (launch latency + two dependent round
// trips); a step of this kernel is one hand-off: 4.4 us (the exchange is 4H words per utterance,
// four times the forward's).  Like every persistent kernel here it needs all its workgroups resident:
// the launch covers the chip with one 1024-thread workgroup per CU, so it is taken only on a device
// with >= 256 CUs and the waits are bounded (a launch that shares the device with another process's
// persistent kernel ends with NaN gradients and a counted expiry instead of a hang).
constexpr unsigned kBwSentinel = 0xffffffffu;
constexpr unsigned kBwSpinLimit = 1u << 20;
typedef unsigned int u32x4_g __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(1024) void lstm_backward_team_kernel(
    const float* __restrict__ gates, const float* __restrict__ c, const float* __restrict__ g_y,
    const float* __restrict__ w_hh_t, const int64_t* __restrict__ lens, float* g_pre,
    unsigned* tab /* [8][32] placement words, then [1] expired waits */, int64_t N, int64_t T) {
  constexpr int H = 512, KW = 4 * H / 16;  // K share of a wave
  __shared__ float s_red[2][16][16][17];
  __shared__ int s_flag;
  const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
  const int team = (int)blockIdx.x & 7, b = (int)blockIdx.x >> 3;
  const int64_t n0 = (int64_t)team * 16;
  if (n0 >= N) return;
  const int64_t u0 = (int64_t)b * 16;
  // ---- placement: the short hand-off only if the whole team reports one XCC id
  const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) + 1u;
  unsigned* mine = tab + team * 32;
  if (tid == 0) __hip_atomic_store(mine + b, xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tid < 64) {
    unsigned seen = xcc;
    if (tid < 32) {
      unsigned spins = 0;
      do {
        seen = __hip_atomic_load(mine + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (seen == 0) __builtin_amdgcn_s_sleep(1);
      } while (seen == 0 && ++spins < kBwSpinLimit);
    }
    const bool all_here = __ballot(seen != xcc) == 0;
    if (tid == 0) s_flag = all_here ? 1 : 0;
  }
  __syncthreads();
  const bool plain_publish = s_flag != 0;
  // ---- resident slice of W_hh^T: lane (unit l & 15, k group l >> 4), the k order of the fused kernel
  float4 wreg[KW / 16];
  {
    const float* bp = w_hh_t + (u0 + (ln & 15)) * 4 * H + wv * KW + 4 * (ln >> 4);
#pragma unroll
    for (int j = 0; j < KW / 16; ++j) wreg[j] = *reinterpret_cast<const float4*>(bp + 16 * j);
  }
  // ---- gate role (first 256 threads)
  const int r = (tid >> 4) & 15, uu = tid & 15;
  const int64_t n = n0 + r, u = u0 + uu;
  const bool gate = tid < 256 && n < N;
  int64_t len = T;
  if (gate && lens) len = lens[n] < 0 ? 0 : (lens[n] > T ? T : lens[n]);
  float gc_carry = 0.f;
  const uint32_t bytes = (uint32_t)(N * T * 4 * H * 4);
  auto rsrc = __builtin_amdgcn_make_buffer_rsrc(g_pre, 0, bytes, 0x00020000);
  const int64_t row = min(n0 + (ln & 15), N - 1);
  const int32_t voff = (int32_t)((row * T * 4 * H + wv * KW + 4 * (ln >> 4)) * 4);
  const int32_t step_bytes = 4 * H * 4;
  bool timed_out = false;
  // the gate threads' operands are requested a step ahead (HBM latency off the per-step chain)
  float nx[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto fetch = [&](int64_t t, float (&o)[7]) {
    const int64_t bs = ((gate ? n : 0) * T + t) * 4 * H + u;
    if (gate && t >= 0 && t < len) {
      o[0] = gates[bs], o[1] = gates[bs + H], o[2] = gates[bs + 2 * H], o[3] = gates[bs + 3 * H];
      o[4] = c[(n * T + t) * H + u];
      o[5] = t > 0 ? c[(n * T + t - 1) * H + u] : 0.f;
      o[6] = g_y[(n * T + t) * H + u];
    }
  };
  fetch(T - 1, nx);
  for (int64_t t = T - 1; t >= 0; --t) {
    float (*red)[16][17] = s_red[t & 1];
    const float gi = nx[0], gf = nx[1], gg = nx[2], g_o = nx[3], ct = nx[4], cp = nx[5], gy = nx[6];
    const int64_t base = ((gate ? n : 0) * T + t) * 4 * H + u;
    fetch(t - 1, nx);
    if (t + 1 < T) {
      u32x4_g v[KW / 16];
      const int32_t soff = (int32_t)((t + 1) * step_bytes);
      unsigned spins = 0;
      // (one 16-byte probe per lane until its last chunk has arrived: a full request moves 128 KB
      // per workgroup -- 4 MB per XCD and attempt -- and the first attempt is always early)
      while (!timed_out) {
        const u32x4_g probe = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 64 * (KW / 16 - 1), soff, 16);
        if (max(max(probe.x, probe.y), max(probe.z, probe.w)) != kBwSentinel) break;
        if (++spins > kBwSpinLimit) {
          timed_out = true;
          __hip_atomic_fetch_add(tab + 256, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      while (true) {
#pragma unroll
        for (int j = 0; j < KW / 16; ++j)
          v[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 64 * j, soff, 16);
        unsigned top = 0;
#pragma unroll
        for (int j = 0; j < KW / 16; ++j) top = max(top, max(max(v[j].x, v[j].y), max(v[j].z, v[j].w)));
        if (top != kBwSentinel || timed_out) break;
        if (++spins > kBwSpinLimit) {
          timed_out = true;
          __hip_atomic_fetch_add(tab + 256, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      f32x4_g acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < KW / 16; ++j) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(v[j].x), wreg[j].x, acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(v[j].y), wreg[j].y, acc2, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(v[j].z), wreg[j].z, acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(v[j].w), wreg[j].w, acc2, 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) red[wv][4 * (ln >> 4) + q][ln & 15] = acc[q] + acc2[q];
    }
    __syncthreads();  // (s_red alternates between two buffers: one barrier per step)
    if (gate) {
      float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
      if (t < len) {
        float gh = gy;
        if (t + 1 < T && t + 1 < len) {
          float rec = 0.f;
#pragma unroll
          for (int w = 0; w < 16; ++w) rec += red[w][r][uu];
          gh += rec;
        }
        const float tc = tanhf(ct);
        const float gc = (t + 1 < len ? gc_carry : 0.f) + gh * g_o * (1.f - tc * tc);
        o0 = gc * gg * gi * (1.f - gi);
        o1 = gc * cp * gf * (1.f - gf);
        o2 = gc * gi * (1.f - gg * gg);
        o3 = gh * tc * g_o * (1.f - g_o);
        gc_carry = gc * gf;
      }
      const int32_t off = (int32_t)(base * 4);
      if (plain_publish) {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o0), rsrc, off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o1), rsrc, off + H * 4, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o2), rsrc, off + 2 * H * 4, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o3), rsrc, off + 3 * H * 4, 0, 0);
      } else {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o0), rsrc, off, 0, 16);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o1), rsrc, off + H * 4, 0, 16);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o2), rsrc, off + 2 * H * 4, 0, 16);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o3), rsrc, off + 3 * H * 4, 0, 16);
      }
    }
  }
}

// Attention backward, the row pass (AttentionBackwardRowsFast<64> of grad_core.h: scores, softmax,
// P^T and dS^T into the workspace, g_q) as a cooperative kernel: one workgroup per (utterance, head)
// stages K, V and the 2T - 1 rows of the relative table the head's offsets touch in LDS ONCE (all
// global reads coalesced), then each wave takes every 4th query row with its lanes along the keys
// (scores, softmax and dS by wave reductions) and, for g_q, along head_dim.  The thread-per-row
// functor kept q, g and the g_q accumulator in 256 registers and walked the keys serially with one
// wave per SIMD on a quarter of the chip: 1.06 ms per call, 12.7 ms of the joint training step.
// T <= 128, head_dim 64; anything else runs the functor.
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__global__ __launch_bounds__(256) void attention_backward_rows_kernel(grad::AttentionGeometry a,
                                                                      float* __restrict__ pt,
                                                                      float* __restrict__ dst,
                                                                      float* __restrict__ g_qkv) {
  constexpr int DH = 64, KP = DH + 1;
  extern __shared__ float s_att_bw[];
  const int T = (int)a.T;
  const bool rel = a.rel != nullptr;
  float* s_k = s_att_bw;                            // [T][65]
  float* s_v = s_k + T * KP;                        // [T][65]
  float* s_e = s_v + T * KP;                        // [2T - 1][65]: row w <-> offset j - i = w - (T - 1)
  float* s_q = s_e + (rel ? (2 * T - 1) * KP : 0);  // [4 waves][64]
  float* s_g = s_q + 4 * DH;                        // [4][64]
  float* s_ds = s_g + 4 * DH;                       // [4][128]
  const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
  const int64_t h = blockIdx.x, n = blockIdx.y;
  const int L = (int)a.keys(n);
  for (int e = tid; e < T * 16; e += 256) {
    const int r = e >> 4, c4 = (e & 15) * 4;
    const float4 k = *reinterpret_cast<const float4*>(a.k(n, r, h) + c4);
    const float4 v = *reinterpret_cast<const float4*>(a.v(n, r, h) + c4);
    float* kd = s_k + r * KP + c4;
    float* vd = s_v + r * KP + c4;
    kd[0] = k.x, kd[1] = k.y, kd[2] = k.z, kd[3] = k.w;
    vd[0] = v.x, vd[1] = v.y, vd[2] = v.z, vd[3] = v.w;
  }
  if (rel) {
    for (int e = tid; e < (2 * T - 1) * 16; e += 256) {
      const int w = e >> 4, c4 = (e & 15) * 4;
      const int64_t r = (int64_t)w - (T - 1) + a.rel_zero;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r >= 0 && r < a.rel_len) v = *reinterpret_cast<const float4*>(a.rel + r * DH + c4);
      float* ed = s_e + w * KP + c4;
      ed[0] = v.x, ed[1] = v.y, ed[2] = v.z, ed[3] = v.w;
    }
  }
  __syncthreads();
  float* wq = s_q + wv * DH;
  float* wg = s_g + wv * DH;
  float* wds = s_ds + wv * 128;
  for (int i = wv; i < T; i += 4) {
    wq[ln] = a.q(n, i, h)[ln];
    wg[ln] = a.g(n, i, h)[ln];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (one wave: LDS serves it in order)
    float sc[2], dp[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int j = ln + 64 * b;
      sc[b] = -INFINITY, dp[b] = 0.f;
      if (j < L) {
        const float* kj = s_k + j * KP;
        const float* vj = s_v + j * KP;
        const float* ej = s_e + (j - i + T - 1) * KP;
        float s = 0.f, d = 0.f;
        if (rel) {
#pragma unroll 8
          for (int x = 0; x < DH; ++x) s += wq[x] * (kj[x] + ej[x]);
        } else {
#pragma unroll 8
          for (int x = 0; x < DH; ++x) s += wq[x] * kj[x];
        }
#pragma unroll 8
        for (int x = 0; x < DH; ++x) d += wg[x] * vj[x];
        sc[b] = s * a.scale;
        dp[b] = d * a.keep(n, h, i, j);
      }
    }
    const float mx = wave_max_f(fmaxf(sc[0], sc[1]));
    float p[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) p[b] = (ln + 64 * b < L) ? expf(sc[b] - mx) : 0.f;
    const float sum = wave_sum_f(p[0] + p[1]);
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;
    p[0] *= inv, p[1] *= inv;
    const float D = wave_sum_f(p[0] * dp[0] + p[1] * dp[1]);
    float* prow = pt + ((n * a.H + h) * T) * (int64_t)T + i;  // element (j, i) at prow[j * T]
    float* drow = dst + ((n * a.H + h) * T) * (int64_t)T + i;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int j = ln + 64 * b;
      const float ds = p[b] * (dp[b] - D) * a.scale;  // (0 for the masked keys: p = 0)
      wds[j] = ds;
      if (j < T) {
        prow[(int64_t)j * T] = p[b];
        drow[(int64_t)j * T] = ds;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float acc = 0.f;  // lane = head dimension
    if (rel) {
      for (int j = 0; j < L; ++j) acc += wds[j] * (s_k[j * KP + ln] + s_e[(j - i + T - 1) * KP + ln]);
    } else {
      for (int j = 0; j < L; ++j) acc += wds[j] * s_k[j * KP + ln];
    }
    g_qkv[((n * T + i) * 3 * a.H + h) * DH + ln] = acc;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the row's LDS reads are done before the next row's writes
  }
}

// The whole attention backward of one (utterance, head) in ONE workgroup when the sequence fits a
// tile (T <= 64, head_dim 64: the joint model's 63 encoder frames): K, V, Q, g_ctx, the table window,
// P and dS all live in LDS (131 KB), nothing goes through the T x T workspace.  Phase 1 = the row
// pass above (g_q), phase 2 = the columns (g_k[j] = sum_i dS[i, j] q_i, g_v[j] = sum_i P[i, j] keep
// g_i: a wave per key, lanes along head_dim), phase 3 = the table (partial[r] = sum_i dS[i, i + r -
// zero] q_i: a wave per table row).  The arithmetic of AttentionBackward{Rows,Columns,Table}Fast<64>
// (grad_core.h); three launches of 91 + 116 + 93 us became one.
__global__ __launch_bounds__(1024) void attention_backward_fused_kernel(grad::AttentionGeometry a,
                                                                       float* __restrict__ g_qkv,
                                                                       float* __restrict__ g_rel_partial) {
  constexpr int DH = 64, KP = DH + 1;
  extern __shared__ float s_att_bw[];
  const int T = (int)a.T;
  const bool rel = a.rel != nullptr;
  float* s_k = s_att_bw;         // [T][65]
  float* s_v = s_k + T * KP;     // [T][65]
  float* s_q = s_v + T * KP;     // [T][65]
  float* s_g = s_q + T * KP;     // [T][65]
  float* s_p = s_g + T * KP;     // [T][65]  P[i][j]
  float* s_d = s_p + T * KP;     // [T][65]  dS[i][j] (times scale)
  float* s_e = s_d + T * KP;     // [2T - 1][65]: row w <-> offset j - i = w - (T - 1)
  // 16 waves: the phases are bound by LDS latency / bandwidth (every product term is an LDS read), so
  // the rows, keys and table rows of a head are spread over as many waves as a workgroup holds
  constexpr int NW = 16;
  const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
  const int64_t h = blockIdx.x, n = blockIdx.y;
  const int L = (int)a.keys(n);
  for (int e = tid; e < T * 16; e += 64 * NW) {
    const int r = e >> 4, c4 = (e & 15) * 4;
    const float4 k = *reinterpret_cast<const float4*>(a.k(n, r, h) + c4);
    const float4 v = *reinterpret_cast<const float4*>(a.v(n, r, h) + c4);
    const float4 q = *reinterpret_cast<const float4*>(a.q(n, r, h) + c4);
    const float4 g = *reinterpret_cast<const float4*>(a.g(n, r, h) + c4);
    float* kd = s_k + r * KP + c4;
    float* vd = s_v + r * KP + c4;
    float* qd = s_q + r * KP + c4;
    float* gd = s_g + r * KP + c4;
    kd[0] = k.x, kd[1] = k.y, kd[2] = k.z, kd[3] = k.w;
    vd[0] = v.x, vd[1] = v.y, vd[2] = v.z, vd[3] = v.w;
    qd[0] = q.x, qd[1] = q.y, qd[2] = q.z, qd[3] = q.w;
    gd[0] = g.x, gd[1] = g.y, gd[2] = g.z, gd[3] = g.w;
  }
  if (rel) {
    for (int e = tid; e < (2 * T - 1) * 16; e += 64 * NW) {
      const int w = e >> 4, c4 = (e & 15) * 4;
      const int64_t r = (int64_t)w - (T - 1) + a.rel_zero;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r >= 0 && r < a.rel_len) v = *reinterpret_cast<const float4*>(a.rel + r * DH + c4);
      float* ed = s_e + w * KP + c4;
      ed[0] = v.x, ed[1] = v.y, ed[2] = v.z, ed[3] = v.w;
    }
  }
  __syncthreads();
  // ---- phase 1: rows (lanes along the keys, then along head_dim for g_q)
  for (int i = wv; i < T; i += NW) {
    const float* wq = s_q + i * KP;
    const float* wg = s_g + i * KP;
    const int j = ln;
    float sc = -INFINITY, dp = 0.f;
    if (j < L) {
      const float* kj = s_k + j * KP;
      const float* vj = s_v + j * KP;
      const float* ej = s_e + (j - i + T - 1) * KP;
      float sdot = 0.f, d = 0.f;
      if (rel) {
#pragma unroll 8
        for (int x = 0; x < DH; ++x) sdot += wq[x] * (kj[x] + ej[x]);
      } else {
#pragma unroll 8
        for (int x = 0; x < DH; ++x) sdot += wq[x] * kj[x];
      }
#pragma unroll 8
      for (int x = 0; x < DH; ++x) d += wg[x] * vj[x];
      sc = sdot * a.scale;
      dp = d * a.keep(n, h, i, j);
    }
    const float mx = wave_max_f(sc);
    float p = (j < L) ? expf(sc - mx) : 0.f;
    const float sum = wave_sum_f(p);
    p *= sum > 0.f ? 1.0f / sum : 0.f;
    const float D = wave_sum_f(p * dp);
    const float ds = p * (dp - D) * a.scale;
    s_p[i * KP + j] = p;   // (lane j = 64 lands in the pad column: harmless, never read)
    s_d[i * KP + j] = ds;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (one wave: LDS serves it in order)
    float acc = 0.f;  // lane = head dimension
    const float* drow = s_d + i * KP;
    if (rel) {
      for (int jj = 0; jj < L; ++jj) acc += drow[jj] * (s_k[jj * KP + ln] + s_e[(jj - i + T - 1) * KP + ln]);
    } else {
      for (int jj = 0; jj < L; ++jj) acc += drow[jj] * s_k[jj * KP + ln];
    }
    g_qkv[((n * T + i) * 3 * a.H + h) * DH + ln] = acc;
  }
  __syncthreads();
  // ---- phase 2: columns (a wave per key, lanes along head_dim)
  for (int j = wv; j < T; j += NW) {
    float ak = 0.f, av = 0.f;
    for (int i = 0; i < T; ++i) {
      const float p = s_p[i * KP + j] * a.keep(n, h, i, j), ds = s_d[i * KP + j];
      ak += ds * s_q[i * KP + ln];
      av += p * s_g[i * KP + ln];
    }
    float* gk = g_qkv + ((n * T + j) * 3 * a.H + a.H + h) * DH;
    gk[ln] = ak;
    gk[a.H * DH + ln] = av;
  }
  // ---- phase 3: the table's partial sums of this (utterance, head)
  if (rel) {
    for (int r = wv; r < (int)a.rel_len; r += NW) {
      float acc = 0.f;
      for (int i = 0; i < T; ++i) {
        const int j = i + r - (int)a.rel_zero;
        if (j < 0 || j >= T) continue;
        acc += s_d[i * KP + j] * s_q[i * KP + ln];
      }
      g_rel_partial[((n * a.H + h) * a.rel_len + r) * DH + ln] = acc;
    }
  }
}

static int launch_attention_backward_fused(const grad::AttentionGeometry& g, float* g_qkv,
                                           float* g_rel_partial, int64_t N, void* stream) {
  if (g.dh != 64 || g.T > 64 || N > 65535 || g.H > 0x7fffffff) return APS_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(g.qkv) & 15) || (reinterpret_cast<uintptr_t>(g.g_ctx) & 15) ||
      (g.rel && (reinterpret_cast<uintptr_t>(g.rel) & 15)))
    return APS_ERR_UNSUPPORTED;
  static const bool off = [] { const char* e = getenv("APS_ATT_BACKWARD"); return e && e[0] != 0; }();
  if (off) return APS_ERR_UNSUPPORTED;  // APS_ATT_BACKWARD=rows / functor: the multi-launch forms (A/B)
  const size_t lds = (size_t)(6 * g.T + (g.rel ? 2 * g.T - 1 : 0)) * 65 * sizeof(float);
  static ApsPerDevice attr_set;
  if (lds > 64 * 1024 &&
      !aps_lds_opt_in(attr_set, reinterpret_cast<const void*>(&attention_backward_fused_kernel), 160 * 1024))
    return APS_ERR_LAUNCH;
  hipLaunchKernelGGL(attention_backward_fused_kernel, dim3((unsigned)g.H, (unsigned)N), dim3(1024), lds,
                     static_cast<hipStream_t>(stream), g, g_qkv, g_rel_partial);
  return aps_launch_status();
}

static int launch_attention_backward_rows(const grad::AttentionGeometry& g, float* pt, float* dst,
                                          float* g_qkv, int64_t N, void* stream) {
  if (g.dh != 64 || g.T > 128 || N > 65535 || g.H > 0x7fffffff) return APS_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(g.qkv) & 15) || (g.rel && (reinterpret_cast<uintptr_t>(g.rel) & 15)))
    return APS_ERR_UNSUPPORTED;
  static const bool off = [] { const char* e = getenv("APS_ATT_BACKWARD"); return e && e[0] == 'f'; }();
  if (off) return APS_ERR_UNSUPPORTED;  // APS_ATT_BACKWARD=functor: the thread-per-row form (A/B)
  const size_t lds = ((size_t)(2 * g.T + (g.rel ? 2 * g.T - 1 : 0)) * 65 + 8 * 64 + 4 * 128) * sizeof(float);
  static ApsPerDevice attr_set;
  if (lds > 64 * 1024 &&
      !aps_lds_opt_in(attr_set, reinterpret_cast<const void*>(&attention_backward_rows_kernel), 160 * 1024))
    return APS_ERR_LAUNCH;
  hipLaunchKernelGGL(attention_backward_rows_kernel, dim3((unsigned)g.H, (unsigned)N), dim3(256), lds,
                     static_cast<hipStream_t>(stream), g, pt, dst, g_qkv);
  return aps_launch_status();
}

// ------------------------------------------------------------------------------------------
// The two frame-axis reductions of the MVDR adjoint as workgroup kernels (round 4).  The functors of
// grad_core.h (CovarianceBackward, BeamformBackwardWeight: one thread per (utterance, bin) walking all T
// frames, which the host build checks) left 8 224 threads on the chip for 0.46 / 0.43 ms per call of the
// joint model's training step; here a workgroup owns 32 bins of one utterance with 8 frame phases per bin
// (lanes along bins: 128-byte runs of the bin-fastest store), the phases folded through LDS.  Same
// arithmetic, sums taken phase by phase.  APS_GRAD_FUNCTORS=1: the functor forms (A/B, tests).
// ------------------------------------------------------------------------------------------
constexpr int kFrBins = 32, kFrPhases = 8;

__device__ __forceinline__ float fold_phases(float (*s)[kFrBins], int tp, int fl, float v, bool take_max) {
  __syncthreads();
  s[tp][fl] = v;
  __syncthreads();
  float r = s[0][fl];
#pragma unroll
  for (int q = 1; q < kFrPhases; ++q) r = take_max ? fmaxf(r, s[q][fl]) : r + s[q][fl];
  return r;
}

template <int C>
__global__ __launch_bounds__(256) void covariance_backward_frames_kernel(grad::CovarianceBackward<C> a) {
  __shared__ float s_red[kFrPhases][kFrBins];
  const int fl = threadIdx.x & (kFrBins - 1), tp = threadIdx.x / kFrBins;
  const int64_t n = blockIdx.y, f = (int64_t)blockIdx.x * kFrBins + fl;
  const bool on = f < a.F;
  const int64_t fc = on ? f : a.F - 1, T = a.T, F = a.F;
  int64_t len = T;
  if (a.lens) len = a.lens[n] < 0 ? 0 : (a.lens[n] > T ? T : a.lens[n]);
  const float* mk = a.mask + n * T * F + fc;
  float* gm = a.g_mask + n * T * F + fc;
  const float* gs = a.g_sub ? a.g_sub + n * T * F + fc : nullptr;
  float peak = 0.f, nmax = 0.f;
  if (a.mask_norm) {
    float mx = 0.f;
    for (int64_t t = tp; t < len; t += kFrPhases) mx = fmaxf(mx, fabsf(mk[t * F]));
    peak = fold_phases(s_red, tp, fl, mx, true);
    float cnt = 0.f;
    for (int64_t t = tp; t < T; t += kFrPhases) {  // zeroed padded frames take part in the tie count at 0
      const float v = t < len ? fabsf(mk[t * F]) : 0.f;
      cnt += v == peak ? 1.f : 0.f;
    }
    nmax = fold_phases(s_red, tp, fl, cnt, false);
  }
  const float s = a.mask_norm ? peak + grad::kEps : 1.f;
  float part = 0.f;
  for (int64_t t = tp; t < len; t += kFrPhases) part += mk[t * F] / s;
  const float msum = fold_phases(s_red, tp, fl, part, false);
  const bool clamped = !(msum > grad::kEps);
  const float den = clamped ? grad::kEps : msum;
  cf G[C][C];
  float r = 0.f;
  const int64_t idx = n * F + fc;
  const float* pg = a.g_cov + idx * C * C * 2;
  const float* pr = a.cov + idx * C * C * 2;
#pragma unroll
  for (int i = 0; i < C; ++i)
#pragma unroll
    for (int j = 0; j < C; ++j) {
      G[i][j] = {pg[(i * C + j) * 2], pg[(i * C + j) * 2 + 1]};
      r += G[i][j].re * pr[(i * C + j) * 2] + G[i][j].im * pr[(i * C + j) * 2 + 1];
    }
  if (clamped) r = 0.f;
  float dot = 0.f;
  const float* xs = a.store + n * a.stride_n + 2 * fc;
  for (int64_t t = tp; t < T; t += kFrPhases) {
    if (t >= len) {
      if (on) gm[t * F] = 0.f;
      continue;
    }
    cf x[C];
#pragma unroll
    for (int c = 0; c < C; ++c)
      x[c] = {xs[c * a.stride_c + t * a.stride_t], xs[c * a.stride_c + t * a.stride_t + 1]};
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < C; ++i)
#pragma unroll
      for (int j = 0; j < C; ++j) {
        const cf p = cmul(x[i], cconj(x[j]));
        q += G[i][j].re * p.re + G[i][j].im * p.im;
      }
    const float g = (q - r) / den - (gs ? gs[t * F] : 0.f);
    if (on) gm[t * F] = g;
    dot += g * mk[t * F];
  }
  if (!a.mask_norm) return;
  const float total = fold_phases(s_red, tp, fl, dot, false);
  const float back = total / (s * s * (nmax > 0.f ? nmax : 1.f));
  if (!on) return;
  for (int64_t t = tp; t < len; t += kFrPhases) {  // (each thread rewrites the frames it wrote itself)
    const float m = mk[t * F];
    float g = gm[t * F] / s;
    if (fabsf(m) == peak) g -= (m > 0.f ? 1.f : (m < 0.f ? -1.f : 0.f)) * back;
    gm[t * F] = g;
  }
}

template <int C>
__global__ __launch_bounds__(256) void beamform_backward_weight_frames_kernel(grad::BeamformBackwardWeight a) {
  __shared__ float s_red[kFrPhases][kFrBins];
  const int fl = threadIdx.x & (kFrBins - 1), tp = threadIdx.x / kFrBins;
  const int64_t n = blockIdx.y, f = (int64_t)blockIdx.x * kFrBins + fl;
  const bool on = f < a.F;
  const int64_t fc = on ? f : a.F - 1, T = a.T, F = a.F;
  float re[C], im[C];
#pragma unroll
  for (int c = 0; c < C; ++c) re[c] = im[c] = 0.f;
  const float* xs = a.store + n * a.stride_n + 2 * fc;
  for (int64_t t = tp; t < T; t += kFrPhases) {
    const float gr = a.g_y[((n * T + t) * F + fc) * 2], gi = a.g_y[((n * T + t) * F + fc) * 2 + 1];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float xr = xs[c * a.stride_c + t * a.stride_t], xi = xs[c * a.stride_c + t * a.stride_t + 1];
      re[c] += gr * xr + gi * xi;  // conj(G) x
      im[c] += gr * xi - gi * xr;
    }
  }
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const float sr = fold_phases(s_red, tp, fl, re[c], false);
    const float si = fold_phases(s_red, tp, fl, im[c], false);
    if (on && tp == 0) {
      a.g_w[((n * F + f) * C + c) * 2] = sr;
      a.g_w[((n * F + f) * C + c) * 2 + 1] = si;
    }
  }
}

// aps_colreduce in ONE launch for the matrices of a training step (rows <= 16 384: 247 -> 141 two-launch
// reductions per step were 1.7 ms): a workgroup owns 32 columns with 32 row phases, lanes along the columns,
// the phases folded through LDS in a fixed order (deterministic).  Same element functions as ColReducePartial.
constexpr int kCrPhases = 32;  // row phases of a column (1024 threads: 32 columns x 32 phases)
__global__ __launch_bounds__(1024) void colreduce_columns_kernel(grad::ColReducePartial a, float scale,
                                                                 int accumulate, float* __restrict__ out) {
  __shared__ float s_red[kCrPhases][kFrBins];
  const int cl = threadIdx.x & (kFrBins - 1), tp = threadIdx.x / kFrBins;
  const int64_t c = (int64_t)blockIdx.x * kFrBins + cl;
  const bool on = c < a.cols;
  const int64_t cc = on ? c : a.cols - 1;
  const int mode = a.mode;
  const int64_t half = a.cols / 2;
  const float* pa = mode == 4 ? (cc < half ? a.A + cc : a.B + (cc - half)) : a.A + cc;
  const int64_t lda = mode == 4 ? (cc < half ? a.lda : a.ldb) : a.lda;
  const float* pb = (mode == 1 || mode == 3) ? a.B + cc : nullptr;
  const float m = (mode == 2 || mode == 3) ? a.v1[cc] : 0.f;
  const float sc = mode == 3 ? a.v2[cc] : 0.f;
  auto term = [&](float x, float y) {
    return mode == 1 ? x * y : (mode == 2 ? (x - m) * (x - m) : (mode == 3 ? x * (y - m) * sc : x));
  };
  float acc = 0.f;
  int64_t r = tp;
  // 8 rows requested together (a thread's rows are kCrPhases apart), summed in row order
  for (; r + 7 * kCrPhases < a.rows; r += 8 * kCrPhases) {
    float x[8], y[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      x[u] = pa[(r + u * kCrPhases) * lda];
      y[u] = pb ? pb[(r + u * kCrPhases) * a.ldb] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += term(x[u], y[u]);
  }
  for (; r < a.rows; r += kCrPhases) acc += term(pa[r * lda], pb ? pb[r * a.ldb] : 0.f);
  s_red[tp][cl] = acc;
  __syncthreads();
  if (tp == 0 && on) {
    float total = 0.f;
#pragma unroll
    for (int q = 0; q < kCrPhases; ++q) total += s_red[q][cl];
    total *= scale;
    out[c] = accumulate ? out[c] + total : total;
  }
}

static int launch_colreduce_columns(const grad::ColReducePartial& op, float scale, int accumulate, float* out,
                                    void* stream);

static bool grad_functors_forced() {
  static const bool on = [] { const char* e = getenv("APS_GRAD_FUNCTORS"); return e && e[0] == '1'; }();
  return on;
}

static int launch_colreduce_columns(const grad::ColReducePartial& op, float scale, int accumulate, float* out,
                                    void* stream) {
  if (grad_functors_forced() || op.rows > 16384) return APS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(colreduce_columns_kernel, dim3((unsigned)((op.cols + kFrBins - 1) / kFrBins)), dim3(1024), 0,
                     static_cast<hipStream_t>(stream), op, scale, accumulate, out);
  return aps_launch_status();
}

template <int C>
static int launch_covariance_backward_frames(const grad::CovarianceBackward<C>& op, int64_t N, void* stream) {
  if (grad_functors_forced() || N > 65535) return APS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((covariance_backward_frames_kernel<C>), dim3((unsigned)((op.F + kFrBins - 1) / kFrBins), (unsigned)N),
                     dim3(256), 0, static_cast<hipStream_t>(stream), op);
  return aps_launch_status();
}

static int launch_beamform_backward_weight_frames(const grad::BeamformBackwardWeight& op, int64_t N, void* stream) {
  if (grad_functors_forced() || N > 65535) return APS_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)((op.F + kFrBins - 1) / kFrBins), (unsigned)N);
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (op.C) {
    case 2: hipLaunchKernelGGL((beamform_backward_weight_frames_kernel<2>), grid, dim3(256), 0, st, op); break;
    case 4: hipLaunchKernelGGL((beamform_backward_weight_frames_kernel<4>), grid, dim3(256), 0, st, op); break;
    case 6: hipLaunchKernelGGL((beamform_backward_weight_frames_kernel<6>), grid, dim3(256), 0, st, op); break;
    case 8: hipLaunchKernelGGL((beamform_backward_weight_frames_kernel<8>), grid, dim3(256), 0, st, op); break;
    default: return APS_ERR_UNSUPPORTED;
  }
  return aps_launch_status();
}

}  // namespace aps

#define APS_GRAD_ATTENTION_ROWS_KERNEL aps::launch_attention_backward_rows
#define APS_GRAD_ATTENTION_FUSED_KERNEL aps::launch_attention_backward_fused
#define APS_GRAD_LAYERNORM_WAVE_KERNEL aps::launch_layernorm_backward
#define APS_GRAD_COLREDUCE_COLUMNS_KERNEL aps::launch_colreduce_columns
#define APS_GRAD_COVARIANCE_FRAMES_KERNEL aps::launch_covariance_backward_frames
#define APS_GRAD_BEAMFORM_WEIGHT_FRAMES_KERNEL aps::launch_beamform_backward_weight_frames
#define APS_GRAD_API(name) aps_##name
#define APS_GRAD_EACH(op, n, stream) aps::launch_each(op, n, stream)
#include "grad_api.inc"

extern "C" int aps_transpose(const float* in, float* out, int64_t rows, int64_t cols, int64_t ld_in,
                             int64_t ld_out, void* stream) {
  APS_CHECK_ARG(in && out && rows > 0 && cols > 0 && ld_in >= cols && ld_out >= rows);
  const int64_t gx = (cols + 31) / 32, gy = (rows + 31) / 32;
  if (gx > 0x7fffffff || gy > 65535) return APS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(aps::transpose_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0,
                     static_cast<hipStream_t>(stream), in, out, rows, cols, ld_in, ld_out);
  return aps_launch_status();
}

// reverse-time sweep of one LSTM layer: g_pre[n, t, :] for t = T-1 .. 0.  w_hh_t = W_hh^T [H, 4H].
extern "C" int aps_lstm_backward_sweep(const float* gates, const float* c, const float* g_y,
                                       const float* w_hh_t, const int64_t* lens, float* g_pre,
                                       float* g_h_rec, float* g_c, int64_t N, int64_t T, int64_t H,
                                       void* stream) {
  APS_CHECK_ARG(gates && c && g_y && w_hh_t && g_pre && g_h_rec && g_c && N > 0 && T > 0 && H > 0 &&
                H % 4 == 0);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (aps_fill_u32(g_c, 0u, (size_t)N * H, st) != APS_OK) return APS_ERR_LAUNCH;
  // one fused launch per step where the geometry allows (unit blocks of 16, 16-byte operand rows);
  // APS_LSTM_SWEEP=steps keeps the GEMM + step pair (A/B, and the shapes outside)
  static const bool fused_on = [] {
    const char* e = getenv("APS_LSTM_SWEEP");
    return !(e && e[0] == 's');
  }();
  // H = 512, N <= 128: the whole sweep in one persistent launch (APS_LSTM_SWEEP=fused: the launch per
  // step instead); g_h_rec (N x H floats of scratch) carries its placement table and time-out word
  static const bool team_on = [] {
    const char* e = getenv("APS_LSTM_SWEEP");
    return !(e && e[0] != 0);
  }();
  if (team_on && H == 512 && N <= 128 && N * T * 4 * H * 4 < ((int64_t)1 << 32) &&
      (reinterpret_cast<uintptr_t>(w_hh_t) & 15) == 0 && (reinterpret_cast<uintptr_t>(g_pre) & 15) == 0) {
    static ApsPerDevice cus;
    int dev = aps_current_device(), ncu = dev >= 0 ? cus.get(dev) : 0;
    if (dev >= 0 && ncu == 0) {
      if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) ncu = 0;
      if (ncu > 0) cus.set(dev, ncu);
    }
    if (ncu >= 256) {  // every workgroup of the launch resident: one 1024-thread workgroup per CU
      unsigned* tab = reinterpret_cast<unsigned*>(g_h_rec);
      if (aps_fill_u32(tab, 0u, 260, st) != APS_OK) return APS_ERR_LAUNCH;
      if (aps_fill_u32(g_pre, aps::kBwSentinel, (size_t)N * T * 4 * H, st) != APS_OK) return APS_ERR_LAUNCH;
      hipLaunchKernelGGL(aps::lstm_backward_team_kernel, dim3(256), dim3(1024), 0, st, gates, c, g_y,
                         w_hh_t, lens, g_pre, tab, N, T);
      return aps_launch_status();
    }
  }
  if (fused_on && H % 64 == 0 && ((N + 15) / 16) <= 65535 &&
      (reinterpret_cast<uintptr_t>(w_hh_t) & 15) == 0 && (reinterpret_cast<uintptr_t>(g_pre) & 15) == 0) {
    const dim3 grid((unsigned)(H / 16), (unsigned)((N + 15) / 16));
    for (int64_t t = T - 1; t >= 0; --t)
      hipLaunchKernelGGL(aps::lstm_backward_fused_kernel, grid, dim3(1024), 0, st, gates, c, g_y, w_hh_t,
                         lens, g_c, g_pre, N, T, H, t, (int)(t + 1 < T));
    return aps_launch_status();
  }
  for (int64_t t = T - 1; t >= 0; --t) {
    const float* rec = nullptr;
    if (t + 1 < T) {
      int rc = aps_linear(g_pre + (t + 1) * 4 * H, w_hh_t, nullptr, nullptr, g_h_rec, N, H, 4 * H,
                          T * 4 * H, 4 * H, H, 0, 1.0f, stream);
      if (rc != APS_OK) return rc;
      rec = g_h_rec;
    }
    int rc = aps_lstm_backward_step(gates, c, g_y, rec, lens, g_c, g_pre, N, T, H, t, stream);
    if (rc != APS_OK) return rc;
  }
  return APS_OK;
}
