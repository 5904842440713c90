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

static int launch_layernorm_backward(const float* x, const float* residual, const float* gamma,
                                     const float* g_y, float* g_x, float* t, int64_t rows, int64_t D,
                                     float eps, void* stream) {
  const int64_t blocks = (rows + 3) / 4;
  if (blocks > 0x7fffffff) return APS_ERR_INVALID;
  hipLaunchKernelGGL(layernorm_backward_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, residual, gamma, g_y, g_x, t, rows, D, eps);
  return aps_launch_status();
}

}  // namespace aps

#define APS_GRAD_LAYERNORM_WAVE_KERNEL aps::launch_layernorm_backward
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
