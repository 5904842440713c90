// Encoder kernels for gfx950: the dense contractions of the transformer encoder on fp32 MFMA
// (exact f32 products / accumulation -- the 1e-4 parity bar rules out bf16) with the bias / ReLU /
// residual epilogues fused in, plus the row-wise LayerNorm, the sinusoid position add and the
// scaled-dot-product attention core.
//
// Replaces the torch ops behind aps/asr/transformer/impl.py:147-185, 377-429, 718-756,
// aps/asr/transformer/pose.py:29-118 and the Linear layers of aps/asr/base/encoder.py:367-441.
#include "common.h"

namespace aps {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------
// C[M, N] = act(A[M, K] . W[N, K]^T + bias[N]) + residual[M, N]
//
// 128 x 128 output tile per workgroup, 4 wavefronts in a 2 x 2 grid, each owning 64 x 64 =
// 2 x 2 v_mfma_f32_32x32x2_f32 tiles (64 accumulator registers).  K is consumed 16 at a time
// through LDS: both operands are K-contiguous in HBM (activations [M, K] and nn.Linear weights
// [N, K]), so a tile row is one 64-byte run; tiles are stored K-major in LDS ([16][128 + 4]) so the
// MFMA operand fetch (lane l: row l & 31 of k = l >> 5) is a conflict-free 128-byte ds_read.
// Double buffered: the global loads of step s+1 are issued before the MFMAs of step s.
// ------------------------------------------------------------------------------------------
constexpr int kTM = 128, kTN = 128, kTK = 16, kLdsPitch = kTM + 4;

struct GemmArgs {
  const float* A;
  const float* W;
  const float* bias;      // [N] or null
  const float* residual;  // [M, N] (ldc) or null
  float* C;
  int64_t M, N, K;
  int64_t lda, ldw, ldc;
  int32_t relu;
};

__device__ __forceinline__ float4 load_row4(const float* base, int64_t row, int64_t rows,
                                            int64_t ld, int64_t k, int64_t K) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (row < rows) {
    const float* p = base + row * ld + k;
    if (k + 3 < K) {
      v = *reinterpret_cast<const float4*>(p);
    } else {
      if (k + 0 < K) v.x = p[0];
      if (k + 1 < K) v.y = p[1];
      if (k + 2 < K) v.z = p[2];
    }
  }
  return v;
}

__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
  __shared__ float s_a[2][kTK][kLdsPitch];
  __shared__ float s_w[2][kTK][kLdsPitch];
  const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
  const int wm = wv >> 1, wn = wv & 1;
  const int64_t m0 = (int64_t)blockIdx.y * kTM, n0 = (int64_t)blockIdx.x * kTN;
  // staging role: thread -> (row r and r + 64, k quad kq) of both tiles
  const int r = tid >> 2, kq = (tid & 3) * 4;
  // A/W base alignment for float4: lda, ldw multiples of 4 are required by the launcher

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  float4 ra[2], rw[2];
  auto gload = [&](int64_t k0) {
    ra[0] = load_row4(g.A, m0 + r, g.M, g.lda, k0 + kq, g.K);
    ra[1] = load_row4(g.A, m0 + r + 64, g.M, g.lda, k0 + kq, g.K);
    rw[0] = load_row4(g.W, n0 + r, g.N, g.ldw, k0 + kq, g.K);
    rw[1] = load_row4(g.W, n0 + r + 64, g.N, g.ldw, k0 + kq, g.K);
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = r + 64 * h;
      s_a[buf][kq + 0][row] = ra[h].x;
      s_a[buf][kq + 1][row] = ra[h].y;
      s_a[buf][kq + 2][row] = ra[h].z;
      s_a[buf][kq + 3][row] = ra[h].w;
      s_w[buf][kq + 0][row] = rw[h].x;
      s_w[buf][kq + 1][row] = rw[h].y;
      s_w[buf][kq + 2][row] = rw[h].z;
      s_w[buf][kq + 3][row] = rw[h].w;
    }
  };

  const int64_t steps = (g.K + kTK - 1) / kTK;
  gload(0);
  sstore(0);
  __syncthreads();
  const int li = ln & 31, lk = ln >> 5;
  for (int64_t s = 0; s < steps; ++s) {
    const int buf = (int)(s & 1);
    if (s + 1 < steps) gload((s + 1) * kTK);
#pragma unroll
    for (int kk = 0; kk < kTK; kk += 2) {
      const float a0 = s_a[buf][kk + lk][wm * 64 + li];
      const float a1 = s_a[buf][kk + lk][wm * 64 + 32 + li];
      const float b0 = s_w[buf][kk + lk][wn * 64 + li];
      const float b1 = s_w[buf][kk + lk][wn * 64 + 32 + li];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (s + 1 < steps) {
      sstore(buf ^ 1);
    }
    __syncthreads();
  }

  // epilogue: C/D layout col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t col = n0 + wn * 64 + j * 32 + li;
      if (col >= g.N) continue;
      const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
        if (row >= g.M) continue;
        float v = acc[i][j][e] + bv;
        if (g.relu) v = fmaxf(v, 0.f);
        if (g.residual) v += g.residual[row * g.ldc + col];
        g.C[row * g.ldc + col] = v;
      }
    }
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

// ------------------------------------------------------------------------------------------
// Scaled dot-product attention core:
//   ctx[i, :] = softmax_j(q_i . k_j / sqrt(dh) + pad_mask_j) V        (impl.py:90-114)
// qkv: [N, T, 3, H, dh] (the fused in-projection output), ctx: [N, T, H, dh].
// Workgroup = (head, utterance, tile of 16 queries); a wavefront owns 4 queries whose running
// max / sum / context live in registers.  Keys are streamed in blocks of 128 through LDS (K and V
// of the head) with the usual rescaling, so any sequence length works; lanes run along the keys
// for the scores and the softmax (wave reductions), then along dh for the context.
// (VALU form: ~2 % of the encoder's flops; the GEMMs carry the MFMA work.)
// ------------------------------------------------------------------------------------------
constexpr int kAttKeys = 128;
constexpr int kAttQ = 4;  // queries per wavefront

template <int DH>
__global__ __launch_bounds__(256) void attention_core_kernel(const float* __restrict__ qkv,
                                                             const int64_t* __restrict__ lens,
                                                             float* __restrict__ ctx, int64_t T,
                                                             int H, float scale) {
  __shared__ float s_k[kAttKeys][DH + 1];
  __shared__ float s_v[kAttKeys][DH + 1];
  __shared__ float s_q[4][kAttQ][DH];
  __shared__ float s_p[4][kAttKeys];
  const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
  const int h = blockIdx.x;
  const int64_t n = blockIdx.y;
  const int64_t q0 = ((int64_t)blockIdx.z * 4 + wv) * kAttQ;
  const int64_t D3 = (int64_t)3 * H * DH;
  const float* base = qkv + n * T * D3 + (int64_t)h * DH;
  const int64_t len = lens ? min(T, max((int64_t)0, lens[n])) : T;
  constexpr int NV = (DH + 63) / 64;  // context elements per lane

  for (int qi = 0; qi < kAttQ; ++qi) {
    const int64_t i = q0 + qi;
    for (int d = ln; d < DH; d += 64) s_q[wv][qi][d] = (i < T) ? base[i * D3 + d] * scale : 0.f;
  }
  float run_max[kAttQ], run_sum[kAttQ], acc[kAttQ][NV];
#pragma unroll
  for (int qi = 0; qi < kAttQ; ++qi) {
    run_max[qi] = -INFINITY;
    run_sum[qi] = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[qi][v] = 0.f;
  }

  for (int64_t j0 = 0; j0 < T; j0 += kAttKeys) {
    __syncthreads();  // previous key block fully consumed (and s_q visible)
    const int64_t nk = min((int64_t)kAttKeys, T - j0);
    for (int64_t e = tid; e < nk * DH; e += 256) {
      const int64_t j = e / DH;
      const int d = (int)(e % DH);
      s_k[j][d] = base[(j0 + j) * D3 + (int64_t)H * DH + d];
      s_v[j][d] = base[(j0 + j) * D3 + (int64_t)2 * H * DH + d];
    }
    __syncthreads();
#pragma unroll
    for (int qi = 0; qi < kAttQ; ++qi) {
      if (q0 + qi >= T) break;  // wave-uniform
      float sc[kAttKeys / 64];
      float blk_max = -INFINITY;
#pragma unroll
      for (int c = 0; c < kAttKeys / 64; ++c) {
        const int64_t j = ln + 64 * c;
        float dot = -INFINITY;
        if (j < nk && j0 + j < len) {
          dot = 0.f;
#pragma unroll 8
          for (int d = 0; d < DH; ++d) dot += s_q[wv][qi][d] * s_k[j][d];
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
        for (int c = 0; c < kAttKeys / 64; ++c) {
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

}  // namespace aps

using namespace aps;

extern "C" int aps_linear(const float* A, const float* W, const float* bias, const float* residual,
                          float* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
                          int64_t ldc, int32_t relu, void* stream) {
  APS_CHECK_ARG(A && W && C && M > 0 && N > 0 && K > 0);
  APS_CHECK_ARG(lda >= K && ldw >= K && ldc >= N);
  // 16-byte aligned row starts for the float4 tile loads
  APS_CHECK_ARG(lda % 4 == 0 && ldw % 4 == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0);
  GemmArgs g{A, W, bias, residual, C, M, N, K, lda, ldw, ldc, relu};
  dim3 grid((unsigned)((N + kTN - 1) / kTN), (unsigned)((M + kTM - 1) / kTM));
  APS_CHECK_ARG(grid.y <= 65535);
  hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), g);
  return aps_launch_status();
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

extern "C" int aps_attention_core(const float* qkv, const int64_t* lens, float* ctx, int64_t N,
                                  int64_t T, int64_t H, int64_t head_dim, void* stream) {
  APS_CHECK_ARG(qkv && ctx && N > 0 && N <= 65535 && T > 0 && H > 0 && H <= 65535);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const float scale = 1.0f / sqrtf((float)head_dim);
  dim3 grid((unsigned)H, (unsigned)N, (unsigned)((T + 4 * kAttQ - 1) / (4 * kAttQ)));
  switch (head_dim) {
    case 32:
      hipLaunchKernelGGL((attention_core_kernel<32>), grid, dim3(256), 0, st, qkv, lens, ctx, T,
                         (int)H, scale);
      break;
    case 64:
      hipLaunchKernelGGL((attention_core_kernel<64>), grid, dim3(256), 0, st, qkv, lens, ctx, T,
                         (int)H, scale);
      break;
    case 128:
      hipLaunchKernelGGL((attention_core_kernel<128>), grid, dim3(256), 0, st, qkv, lens, ctx, T,
                         (int)H, scale);
      break;
    default:
      return APS_ERR_UNSUPPORTED;
  }
  return aps_launch_status();
}
