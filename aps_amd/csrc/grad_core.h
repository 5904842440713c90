// Backward (adjoint) arithmetic of the hot path -- section 8(f) row 1 of the scope table: what
// `loss.backward()` needs under cmd/train_ss.py / cmd/train_am.py (aps/trainer/ddp.py:124-200) for
// the mask-based MVDR front end (aps/asr/filter/mvdr.py:19-174), its RNN mask estimator
// (aps/asr/base/encoder.py:87-184) and the conformer encoder (aps/asr/transformer/impl.py:225-541).
//
// Every operation is an index functor: `op(idx)` does all the work of output element / row `idx`
// from plain pointers.  grad.hip runs them one GPU thread per index; tests/csrc/grad_host.cc runs the
// SAME functors in a host loop, so the adjoint arithmetic and every piece of index math is checked
// against torch autograd through the CPU oracle without a GPU.  The heavy contractions of the
// backward pass (dX = dY W, dW = dY^T X) are launches of the forward's fp32 MFMA GEMM on transposed
// operands (grad.hip), not functors.
//
// Conventions: complex values are (re, im) float pairs; the gradient of a real loss with respect to
// a complex value z is stored the same way, G = dL/d(re z) + i dL/d(im z).  Then for z = a b:
// G_a = G_z conj(b); for z = conj(a) b: G_a = conj(G_z) b, G_b = G_z a; for Z = A B (matrices):
// G_A = G_Z B^H, G_B = A^H G_Z; for Z = A^-1: G_A = -A^-H G_Z A^-H.
#ifndef APS_AMD_GRAD_CORE_H_
#define APS_AMD_GRAD_CORE_H_

#include <math.h>
#include <stdint.h>

#include "fft_core.h"  // APS_HD, cf and its algebra

namespace aps {
namespace grad {

constexpr float kEps = 1.1920928955078125e-07f;  // aps/const.py:17 EPSILON

APS_HD float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// activation codes of aps_linear: 0 none, 1 relu, 2 swish, 3 sigmoid, 4 tanh, 5 gelu (erf form);
// 6 (the stand-alone pass only): nn.LeakyReLU() with its default slope 0.01 (the DCCRN blocks)
APS_HD float act_value(float x, int act) {
  switch (act) {
    case 1: return x > 0.f ? x : 0.f;
    case 2: return x * sigmoidf_(x);
    case 3: return sigmoidf_(x);
    case 4: return tanhf(x);
    case 5: return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
    case 6: return x > 0.f ? x : 0.01f * x;
    case 7: return x * x;
    default: return x;
  }
}
APS_HD float act_slope(float x, int act) {  // d act / dx at the pre-activation x
  switch (act) {
    case 1: return x > 0.f ? 1.f : 0.f;
    case 2: {
      const float s = sigmoidf_(x);
      return s * (1.0f + x * (1.0f - s));
    }
    case 3: {
      const float s = sigmoidf_(x);
      return s * (1.0f - s);
    }
    case 4: {
      const float t = tanhf(x);
      return 1.0f - t * t;
    }
    case 5:
      return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) +
             x * 0.39894228040143268f * expf(-0.5f * x * x);
    case 6: return x > 0.f ? 1.f : 0.01f;
    case 7: return 2.f * x;
    default: return 1.f;
  }
}

// out = act(pre) * alpha (+ residual)          (the GEMM epilogue as its own pass: training keeps
// the pre-activation for the backward)
struct ActForward {
  const float* pre;
  const float* residual;  // or null
  float* out;
  int act;
  float alpha;
  APS_HD void operator()(int64_t i) const {
    float v = act_value(pre[i], act) * alpha;
    if (residual) v += residual[i];
    out[i] = v;
  }
};

// g_pre = g_out * alpha * act'(pre)
struct ActBackward {
  const float* g_out;
  const float* pre;
  float* g_pre;
  int act;
  float alpha;
  APS_HD void operator()(int64_t i) const { g_pre[i] = g_out[i] * alpha * act_slope(pre[i], act); }
};

// ----------------------------------------------------------------------------------------------
// FixedBeamformer (aps/transform/enh.py:349-384): b = sum_c conj(w[beam, c, f]) x[n, c, f, t], all B beams or
// the one beam[n] selects.  x real / imag [N, C, F, T], w real / imag [B, C, F], g_b real / imag [N, (B), F, T].
//   input:  g_x[n,c,f,t] = sum_b w[b,c,f] g_b[n,b,f,t]          (one (n, c, f, t) per index)
//   weight: g_w[b,c,f]   = sum_{n,t} conj(g_b[n,b,f,t]) x[n,c,f,t], i.e.
//           g_wr = sum g_br x_r + g_bi x_i,  g_wi = sum g_br x_i - g_bi x_r   (one (b, c, f) per index)
// ----------------------------------------------------------------------------------------------
struct FixedBeamBackwardInput {
  const float* g_br;
  const float* g_bi;
  const float* wr;
  const float* wi;
  const int64_t* beam;  // [N] or null (all beams)
  float* g_xr;
  float* g_xi;
  int64_t C, F, T, B;
  APS_HD void operator()(int64_t idx) const {
    const int64_t t = idx % T, f = (idx / T) % F, c = (idx / (T * F)) % C, n = idx / (T * F * C);
    float re = 0.f, im = 0.f;
    if (beam) {
      const int64_t b = beam[n];
      const float gr = g_br[(n * F + f) * T + t], gi = g_bi[(n * F + f) * T + t];
      const float a = wr[(b * C + c) * F + f], d = wi[(b * C + c) * F + f];
      re = gr * a - gi * d;
      im = gr * d + gi * a;
    } else {
      for (int64_t b = 0; b < B; ++b) {
        const float gr = g_br[((n * B + b) * F + f) * T + t], gi = g_bi[((n * B + b) * F + f) * T + t];
        const float a = wr[(b * C + c) * F + f], d = wi[(b * C + c) * F + f];
        re += gr * a - gi * d;
        im += gr * d + gi * a;
      }
    }
    g_xr[idx] = re, g_xi[idx] = im;
  }
};
struct FixedBeamBackwardWeight {
  const float* g_br;
  const float* g_bi;
  const float* xr;
  const float* xi;
  const int64_t* beam;
  float* g_wr;
  float* g_wi;
  int64_t N, C, F, T, B;
  APS_HD void operator()(int64_t idx) const {
    const int64_t f = idx % F, c = (idx / F) % C, b = idx / (F * C);
    float re = 0.f, im = 0.f;
    for (int64_t n = 0; n < N; ++n) {
      if (beam && beam[n] != b) continue;
      const float* gr = beam ? g_br + (n * F + f) * T : g_br + ((n * B + b) * F + f) * T;
      const float* gi = beam ? g_bi + (n * F + f) * T : g_bi + ((n * B + b) * F + f) * T;
      const float* pr = xr + ((n * C + c) * F + f) * T;
      const float* pi = xi + ((n * C + c) * F + f) * T;
      for (int64_t t = 0; t < T; ++t) {
        re += gr[t] * pr[t] + gi[t] * pi[t];
        im += gr[t] * pi[t] - gi[t] * pr[t];
      }
    }
    g_wr[idx] = re, g_wi[idx] = im;
  }
};

// ----------------------------------------------------------------------------------------------
// One time step of nn.GRU / nn.RNN(tanh | relu) / nn.LSTM backwards (the step-by-step recurrences of
// var_len_rnn_forward, aps/asr/base/component.py:26-55, that have no persistent kernel; forward:
// rnn_step_kernel in decoder.hip, same modes and gate orders).  One (utterance, unit) per index.
//   gx = x_t W_ih^T + b_ih (row pitch ldx), gh = h_{t-1} W_hh^T + b_hh (recomputed by the caller),
//   g_y = gradient of the step's OUTPUT (or null), g_h = gradient of the state carried back from t + 1
//   -> g_gx, g_gh (this step's rows of the two pre-activation gradients: the batched weight-gradient
//      products consume them), g_hp = the part of the previous state's gradient that does NOT go through
//      W_hh (the caller adds g_gh W_hh), g_cp (LSTM)
// Packed-sequence semantics: a row with t >= len kept its state and emitted zeros, so its state gradient
// passes through untouched and its pre-activation gradients are zero.
// ----------------------------------------------------------------------------------------------
struct RnnStepBackward {
  const float* gx;
  int64_t ldx;
  const float* gh;       // [N, G H]
  const float* h_prev;   // [N, H] or null (t = 0: zeros)
  const float* c_prev;   // LSTM: [N, H] or null
  const int64_t* lens;   // [N] or null
  int64_t t;
  const float* g_y;      // [N, ldgy] rows of this step, or null
  int64_t ldgy;
  const float* g_h;      // [N, H] or null (the last step)
  const float* g_c;      // LSTM: [N, H] or null
  float* g_gx;           // rows of this step, pitch ldg
  float* g_gh;           // rows of this step, pitch ldg
  int64_t ldg;
  float* g_hp;           // [N, H]
  float* g_cp;           // LSTM: [N, H]
  int H, mode;
  APS_HD void operator()(int64_t i) const {
    const int G = mode == 0 ? 3 : (mode == 3 ? 4 : 1);
    const int64_t n = i / H;
    const int u = (int)(i % H);
    const bool live = !lens || t < lens[n];
    const float carried = g_h ? g_h[i] : 0.f;
    float* ox = g_gx + n * ldg + u;
    float* oh = g_gh + n * ldg + u;
    if (!live) {
      for (int q = 0; q < G; ++q) ox[q * H] = oh[q * H] = 0.f;
      g_hp[i] = carried;
      if (mode == 3) g_cp[i] = g_c ? g_c[i] : 0.f;
      return;
    }
    const float g = carried + (g_y ? g_y[n * ldgy + u] : 0.f);
    const float hp = h_prev ? h_prev[i] : 0.f;
    const float* px = gx + n * ldx + u;
    const float* ph = gh + n * (int64_t)G * H + u;
    if (mode == 0) {  // r | z | n;  h = (1 - z) n + z h'
      const float r = sigmoidf_(px[0] + ph[0]), z = sigmoidf_(px[H] + ph[H]);
      const float hn = ph[2 * H];
      const float nn_ = tanhf(px[2 * H] + r * hn);
      const float g_n = g * (1.0f - z) * (1.0f - nn_ * nn_);   // pre-activation of n
      const float g_z = g * (hp - nn_) * z * (1.0f - z);        // pre-activation of z
      const float g_r = g_n * hn * r * (1.0f - r);              // pre-activation of r
      ox[0] = g_r, ox[H] = g_z, ox[2 * H] = g_n;
      oh[0] = g_r, oh[H] = g_z, oh[2 * H] = g_n * r;
      g_hp[i] = g * z;
    } else if (mode == 3) {  // i | f | g | o;  c = f c' + i g,  h = o tanh(c)
      const float cp = c_prev ? c_prev[i] : 0.f;
      const float gi = sigmoidf_(px[0] + ph[0]), gf = sigmoidf_(px[H] + ph[H]);
      const float gg = tanhf(px[2 * H] + ph[2 * H]), go = sigmoidf_(px[3 * H] + ph[3 * H]);
      const float c = gf * cp + gi * gg, tc = tanhf(c);
      const float gc = (g_c ? g_c[i] : 0.f) + g * go * (1.0f - tc * tc);
      const float a_i = gc * gg * gi * (1.0f - gi), a_f = gc * cp * gf * (1.0f - gf);
      const float a_g = gc * gi * (1.0f - gg * gg), a_o = g * tc * go * (1.0f - go);
      ox[0] = oh[0] = a_i, ox[H] = oh[H] = a_f, ox[2 * H] = oh[2 * H] = a_g, ox[3 * H] = oh[3 * H] = a_o;
      g_hp[i] = 0.f;
      g_cp[i] = gc * gf;
    } else {
      const float v = px[0] + ph[0];
      float a;
      if (mode == 1) {
        const float th_ = tanhf(v);
        a = g * (1.0f - th_ * th_);
      } else {
        a = v > 0.f ? g : 0.f;
      }
      ox[0] = oh[0] = a;
      g_hp[i] = 0.f;
    }
  }
};

// ----------------------------------------------------------------------------------------------
// dropout (nn.Dropout in train() mode): counter-based -- the keep decision of element `idx` is a
// hash of (seed, idx), so the backward recomputes the mask instead of storing it and the forward of
// an attention row can draw the mask of its weights on the fly.  The stream differs from torch's
// Philox (no parity target exists for a random layer); the seed is drawn from torch's CPU generator,
// so torch.manual_seed makes a run reproducible.
// ----------------------------------------------------------------------------------------------
APS_HD uint32_t mix32(uint64_t x) {  // murmur3 finaliser
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return (uint32_t)x;
}
// 0 (dropped) or 1 / (1 - p) (kept); p = 0 -> 1
APS_HD float keep_scale(uint64_t seed, uint64_t idx, float p) {
  if (!(p > 0.f)) return 1.f;
  const float u = (float)(mix32(seed ^ (idx * 0x9E3779B97F4A7C15ULL)) >> 8) * (1.0f / 16777216.0f);
  return u >= p ? 1.0f / (1.0f - p) : 0.f;
}
// out = x * keep_scale: forward on x, backward on g (the same mask)
struct Dropout {
  const float* x;
  float* out;
  uint64_t seed;
  float p;
  APS_HD void operator()(int64_t i) const { out[i] = x[i] * keep_scale(seed, (uint64_t)i, p); }
};

// out[r, d] = x[r, d] + b[d]   (the bias of a convolution in front of a training-mode BatchNorm)
struct RowBiasAdd {
  const float* x;
  const float* b;
  float* out;
  int64_t D;
  APS_HD void operator()(int64_t i) const { out[i] = x[i] + b[i % D]; }
};

// rows of an embedding table gathered by index (RelPosEncoding, pose.py:65-88: table = E[clamp(j - i)])
// and the adjoint: g_weight[v, d] = sum over the rows r with index[r] == v of g_table[r, d]
struct GatherRowsBackward {
  const int64_t* index;  // [R]
  const float* g_table;  // [R, D]
  float* g_weight;       // [V, D]
  int64_t R, D;
  APS_HD void operator()(int64_t i) const {
    const int64_t d = i % D, v = i / D;
    float acc = 0.f;
    for (int64_t r = 0; r < R; ++r)
      if (index[r] == v) acc += g_table[r * D + d];
    g_weight[i] = acc;
  }
};

// ----------------------------------------------------------------------------------------------
// column reductions over the rows of a [rows, cols] matrix (row pitch ld), two deterministic
// stages: partial[chunk, c] over `rows_per_chunk` rows, then the sum over chunks.
//   mode 0: sum_r A          mode 1: sum_r A * B           mode 2: sum_r (A - v1[c])^2
//   mode 3: sum_r A * (B - v1[c]) * v2[c]   (BatchNorm: sum g_y * xhat)
// ----------------------------------------------------------------------------------------------
struct ColReducePartial {
  const float* A;
  const float* B;
  const float* v1;
  const float* v2;
  float* partial;  // [chunks, cols]
  int64_t rows, cols, lda, ldb, rows_per_chunk;
  int mode;
  APS_HD void operator()(int64_t idx) const {
    const int64_t c = idx % cols, chunk = idx / cols;
    const int64_t r0 = chunk * rows_per_chunk;
    const int64_t r1 = r0 + rows_per_chunk < rows ? r0 + rows_per_chunk : rows;
    float acc = 0.f;
    const float m = (mode == 2 || mode == 3) ? v1[c] : 0.f;
    const float s = (mode == 3) ? v2[c] : 0.f;
    if (mode == 4) {
      // two plain column sums in one launch: columns [0, cols / 2) of A, then those of B (e.g. the
      // LayerNorm's g_gamma | g_beta from t and g_y)
      const int64_t half = cols / 2;
      const float* src = c < half ? A + c : B + (c - half);
      const int64_t ld = c < half ? lda : ldb;
      int64_t r = r0;
      for (; r + 8 <= r1; r += 8) {
        float a[8];
        for (int u = 0; u < 8; ++u) a[u] = src[(r + u) * ld];
        for (int u = 0; u < 8; ++u) acc += a[u];
      }
      for (; r < r1; ++r) acc += src[r * ld];
      partial[idx] = acc;
      return;
    }
    const bool two = mode == 1 || mode == 3;
    auto add = [&](float a, float b) {
      if (mode == 0) {
        acc += a;
      } else if (mode == 1) {
        acc += a * b;
      } else if (mode == 2) {
        acc += (a - m) * (a - m);
      } else {
        acc += a * (b - m) * s;
      }
    };
    int64_t r = r0;
    for (; r + 8 <= r1; r += 8) {  // 8 rows requested together, summed in row order
      float a[8], b[8];
      for (int u = 0; u < 8; ++u) {
        a[u] = A[(r + u) * lda + c];
        b[u] = two ? B[(r + u) * ldb + c] : 0.f;
      }
      for (int u = 0; u < 8; ++u) add(a[u], b[u]);
    }
    for (; r < r1; ++r) add(A[r * lda + c], two ? B[r * ldb + c] : 0.f);
    partial[idx] = acc;
  }
};
struct ColReduceFinal {
  const float* partial;
  float* out;  // [cols]
  int64_t cols, chunks;
  float scale;
  int accumulate;  // 1: out += (gradient accumulation into an existing buffer)
  APS_HD void operator()(int64_t c) const {
    // 8 partial sums side by side: 8 loads in flight instead of a chain of `chunks` round trips
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int64_t k = 0;
    for (; k + 8 <= chunks; k += 8)
      for (int u = 0; u < 8; ++u) a[u] += partial[(k + u) * cols + c];
    for (; k < chunks; ++k) a[0] += partial[k * cols + c];
    const float acc = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    out[c] = (accumulate ? out[c] : 0.f) + acc * scale;
  }
};

// ----------------------------------------------------------------------------------------------
// LayerNorm backward, one row per index:  y = (s - mu) rstd gamma + beta,  s = x (+ residual)
//   g_s = rstd (g_y gamma - mean(g_y gamma) - xhat mean(g_y gamma xhat));   t = g_y xhat
// (g_gamma = colsum(t), g_beta = colsum(g_y): column reductions above)
// ----------------------------------------------------------------------------------------------
struct LayerNormBackward {
  const float* x;
  const float* residual;  // or null
  const float* gamma;     // or null (= 1)
  const float* g_y;
  float* g_x;  // also the gradient of the residual
  float* t;    // [rows, D] g_y * xhat, or null
  int64_t D;
  float eps;
  APS_HD void operator()(int64_t r) const {
    const float* xr = x + r * D;
    const float* rr = residual ? residual + r * D : nullptr;
    const float* gr = g_y + r * D;
    float mu = 0.f;
    for (int64_t d = 0; d < D; ++d) mu += xr[d] + (rr ? rr[d] : 0.f);
    mu /= (float)D;
    float var = 0.f;
    for (int64_t d = 0; d < D; ++d) {
      const float c = xr[d] + (rr ? rr[d] : 0.f) - mu;
      var += c * c;
    }
    const float rstd = 1.0f / sqrtf(var / (float)D + eps);
    float m1 = 0.f, m2 = 0.f;
    for (int64_t d = 0; d < D; ++d) {
      const float xh = (xr[d] + (rr ? rr[d] : 0.f) - mu) * rstd;
      const float gh = gr[d] * (gamma ? gamma[d] : 1.f);
      m1 += gh;
      m2 += gh * xh;
    }
    m1 /= (float)D;
    m2 /= (float)D;
    for (int64_t d = 0; d < D; ++d) {
      const float xh = (xr[d] + (rr ? rr[d] : 0.f) - mu) * rstd;
      const float gh = gr[d] * (gamma ? gamma[d] : 1.f);
      g_x[r * D + d] = rstd * (gh - m1 - xh * m2);
      if (t) t[r * D + d] = gr[d] * xh;
    }
  }
};

// ----------------------------------------------------------------------------------------------
// BatchNorm over the rows of [rows, D] (BatchNorm1d on N x T x D / BatchNorm2d on channels-last
// N x H x W x C, training mode: batch statistics; impl.py:478-489, component.py:251-307)
// ----------------------------------------------------------------------------------------------
// statistics from the column sums: mean, rstd, and torch's running-statistics update
struct BatchNormStats {
  const float* sum;    // [D] sum_r x
  const float* sumsq;  // [D] sum_r (x - mean)^2   (second pass)
  float* mean;
  float* rstd;
  float* running_mean;  // or null
  float* running_var;   // or null
  int64_t rows;
  float eps, momentum;
  int stage;  // 0: mean only (before the centred second pass); 1: rstd + running statistics
  APS_HD void operator()(int64_t d) const {
    if (stage == 0) {
      mean[d] = sum[d] / (float)rows;
      return;
    }
    const float var = sumsq[d] / (float)rows;  // biased: normalisation
    rstd[d] = 1.0f / sqrtf(var + eps);
    if (running_mean) running_mean[d] = (1.f - momentum) * running_mean[d] + momentum * mean[d];
    if (running_var) {
      const float unbiased = rows > 1 ? sumsq[d] / (float)(rows - 1) : var;
      running_var[d] = (1.f - momentum) * running_var[d] + momentum * unbiased;
    }
  }
};
// y = (x - mean) rstd gamma + beta
struct BatchNormApply {
  const float* x;
  const float* mean;
  const float* rstd;
  const float* gamma;  // or null
  const float* beta;   // or null
  float* y;
  int64_t D;
  APS_HD void operator()(int64_t i) const {
    const int64_t d = i % D;
    float v = (x[i] - mean[d]) * rstd[d];
    if (gamma) v *= gamma[d];
    if (beta) v += beta[d];
    y[i] = v;
  }
};
// g_x = gamma rstd (g_y - sum_gy / M - xhat sum_gy_xhat / M)    (batch statistics are functions of x)
// eval-mode statistics (constants): g_x = gamma rstd g_y  (sum_gy = null)
struct BatchNormBackward {
  const float* x;
  const float* mean;
  const float* rstd;
  const float* gamma;       // or null
  const float* g_y;
  const float* sum_gy;      // [D] or null
  const float* sum_gy_xhat; // [D] or null
  float* g_x;
  int64_t D, rows;
  APS_HD void operator()(int64_t i) const {
    const int64_t d = i % D;
    const float xh = (x[i] - mean[d]) * rstd[d];
    float g = g_y[i];
    if (sum_gy) g -= (sum_gy[d] + xh * sum_gy_xhat[d]) / (float)rows;
    g_x[i] = g * rstd[d] * (gamma ? gamma[d] : 1.f);
  }
};

// ----------------------------------------------------------------------------------------------
// softmax over short rows (the channel softmax of ChannelAttention, mvdr.py:174)
// ----------------------------------------------------------------------------------------------
struct SoftmaxRows {
  const float* x;
  float* y;
  int64_t D;
  APS_HD void operator()(int64_t r) const {
    float mx = -INFINITY;
    for (int64_t d = 0; d < D; ++d) mx = fmaxf(mx, x[r * D + d]);
    float den = 0.f;
    for (int64_t d = 0; d < D; ++d) den += expf(x[r * D + d] - mx);
    for (int64_t d = 0; d < D; ++d) y[r * D + d] = expf(x[r * D + d] - mx) / den;
  }
};
struct SoftmaxRowsBackward {
  const float* y;
  const float* g_y;
  float* g_x;
  int64_t D;
  APS_HD void operator()(int64_t r) const {
    float dot = 0.f;
    for (int64_t d = 0; d < D; ++d) dot += y[r * D + d] * g_y[r * D + d];
    for (int64_t d = 0; d < D; ++d) g_x[r * D + d] = y[r * D + d] * (g_y[r * D + d] - dot);
  }
};

// ----------------------------------------------------------------------------------------------
// |z + eps| of interleaved complex values and its adjoint (AbsTransform on a ComplexTensor,
// asr.py:306-332: eps joins the REAL part)
// ----------------------------------------------------------------------------------------------
struct MagnitudeForward {
  const float* z;  // [n, 2]
  float* mag;      // [n]
  float eps;
  APS_HD void operator()(int64_t i) const {
    const float re = z[2 * i] + eps, im = z[2 * i + 1];
    mag[i] = sqrtf(re * re + im * im);
  }
};
struct MagnitudeBackward {
  const float* z;      // [n, 2]
  const float* g_mag;  // [n]
  float* g_z;          // [n, 2]
  float eps;
  APS_HD void operator()(int64_t i) const {
    const float re = z[2 * i] + eps, im = z[2 * i + 1];
    const float m = sqrtf(re * re + im * im);
    const float s = m > 0.f ? g_mag[i] / m : 0.f;
    g_z[2 * i] = s * re;
    g_z[2 * i + 1] = s * im;
  }
};

// log(clamp(m, eps)) (or log(lb + m)) followed by the per-row CMVN of cmvn(per_band) (asr.py:431-464,
// 576-618): l = log(.), c = l - mean(l) (norm_mean), z = c / sqrt(mean(c^2) + eps) (norm_var; with
// norm_mean off the variance is var(l, unbiased=False)).  One row per index.
struct LogCmvnBackward {
  const float* m;    // [rows, D] input of the log
  const float* g_z;  // [rows, D]
  float* g_m;
  int64_t D;
  float log_eps, lower_bound, cmvn_eps;
  int apply_log, norm_mean, norm_var;
  APS_HD float logv(float v) const {
    if (!apply_log) return v;
    return lower_bound > 0.f ? logf(lower_bound + v) : logf(v < log_eps ? log_eps : v);
  }
  APS_HD void operator()(int64_t r) const {
    const float* mr = m + r * D;
    const float* gr = g_z + r * D;
    float mu = 0.f;
    for (int64_t d = 0; d < D; ++d) mu += logv(mr[d]);
    mu /= (float)D;
    const float shift = norm_mean ? mu : 0.f;
    // variance of the values that get divided: centred (norm_mean) or var(l) about its own mean
    float var = 0.f;
    for (int64_t d = 0; d < D; ++d) {
      const float c = logv(mr[d]) - mu;
      var += c * c;
    }
    var /= (float)D;
    const float rstd = norm_var ? 1.0f / sqrtf(var + cmvn_eps) : 1.f;
    // z = (l - shift) rstd;  g_l = rstd g_z - [norm_mean] mean(rstd g_z) - [norm_var] (l - mu) rstd^3 mean(g_z (l - shift))
    float a = 0.f, b = 0.f;
    for (int64_t d = 0; d < D; ++d) {
      const float l = logv(mr[d]);
      a += gr[d];
      b += gr[d] * (l - shift);
    }
    a /= (float)D;
    b /= (float)D;
    for (int64_t d = 0; d < D; ++d) {
      const float l = logv(mr[d]);
      float gl = rstd * gr[d];
      if (norm_mean) gl -= rstd * a;
      if (norm_var) gl -= (l - mu) * rstd * rstd * rstd * b;
      float gm = gl;
      if (apply_log) {
        if (lower_bound > 0.f)
          gm = gl / (lower_bound + mr[d]);
        else
          gm = mr[d] < log_eps ? 0.f : gl / mr[d];  // clamp passes no gradient below eps
      }
      g_m[r * D + d] = gm;
    }
  }
};

// ----------------------------------------------------------------------------------------------
// GLU -> depthwise Conv1d over time (the conformer convolution between its pointwise layers,
// impl.py:478-489):  g = a sigmoid(b), (a | b) = x[..., :D | D:],  c[n,t,d] = bias[d] +
// sum_k w[d,k] g[n, t + k - pad, d],  pad = (K - 1) / 2, zero outside [0, T)
// ----------------------------------------------------------------------------------------------
struct GluDwconvBackwardInput {  // index = (n, t, d): g_x[n,t,d] and g_x[n,t,D+d]
  const float* x;    // [N, T, 2D]
  const float* w;    // [D, K]
  const float* g_c;  // [N, T, D]
  float* g_x;        // [N, T, 2D]
  int64_t T, D, K;
  int64_t pad;       // left context: (K - 1) / 2, or K - 1 for the causal form
  APS_HD void operator()(int64_t idx) const {
    const int64_t d = idx % D, t = (idx / D) % T, n = idx / (D * T);
    float gg = 0.f;  // g_g[n,t,d] = sum_k w[d,k] g_c[n, t - k + pad, d]
    for (int64_t k = 0; k < K; ++k) {
      const int64_t tt = t - k + pad;
      if (tt >= 0 && tt < T) gg += w[d * K + k] * g_c[(n * T + tt) * D + d];
    }
    const float a = x[(n * T + t) * 2 * D + d], b = x[(n * T + t) * 2 * D + D + d];
    const float s = sigmoidf_(b);
    g_x[(n * T + t) * 2 * D + d] = gg * s;
    g_x[(n * T + t) * 2 * D + D + d] = gg * a * s * (1.f - s);
  }
};
// index = (chunk, k, d) with the channel fastest (lanes along d: coalesced rows of g_c and x; round 4 -- with the
// tap fastest a wave touched 4 - 5 channels and took 55 us per call): partial[chunk, d, k] over a range of (n, t) rows
struct GluDwconvBackwardWeight {
  const float* x;
  const float* g_c;
  float* partial;  // [chunks, D, K]
  int64_t N, T, D, K, rows_per_chunk;
  int64_t pad;            // (see GluDwconvBackwardInput)
  const float* pad_bias;  // causal form: [2D], the frames in front of the sequence carry glu(pad_bias); or null (zeros)
  APS_HD void operator()(int64_t idx) const {
    const int64_t d = idx % D, k = (idx / D) % K, chunk = idx / (K * D);
    const int64_t r0 = chunk * rows_per_chunk;
    const int64_t r1 = r0 + rows_per_chunk < N * T ? r0 + rows_per_chunk : N * T;
    const float g0 = pad_bias ? pad_bias[d] * sigmoidf_(pad_bias[D + d]) : 0.f;
    float acc = 0.f;
    for (int64_t r = r0; r < r1; ++r) {
      const int64_t n = r / T, t = r % T;
      const int64_t tt = t + k - pad;
      if (tt < 0) {
        acc += g_c[r * D + d] * g0;
        continue;
      }
      if (tt >= T) continue;
      const float a = x[(n * T + tt) * 2 * D + d], b = x[(n * T + tt) * 2 * D + D + d];
      acc += g_c[r * D + d] * a * sigmoidf_(b);
    }
    partial[(chunk * D + d) * K + k] = acc;
  }
};
// causal form: gradient of the [2D] vector whose GLU fills the K - 1 frames in front of the sequence
// (the pointwise projection's bias, impl.py:491-505), index = channel d
struct GluDwconvBackwardPad {
  const float* w;         // [D, K]
  const float* g_c;       // [N, T, D]
  const float* pad_bias;  // [2D]
  float* g_pad;           // [2D]
  int64_t N, T, D, K, pad;
  APS_HD void operator()(int64_t d) const {
    float g0 = 0.f;  // d loss / d glu(pad_bias)[d]
    for (int64_t n = 0; n < N; ++n)
      for (int64_t t = 0; t < T && t < pad; ++t)
        for (int64_t k = 0; k + t < pad && k < K; ++k) g0 += g_c[(n * T + t) * D + d] * w[d * K + k];
    const float a = pad_bias[d], s = sigmoidf_(pad_bias[D + d]);
    g_pad[d] = g0 * s;
    g_pad[D + d] = g0 * a * s * (1.f - s);
  }
};

// ----------------------------------------------------------------------------------------------
// im2col of a channels-last image: patches[m, (kh, kw, ci)] for output pixel m = (n, ho, wo)
// (dW of Conv2d = dY^T patches, one GEMM)
// ----------------------------------------------------------------------------------------------
struct Im2Col {
  const float* x;  // [N, H, W, Ci]
  float* out;      // [N Ho Wo, ld] (columns >= KH KW Ci are zero padding)
  int64_t H, W, Ci, KH, KW, sh, sw, ph, pw, Ho, Wo, ld;
  APS_HD void operator()(int64_t idx) const {
    const int64_t col = idx % ld, m = idx / ld;
    float v = 0.f;
    if (col < KH * KW * Ci) {
      const int64_t ci = col % Ci, kw = (col / Ci) % KW, kh = col / (Ci * KW);
      const int64_t wo = m % Wo, ho = (m / Wo) % Ho, n = m / (Wo * Ho);
      const int64_t hi = ho * sh + kh - ph, wi = wo * sw + kw - pw;
      if (hi >= 0 && hi < H && wi >= 0 && wi < W) v = x[((n * H + hi) * W + wi) * Ci + ci];
    }
    out[idx] = v;
  }
};

// ----------------------------------------------------------------------------------------------
// multi-head self attention with the learnt relative position term (RelMultiheadAttention,
// impl.py:225-296): S[i,j] = scale (q_i . k_j + q_i . rel[j - i + zero]) (rows of rel outside the
// table count as zero), keys j >= len masked, P = softmax_j S, ctx_i = sum_j P[i,j] v_j.
// qkv [N, T, 3, H, dh].  Three passes, all recomputing P from q, k (nothing T x T is stored):
//   rows    (n, h, i): row statistics (max, sum, D_i = sum_j P dP) and g_q[i]
//   columns (n, h, j): g_k[j], g_v[j]
//   table   (n, h, r): partial g_rel[r] of this (n, h)   (summed over (n, h) by a column reduction)
// ----------------------------------------------------------------------------------------------
struct AttentionGeometry {
  const float* qkv;
  const int64_t* lens;  // or null
  const float* rel;     // [rel_len, dh] (shared) or [H, rel_len, dh] (rel_head_stride = rel_len dh) or null
  const float* g_ctx;   // [N, T, H, dh]
  int64_t T, H, dh, rel_zero, rel_len;
  float scale;
  float drop_p;        // dropout of the attention WEIGHTS (impl.py:104), 0 = none
  uint64_t drop_seed;
  // context window of prep_context_mask (transformer/utils.py:60-98; chunk 1, lctx = rctx = -1: none)
  int64_t chunk = 1, lctx = -1, rctx = -1;
  // Transformer-XL form (XlMultiheadAttention.dot_att, impl.py:322-374): per-head table, the biases
  // u / v [H, dh] (or null) and the query taken from slot `qslot` of qkv (2 = the value projection,
  // the reference's quirk)
  int64_t rel_head_stride = 0;
  const float* rel_u = nullptr;
  const float* rel_v = nullptr;
  int64_t qslot = 0;
  // any additive mask [T, T] on the scaled logits (0 / -inf or a bias: the `src_mask` / `tgt_mask` tensors of
  // impl.py:104-114, decoder.py:150-186), or null.  Data, not a parameter: no gradient flows into it; a -inf
  // entry is a masked pair (its weight and its gradient are exactly 0), a row of -inf a row without keys.
  const float* add_mask = nullptr;
  // keep factor of weight (n, h, i, j)
  APS_HD float keep(int64_t n, int64_t h, int64_t i, int64_t j) const {
    return keep_scale(drop_seed, (uint64_t)(((n * H + h) * T + i) * T + j), drop_p);
  }
  APS_HD const float* q(int64_t n, int64_t t, int64_t h) const {
    return qkv + ((n * T + t) * 3 * H + h) * dh;
  }
  APS_HD const float* k(int64_t n, int64_t t, int64_t h) const { return q(n, t, h) + H * dh; }
  APS_HD const float* v(int64_t n, int64_t t, int64_t h) const { return q(n, t, h) + 2 * H * dh; }
  // the row the scores' "query" is read from (slot 0, or the value projection)
  APS_HD const float* qsrc(int64_t n, int64_t t, int64_t h) const { return q(n, t, h) + qslot * H * dh; }
  APS_HD const float* g(int64_t n, int64_t t, int64_t h) const {
    return g_ctx + ((n * T + t) * H + h) * dh;
  }
  APS_HD int64_t keys(int64_t n) const {
    if (!lens) return T;
    const int64_t l = lens[n];
    return l < 0 ? 0 : (l > T ? T : l);
  }
  // is key j inside query i's context window?
  APS_HD bool visible(int64_t i, int64_t j) const {
    const int64_t cf = i / chunk;
    if (rctx >= 0 && j >= (cf + rctx + 1) * chunk) return false;
    if (lctx >= 0 && j < (cf - lctx) * chunk) return false;
    return true;
  }
  // table row of offset j - i for head h (null outside the table)
  APS_HD const float* table(int64_t h, int64_t i, int64_t j) const {
    if (!rel) return nullptr;
    const int64_t r = j - i + rel_zero;
    return (r >= 0 && r < rel_len) ? rel + h * rel_head_stride + r * dh : nullptr;
  }
  // (q + u) . k_j + (q + v) . E[j - i], scaled; u = v = 0 without the XL biases
  APS_HD float score(int64_t n, int64_t h, int64_t i, int64_t j) const {
    const float* qi = qsrc(n, i, h);
    const float* kj = k(n, j, h);
    const float* uh = rel_u ? rel_u + h * dh : nullptr;
    const float* vh = rel_v ? rel_v + h * dh : nullptr;
    const float* e = table(h, i, j);
    float s = 0.f;
    for (int64_t d = 0; d < dh; ++d) {
      s += (qi[d] + (uh ? uh[d] : 0.f)) * kj[d];
      if (e) s += (qi[d] + (vh ? vh[d] : 0.f)) * e[d];
    }
    return add_mask ? s * scale + add_mask[i * T + j] : s * scale;
  }
};
struct AttentionBackwardRows {
  AttentionGeometry a;
  float* stats;  // [N, H, T, 3]: row max, row sum of exp, D_i
  float* g_qkv;  // [N, T, 3, H, dh]: the q slot is written here (the gradient of the scores' query row)
  float* g_row_k = nullptr;  // XL: [N, T, H, dh] sum_j dS k_j   (its sum over (n, i) is g_u)
  float* g_row_e = nullptr;  // XL: [N, T, H, dh] sum_j dS E_ij  (... g_v)
  APS_HD void operator()(int64_t idx) const {
    const int64_t i = idx % a.T, h = (idx / a.T) % a.H, n = idx / (a.T * a.H);
    const int64_t L = a.keys(n);
    float* gq = g_qkv + ((n * a.T + i) * 3 * a.H + h) * a.dh;
    float* rk = g_row_k ? g_row_k + ((n * a.T + i) * a.H + h) * a.dh : nullptr;
    float* re = g_row_e ? g_row_e + ((n * a.T + i) * a.H + h) * a.dh : nullptr;
    float* st = stats + idx * 3;
    for (int64_t d = 0; d < a.dh; ++d) {
      gq[d] = 0.f;
      if (rk) rk[d] = 0.f;
      if (re) re[d] = 0.f;
    }
    float mx = -INFINITY;
    for (int64_t j = 0; j < L; ++j)
      if (a.visible(i, j)) mx = fmaxf(mx, a.score(n, h, i, j));
    if (!(mx > -INFINITY)) {  // no visible key: the forward's output row is 0 and passes no gradient
      st[0] = 0.f, st[1] = 1.f, st[2] = 0.f;
      return;
    }
    float sum = 0.f;
    for (int64_t j = 0; j < L; ++j)
      if (a.visible(i, j)) sum += expf(a.score(n, h, i, j) - mx);
    const float* gi = a.g(n, i, h);
    float D = 0.f;
    for (int64_t j = 0; j < L; ++j) {
      if (!a.visible(i, j)) continue;
      const float p = expf(a.score(n, h, i, j) - mx) / sum;
      const float* vj = a.v(n, j, h);
      float dp = 0.f;
      for (int64_t d = 0; d < a.dh; ++d) dp += gi[d] * vj[d];
      dp *= a.keep(n, h, i, j);  // ctx = sum_j (P keep) v: the gradient reaches P through the kept weights
      D += p * dp;
    }
    st[0] = mx, st[1] = sum, st[2] = D;
    for (int64_t j = 0; j < L; ++j) {
      if (!a.visible(i, j)) continue;
      const float p = expf(a.score(n, h, i, j) - mx) / sum;
      const float* vj = a.v(n, j, h);
      float dp = 0.f;
      for (int64_t d = 0; d < a.dh; ++d) dp += gi[d] * vj[d];
      dp *= a.keep(n, h, i, j);
      const float ds = p * (dp - D) * a.scale;
      const float* kj = a.k(n, j, h);
      const float* e = a.table(h, i, j);
      for (int64_t d = 0; d < a.dh; ++d) {
        gq[d] += ds * (kj[d] + (e ? e[d] : 0.f));
        if (rk) rk[d] += ds * kj[d];
        if (re && e) re[d] += ds * e[d];
      }
    }
  }
};
struct AttentionBackwardColumns {
  AttentionGeometry a;
  const float* stats;
  float* g_qkv;  // the k and v slots are written here
  APS_HD void operator()(int64_t idx) const {
    const int64_t j = idx % a.T, h = (idx / a.T) % a.H, n = idx / (a.T * a.H);
    const int64_t L = a.keys(n);
    float* gk = g_qkv + ((n * a.T + j) * 3 * a.H + a.H + h) * a.dh;
    float* gv = gk + a.H * a.dh;
    for (int64_t d = 0; d < a.dh; ++d) gk[d] = gv[d] = 0.f;
    if (j >= L) return;  // a masked key receives nothing
    const float* vj = a.v(n, j, h);
    const float* uh = a.rel_u ? a.rel_u + h * a.dh : nullptr;
    for (int64_t i = 0; i < a.T; ++i) {
      if (!a.visible(i, j)) continue;
      const float* st = stats + ((n * a.H + h) * a.T + i) * 3;
      const float p = expf(a.score(n, h, i, j) - st[0]) / st[1];
      const float* gi = a.g(n, i, h);
      const float* qi = a.qsrc(n, i, h);
      const float keep = a.keep(n, h, i, j);
      float dp = 0.f;
      for (int64_t d = 0; d < a.dh; ++d) dp += gi[d] * vj[d];
      const float ds = p * (dp * keep - st[2]) * a.scale;
      for (int64_t d = 0; d < a.dh; ++d) {
        gk[d] += ds * (qi[d] + (uh ? uh[d] : 0.f));
        gv[d] += p * keep * gi[d];
      }
    }
  }
};
struct AttentionBackwardTable {
  AttentionGeometry a;
  const float* stats;
  float* partial;  // [N H, rel_len, dh]
  APS_HD void operator()(int64_t idx) const {
    const int64_t r = idx % a.rel_len, h = (idx / a.rel_len) % a.H, n = idx / (a.rel_len * a.H);
    const int64_t L = a.keys(n);
    float* out = partial + idx * a.dh;
    for (int64_t d = 0; d < a.dh; ++d) out[d] = 0.f;
    const float* vh = a.rel_v ? a.rel_v + h * a.dh : nullptr;
    for (int64_t i = 0; i < a.T; ++i) {
      const int64_t j = i + r - a.rel_zero;
      if (j < 0 || j >= L || !a.visible(i, j)) continue;
      const float* st = stats + ((n * a.H + h) * a.T + i) * 3;
      const float p = expf(a.score(n, h, i, j) - st[0]) / st[1];
      const float* gi = a.g(n, i, h);
      const float* vj = a.v(n, j, h);
      float dp = 0.f;
      for (int64_t d = 0; d < a.dh; ++d) dp += gi[d] * vj[d];
      const float ds = p * (dp * a.keep(n, h, i, j) - st[2]) * a.scale;
      const float* qi = a.qsrc(n, i, h);
      for (int64_t d = 0; d < a.dh; ++d) out[d] += ds * (qi[d] + (vh ? vh[d] : 0.f));
    }
  }
};

// training forward of the general form (windows, XL biases, per-head tables) with dropout on the
// attention weights, one (n, h, i) row per index, scores recomputed (nothing T x T is stored):
// ctx_i = sum_j softmax_j(S)[j] keep(i, j) v_j; a row without a visible key is 0
struct AttentionForwardGeneral {
  AttentionGeometry a;
  float* ctx;  // [N, T, H, dh]
  APS_HD void operator()(int64_t idx) const {
    const int64_t i = idx % a.T, h = (idx / a.T) % a.H, n = idx / (a.T * a.H);
    const int64_t L = a.keys(n);
    float* out = ctx + ((n * a.T + i) * a.H + h) * a.dh;
    for (int64_t d = 0; d < a.dh; ++d) out[d] = 0.f;
    float mx = -INFINITY;
    for (int64_t j = 0; j < L; ++j)
      if (a.visible(i, j)) mx = fmaxf(mx, a.score(n, h, i, j));
    if (!(mx > -INFINITY)) return;
    float sum = 0.f;
    for (int64_t j = 0; j < L; ++j)
      if (a.visible(i, j)) sum += expf(a.score(n, h, i, j) - mx);
    for (int64_t j = 0; j < L; ++j) {
      if (!a.visible(i, j)) continue;
      const float w = expf(a.score(n, h, i, j) - mx) / sum * a.keep(n, h, i, j);
      if (w == 0.f) continue;
      const float* vj = a.v(n, j, h);
      for (int64_t d = 0; d < a.dh; ++d) out[d] += w * vj[d];
    }
  }
};

// ----------------------------------------------------------------------------------------------
// cross attention of the transformer decoder (aps_attention_cross; nn.MultiheadAttention(tgt, memory,
// memory), aps/asr/transformer/decoder.py:78-86): q [N, Tq, H, dh], kv [N, Tk, 2, H, dh] (key | value),
// keys j >= key_lens[n] masked, dropout on the weights (keep factor of (n, h, i, j) from the counter
// hash).  Forward (training: the dropout form) and the two backward passes -- rows (n, h, i): row
// statistics and g_q[i]; columns (n, h, j): g_k[j], g_v[j] -- all recomputing P from q, k.
// ----------------------------------------------------------------------------------------------
struct CrossAttentionGeometry {
  const float* q;
  const float* kv;
  const int64_t* key_lens;  // or null
  const float* g_ctx;       // [N, Tq, H, dh] (backward)
  int64_t Tq, Tk, H, dh;
  float scale;
  float drop_p;
  uint64_t drop_seed;
  const float* add_mask = nullptr;  // additive [Tq, Tk] on the scaled logits (memory_mask, decoder.py:150-186) or null
  APS_HD float keep(int64_t n, int64_t h, int64_t i, int64_t j) const {
    return keep_scale(drop_seed, (uint64_t)(((n * H + h) * Tq + i) * Tk + j), drop_p);
  }
  APS_HD const float* qrow(int64_t n, int64_t i, int64_t h) const { return q + ((n * Tq + i) * H + h) * dh; }
  APS_HD const float* krow(int64_t n, int64_t j, int64_t h) const {
    return kv + ((n * Tk + j) * 2 * H + h) * dh;
  }
  APS_HD const float* vrow(int64_t n, int64_t j, int64_t h) const { return krow(n, j, h) + H * dh; }
  APS_HD const float* g(int64_t n, int64_t i, int64_t h) const { return g_ctx + ((n * Tq + i) * H + h) * dh; }
  APS_HD int64_t keys(int64_t n) const {
    if (!key_lens) return Tk;
    const int64_t l = key_lens[n];
    return l < 0 ? 0 : (l > Tk ? Tk : l);
  }
  APS_HD float score(int64_t n, int64_t h, int64_t i, int64_t j) const {
    const float* qi = qrow(n, i, h);
    const float* kj = krow(n, j, h);
    float s = 0.f;
    for (int64_t d = 0; d < dh; ++d) s += qi[d] * kj[d];
    return add_mask ? s * scale + add_mask[i * Tk + j] : s * scale;
  }
};
// ctx_i = sum_j softmax_j(S)[j] keep(i, j) v_j, one (n, h, i) row per index; no valid key: 0
struct CrossAttentionForward {
  CrossAttentionGeometry a;
  float* ctx;  // [N, Tq, H, dh]
  APS_HD void operator()(int64_t idx) const {
    const int64_t i = idx % a.Tq, h = (idx / a.Tq) % a.H, n = idx / (a.Tq * a.H);
    const int64_t L = a.keys(n);
    float* out = ctx + ((n * a.Tq + i) * a.H + h) * a.dh;
    for (int64_t d = 0; d < a.dh; ++d) out[d] = 0.f;
    if (L == 0) return;
    float mx = -INFINITY;
    for (int64_t j = 0; j < L; ++j) mx = fmaxf(mx, a.score(n, h, i, j));
    if (!(mx > -INFINITY)) return;  // (every key masked by the additive mask: like a row without keys)
    float sum = 0.f;
    for (int64_t j = 0; j < L; ++j) sum += expf(a.score(n, h, i, j) - mx);
    for (int64_t j = 0; j < L; ++j) {
      const float w = expf(a.score(n, h, i, j) - mx) / sum * a.keep(n, h, i, j);
      if (w == 0.f) continue;
      const float* vj = a.vrow(n, j, h);
      for (int64_t d = 0; d < a.dh; ++d) out[d] += w * vj[d];
    }
  }
};
struct CrossAttentionBackwardRows {
  CrossAttentionGeometry a;
  float* stats;  // [N, H, Tq, 3]: row max, row sum of exp, D_i = sum_j P dP
  float* g_q;    // [N, Tq, H, dh]
  APS_HD void operator()(int64_t idx) const {
    const int64_t i = idx % a.Tq, h = (idx / a.Tq) % a.H, n = idx / (a.Tq * a.H);
    const int64_t L = a.keys(n);
    float* gq = g_q + ((n * a.Tq + i) * a.H + h) * a.dh;
    float* st = stats + idx * 3;
    for (int64_t d = 0; d < a.dh; ++d) gq[d] = 0.f;
    if (L == 0) {
      st[0] = 0.f, st[1] = 1.f, st[2] = 0.f;
      return;
    }
    float mx = -INFINITY;
    for (int64_t j = 0; j < L; ++j) mx = fmaxf(mx, a.score(n, h, i, j));
    if (!(mx > -INFINITY)) {  // every key masked by the additive mask
      st[0] = 0.f, st[1] = 1.f, st[2] = 0.f;
      return;
    }
    float sum = 0.f;
    for (int64_t j = 0; j < L; ++j) sum += expf(a.score(n, h, i, j) - mx);
    const float* gi = a.g(n, i, h);
    float D = 0.f;
    for (int64_t j = 0; j < L; ++j) {
      const float p = expf(a.score(n, h, i, j) - mx) / sum;
      const float* vj = a.vrow(n, j, h);
      float dp = 0.f;
      for (int64_t d = 0; d < a.dh; ++d) dp += gi[d] * vj[d];
      D += p * dp * a.keep(n, h, i, j);
    }
    st[0] = mx, st[1] = sum, st[2] = D;
    for (int64_t j = 0; j < L; ++j) {
      const float p = expf(a.score(n, h, i, j) - mx) / sum;
      const float* vj = a.vrow(n, j, h);
      float dp = 0.f;
      for (int64_t d = 0; d < a.dh; ++d) dp += gi[d] * vj[d];
      const float ds = p * (dp * a.keep(n, h, i, j) - D) * a.scale;
      const float* kj = a.krow(n, j, h);
      for (int64_t d = 0; d < a.dh; ++d) gq[d] += ds * kj[d];
    }
  }
};
struct CrossAttentionBackwardColumns {
  CrossAttentionGeometry a;
  const float* stats;
  float* g_kv;  // [N, Tk, 2, H, dh]
  APS_HD void operator()(int64_t idx) const {
    const int64_t j = idx % a.Tk, h = (idx / a.Tk) % a.H, n = idx / (a.Tk * a.H);
    float* gk = g_kv + ((n * a.Tk + j) * 2 * a.H + h) * a.dh;
    float* gv = gk + a.H * a.dh;
    for (int64_t d = 0; d < a.dh; ++d) gk[d] = gv[d] = 0.f;
    if (j >= a.keys(n)) return;  // a masked key receives nothing
    const float* vj = a.vrow(n, j, h);
    for (int64_t i = 0; i < a.Tq; ++i) {
      const float* st = stats + ((n * a.H + h) * a.Tq + i) * 3;
      const float p = expf(a.score(n, h, i, j) - st[0]) / st[1];
      const float keep = a.keep(n, h, i, j);
      const float* gi = a.g(n, i, h);
      const float* qi = a.qrow(n, i, h);
      float dp = 0.f;
      for (int64_t d = 0; d < a.dh; ++d) dp += gi[d] * vj[d];
      const float ds = p * (dp * keep - st[2]) * a.scale;
      for (int64_t d = 0; d < a.dh; ++d) {
        gk[d] += ds * qi[d];
        gv[d] += p * keep * gi[d];
      }
    }
  }
};

// adjoint of an embedding lookup with many rows per table entry (the decoder's token embedding,
// decoder.py:150): the lookups sorted by token id (`order` = the permutation, `sorted_ids` = ids in that
// order); the thread at the start of a run of equal ids sums the run's gradient rows in order.
// g_weight is zero-filled by the caller (tokens that do not occur), index = (position p, column d)
struct EmbeddingBackward {
  const int64_t* sorted_ids;  // [R]
  const int64_t* order;       // [R] row of g for sorted position p
  const float* g;             // [R, D]
  float* g_weight;            // [V, D]
  int64_t R, D, V;
  float scale;
  APS_HD void operator()(int64_t idx) const {
    const int64_t d = idx % D, p = idx / D;
    const int64_t v = sorted_ids[p];
    if (v < 0 || v >= V) return;  // (ids outside the table embed as zeros)
    if (p > 0 && sorted_ids[p - 1] == v) return;
    float acc = 0.f;
    for (int64_t r = p; r < R && sorted_ids[r] == v; ++r) acc += g[order[r] * D + d];
    g_weight[v * D + d] = acc * scale;
  }
};

// The same three passes with the query / gradient rows in registers and the T x T intermediates in
// a scratch (P^T and dS^T, [N, H, T(j), T(i)], i fastest: row threads -- consecutive i -- write them
// coalesced).  ~50x faster than the recomputing form above on the GPU (profiles/r02: 190 ms -> a few
// ms per training step of the benchmark model); DH = head dimension (32 / 64).
// training forward with dropout on the attention weights (one (n, h, i) row per index):
// ctx_i = sum_j softmax_j(S)[j] keep(i, j) v_j
template <int DH>
struct AttentionForwardDropout {
  AttentionGeometry a;
  float* scratch;  // [N, H, T, T] scores (i fastest)
  float* ctx;      // [N, T, H, DH]
  APS_HD void operator()(int64_t idx) const {
    const int64_t T = a.T, H = a.H;
    const int64_t i = idx % T, h = (idx / T) % H, n = idx / (T * H);
    const int64_t L = a.keys(n);
    float* srow = scratch + (n * H + h) * T * T + i;
    float q[DH], acc[DH];
    const float* qi = a.q(n, i, h);
    for (int d = 0; d < DH; ++d) q[d] = qi[d], acc[d] = 0.f;
    float mx = -INFINITY;
    for (int64_t j = 0; j < L; ++j) {
      const float* kj = a.k(n, j, h);
      const int64_t r = j - i + a.rel_zero;
      const float* e = (a.rel && r >= 0 && r < a.rel_len) ? a.rel + r * DH : nullptr;
      float s = 0.f;
      for (int d = 0; d < DH; ++d) s += q[d] * (kj[d] + (e ? e[d] : 0.f));
      s *= a.scale;
      srow[j * T] = s;
      mx = fmaxf(mx, s);
    }
    float sum = 0.f;
    for (int64_t j = 0; j < L; ++j) sum += expf(srow[j * T] - mx);
    for (int64_t j = 0; j < L; ++j) {
      const float w = expf(srow[j * T] - mx) / sum * a.keep(n, h, i, j);
      const float* vj = a.v(n, j, h);
      for (int d = 0; d < DH; ++d) acc[d] += w * vj[d];
    }
    float* out = ctx + ((n * T + i) * H + h) * DH;
    for (int d = 0; d < DH; ++d) out[d] = acc[d];
  }
};

template <int DH>
struct AttentionBackwardRowsFast {
  AttentionGeometry a;
  float* pt;     // [N, H, T, T]: scores, then P^T
  float* dst;    // [N, H, T, T]: dS^T (already times scale)
  float* g_qkv;  // q slot written
  APS_HD void operator()(int64_t idx) const {
    const int64_t T = a.T, H = a.H;
    const int64_t i = idx % T, h = (idx / T) % H, n = idx / (T * H);
    const int64_t L = a.keys(n);
    float* gq = g_qkv + ((n * T + i) * 3 * H + h) * DH;
    float* prow = pt + (n * H + h) * T * T + i;    // element (j, i) at prow[j * T]
    float* drow = dst + (n * H + h) * T * T + i;
    float q[DH], g[DH], acc[DH];
    const float* qi = a.q(n, i, h);
    const float* gi = a.g(n, i, h);
    for (int d = 0; d < DH; ++d) q[d] = qi[d], g[d] = gi[d], acc[d] = 0.f;
    float mx = -INFINITY;
    for (int64_t j = 0; j < L; ++j) {
      const float* kj = a.k(n, j, h);
      const int64_t r = j - i + a.rel_zero;
      const float* e = (a.rel && r >= 0 && r < a.rel_len) ? a.rel + r * DH : nullptr;
      float s = 0.f;
      for (int d = 0; d < DH; ++d) s += q[d] * (kj[d] + (e ? e[d] : 0.f));
      s *= a.scale;
      prow[j * T] = s;
      mx = fmaxf(mx, s);
    }
    float sum = 0.f;
    for (int64_t j = 0; j < L; ++j) sum += expf(prow[j * T] - mx);
    float D = 0.f;
    for (int64_t j = 0; j < L; ++j) {
      const float p = expf(prow[j * T] - mx) / sum;
      const float* vj = a.v(n, j, h);
      float dp = 0.f;
      for (int d = 0; d < DH; ++d) dp += g[d] * vj[d];
      dp *= a.keep(n, h, i, j);  // ctx = sum_j (P keep) v: the gradient reaches P through the kept weights
      prow[j * T] = p;
      drow[j * T] = dp;
      D += p * dp;
    }
    for (int64_t j = 0; j < T; ++j) {
      if (j >= L) {  // masked keys: P = 0, dS = 0
        prow[j * T] = 0.f;
        drow[j * T] = 0.f;
        continue;
      }
      const float ds = prow[j * T] * (drow[j * T] - D) * a.scale;
      drow[j * T] = ds;
      const float* kj = a.k(n, j, h);
      const int64_t r = j - i + a.rel_zero;
      const float* e = (a.rel && r >= 0 && r < a.rel_len) ? a.rel + r * DH : nullptr;
      for (int d = 0; d < DH; ++d) acc[d] += ds * (kj[d] + (e ? e[d] : 0.f));
    }
    for (int d = 0; d < DH; ++d) gq[d] = acc[d];
  }
};
template <int DH>
struct AttentionBackwardColumnsFast {
  AttentionGeometry a;
  const float* pt;
  const float* dst;
  float* g_qkv;  // k and v slots written
  APS_HD void operator()(int64_t idx) const {
    const int64_t T = a.T, H = a.H;
    const int64_t j = idx % T, h = (idx / T) % H, n = idx / (T * H);
    float* gk = g_qkv + ((n * T + j) * 3 * H + H + h) * DH;
    float* gv = gk + H * DH;
    const float* prow = pt + ((n * H + h) * T + j) * T;   // P[i, j] at prow[i]
    const float* drow = dst + ((n * H + h) * T + j) * T;
    float ak[DH], av[DH];
    for (int d = 0; d < DH; ++d) ak[d] = av[d] = 0.f;
    for (int64_t i = 0; i < T; ++i) {
      const float p = prow[i] * a.keep(n, h, i, j), ds = drow[i];
      const float* qi = a.q(n, i, h);
      const float* gi = a.g(n, i, h);
      for (int d = 0; d < DH; ++d) {
        ak[d] += ds * qi[d];
        av[d] += p * gi[d];
      }
    }
    for (int d = 0; d < DH; ++d) gk[d] = ak[d], gv[d] = av[d];
  }
};
template <int DH>
struct AttentionBackwardTableFast {
  AttentionGeometry a;
  const float* dst;
  float* partial;  // [N H, rel_len, DH]
  APS_HD void operator()(int64_t idx) const {
    const int64_t T = a.T, H = a.H;
    const int64_t r = idx % a.rel_len, h = (idx / a.rel_len) % H, n = idx / (a.rel_len * H);
    const float* dmat = dst + (n * H + h) * T * T;
    float acc[DH];
    for (int d = 0; d < DH; ++d) acc[d] = 0.f;
    for (int64_t i = 0; i < T; ++i) {
      const int64_t j = i + r - a.rel_zero;
      if (j < 0 || j >= T) continue;
      const float ds = dmat[j * T + i];
      const float* qi = a.q(n, i, h);
      for (int d = 0; d < DH; ++d) acc[d] += ds * qi[d];
    }
    float* out = partial + idx * DH;
    for (int d = 0; d < DH; ++d) out[d] = acc[d];
  }
};

// ----------------------------------------------------------------------------------------------
// LSTM backward through time (nn.LSTM, gate order i | f | g | o; aps/asr/base/component.py:26-55)
// ----------------------------------------------------------------------------------------------
// out[n, t] = y[n, t - 1], zeros at t = 0  (h_{t-1} of every step as one matrix)
struct TimeShift {
  const float* y;
  float* out;
  int64_t T, H;
  APS_HD void operator()(int64_t i) const {
    const int64_t t = (i / H) % T;
    out[i] = t == 0 ? 0.f : y[i - H];
  }
};
// out[n, t] = x[n, len - 1 - t] for t < len, zeros past the length: the backward direction of a
// bidirectional layer is the forward recurrence on time-reversed utterances (each from its own last
// frame, packed-sequence semantics), and the map is its own inverse
struct ReverseTime {
  const float* x;
  const int64_t* lens;  // or null (= T)
  float* out;
  int64_t T, D;
  APS_HD void operator()(int64_t i) const {
    const int64_t d = i % D, t = (i / D) % T, n = i / (D * T);
    int64_t len = T;
    if (lens) len = lens[n] < 0 ? 0 : (lens[n] > T ? T : lens[n]);
    out[i] = t < len ? x[(n * T + (len - 1 - t)) * D + d] : 0.f;
  }
};
// forward recomputation of the gates and cell states of one layer, one (n, unit) per index:
// gates[n,t,:] <- activated (i, f, g, o), c[n,t,u]; in: pre = x W_ih^T + b_ih, hh = h_{t-1} W_hh^T
struct LstmGateScan {
  const float* pre;   // [N, T, 4H]
  const float* hh;    // [N, T, 4H]
  const float* b_hh;  // [4H] or null
  const int64_t* lens;
  float* gates;  // [N, T, 4H]
  float* c;      // [N, T, H]
  int64_t T, H;
  APS_HD void operator()(int64_t idx) const {
    const int64_t u = idx % H, n = idx / H;
    int64_t len = T;
    if (lens) len = lens[n] < 0 ? 0 : (lens[n] > T ? T : lens[n]);
    float cs = 0.f;
    for (int64_t t = 0; t < T; ++t) {
      const int64_t base = (n * T + t) * 4 * H + u;
      float* go = gates + base;
      if (t >= len) {  // packed sequence: nothing happens past the utterance's length
        go[0] = go[H] = go[2 * H] = go[3 * H] = 0.f;
        c[(n * T + t) * H + u] = 0.f;
        continue;
      }
      float z[4];
      for (int g = 0; g < 4; ++g)
        z[g] = pre[base + g * H] + hh[base + g * H] + (b_hh ? b_hh[g * H + u] : 0.f);
      const float gi = sigmoidf_(z[0]), gf = sigmoidf_(z[1]), gg = tanhf(z[2]), g_o = sigmoidf_(z[3]);
      cs = gf * cs + gi * gg;
      go[0] = gi, go[H] = gf, go[2 * H] = gg, go[3 * H] = g_o;
      c[(n * T + t) * H + u] = cs;
    }
  }
};
// one time step of the reverse sweep, one (n, unit) per index:
//   g_h = g_y[n,t,u] + g_h_rec[n,u]  (g_h_rec = g_pre[n, t+1, :] W_hh, null at the last step)
//   g_c += g_h o (1 - tanh(c)^2);  g_pre = (g_c g i(1-i), g_c c_{t-1} f(1-f), g_c i (1-g^2), g_h tanh(c) o(1-o))
//   g_c <- g_c f
struct LstmBackwardStep {
  const float* gates;
  const float* c;
  const float* g_y;      // [N, T, H]
  const float* g_h_rec;  // [N, H] or null
  const int64_t* lens;
  float* g_c;    // [N, H] carried cell gradient (zeroed by the caller before the sweep)
  float* g_pre;  // [N, T, 4H]
  int64_t T, H, t;
  APS_HD void operator()(int64_t idx) const {
    const int64_t u = idx % H, n = idx / H;
    int64_t len = T;
    if (lens) len = lens[n] < 0 ? 0 : (lens[n] > T ? T : lens[n]);
    const int64_t base = (n * T + t) * 4 * H + u;
    if (t >= len) {
      g_pre[base] = g_pre[base + H] = g_pre[base + 2 * H] = g_pre[base + 3 * H] = 0.f;
      return;
    }
    const float gi = gates[base], gf = gates[base + H], gg = gates[base + 2 * H],
                g_o = gates[base + 3 * H];
    const float ct = c[(n * T + t) * H + u];
    const float cp = t > 0 ? c[(n * T + t - 1) * H + u] : 0.f;
    float gh = g_y[(n * T + t) * H + u];
    if (g_h_rec && t + 1 < len) gh += g_h_rec[n * H + u];
    const float tc = tanhf(ct);
    float gc = (t + 1 < len ? g_c[n * H + u] : 0.f) + gh * g_o * (1.f - tc * tc);
    g_pre[base] = gc * gg * gi * (1.f - gi);
    g_pre[base + H] = gc * cp * gf * (1.f - gf);
    g_pre[base + 2 * H] = gc * gi * (1.f - gg * gg);
    g_pre[base + 3 * H] = gh * tc * g_o * (1.f - g_o);
    g_c[n * H + u] = gc * gf;
  }
};

// ----------------------------------------------------------------------------------------------
// mask-based MVDR (aps/asr/filter/mvdr.py:42-174), adjoints.  The spectrogram X is data (the STFT
// of the input, no trainable parameter upstream): gradients flow to the masks and to the
// ChannelAttention parameters only.
// ----------------------------------------------------------------------------------------------
// v[n, c, f] = |sum_{j != c} Rs[n, f, c, j]| / (C - 1)       (mvdr.py:165-170)
struct OffdiagAbs {
  const float* cov;  // [N, F, C, C, 2]
  float* v;          // [N, C, F]
  int64_t F, C;
  APS_HD void operator()(int64_t idx) const {
    const int64_t f = idx % F, c = (idx / F) % C, n = idx / (F * C);
    const float* row = cov + (((n * F + f) * C + c) * C) * 2;
    float re = 0.f, im = 0.f;
    for (int64_t j = 0; j < C; ++j)
      if (j != c) re += row[2 * j], im += row[2 * j + 1];
    re /= (float)(C - 1), im /= (float)(C - 1);
    v[idx] = sqrtf(re * re + im * im);
  }
};
// one (n, f, c) row of G_Rs per index: G[c, j] += g_v[n,c,f] mean / (|mean| (C - 1)), j != c
struct OffdiagAbsBackward {
  const float* cov;
  const float* g_v;  // [N, C, F]
  float* g_cov;      // [N, F, C, C, 2], ACCUMULATED into
  int64_t F, C;
  APS_HD void operator()(int64_t idx) const {
    const int64_t c = idx % C, f = (idx / C) % F, n = idx / (C * F);
    const float* row = cov + (((n * F + f) * C + c) * C) * 2;
    float re = 0.f, im = 0.f;
    for (int64_t j = 0; j < C; ++j)
      if (j != c) re += row[2 * j], im += row[2 * j + 1];
    re /= (float)(C - 1), im /= (float)(C - 1);
    const float m = sqrtf(re * re + im * im);
    const float s = m > 0.f ? g_v[(n * C + c) * F + f] / (m * (float)(C - 1)) : 0.f;
    float* grow = g_cov + (((n * F + f) * C + c) * C) * 2;
    for (int64_t j = 0; j < C; ++j)
      if (j != c) grow[2 * j] += s * re, grow[2 * j + 1] += s * im;
  }
};

// w = B u / tau,  B = (Rn + eps I)^-1 Rs,  tau = tr(B) + eps        (mvdr.py:75-101), one (n, f)
// per index.  Adjoint: G_v = G_w / conj(tau);  G_tau = -sum_c G_w[c] conj(w[c]) / conj(tau);
// G_B = G_v u^T + G_tau I;  X = A^-H G_B;  G_Rs = X;  G_Rn = -X B^H;  g_u[j] = Re sum_i conj(B[i,j]) G_v[i]
template <int C>
struct MvdrWeightBackward {
  const float* cov_s;  // [N, F, C, C, 2]
  const float* cov_n;
  const float* u;    // [N, C]
  const float* g_w;  // [N, F, C, 2]
  float* g_cov_s;    // [N, F, C, C, 2]
  float* g_cov_n;
  float* g_u_part;   // [N, F, C]  (summed over f by a column reduction)
  int64_t F;
  float eps;
  APS_HD void operator()(int64_t idx) const {
    const int64_t n = idx / F;
    cf A[C][C], Ai[C][C], B[C][C], S[C][C];
    const float* pn = cov_n + idx * C * C * 2;
    const float* ps = cov_s + idx * C * C * 2;
    for (int i = 0; i < C; ++i)
      for (int j = 0; j < C; ++j) {
        A[i][j] = {pn[(i * C + j) * 2] + (i == j ? eps : 0.f), pn[(i * C + j) * 2 + 1]};
        S[i][j] = {ps[(i * C + j) * 2], ps[(i * C + j) * 2 + 1]};
        Ai[i][j] = {i == j ? 1.f : 0.f, 0.f};
      }
    // A^-1 by Gauss-Jordan with partial pivoting
    for (int k = 0; k < C; ++k) {
      int piv = k;
      float best = A[k][k].re * A[k][k].re + A[k][k].im * A[k][k].im;
      for (int r = k + 1; r < C; ++r) {
        const float mag = A[r][k].re * A[r][k].re + A[r][k].im * A[r][k].im;
        if (mag > best) best = mag, piv = r;
      }
      for (int j = 0; j < C; ++j) {
        const cf t0 = A[k][j], t1 = A[piv][j];
        A[k][j] = t1, A[piv][j] = t0;
        const cf s0 = Ai[k][j], s1 = Ai[piv][j];
        Ai[k][j] = s1, Ai[piv][j] = s0;
      }
      const float den = A[k][k].re * A[k][k].re + A[k][k].im * A[k][k].im;
      const cf inv = {A[k][k].re / den, -A[k][k].im / den};
      for (int j = 0; j < C; ++j) A[k][j] = cmul(A[k][j], inv), Ai[k][j] = cmul(Ai[k][j], inv);
      for (int i = 0; i < C; ++i) {
        if (i == k) continue;
        const cf fct = A[i][k];
        for (int j = 0; j < C; ++j) {
          A[i][j] = A[i][j] - cmul(fct, A[k][j]);
          Ai[i][j] = Ai[i][j] - cmul(fct, Ai[k][j]);
        }
      }
    }
    cf tau = {eps, 0.f};
    for (int i = 0; i < C; ++i)
      for (int j = 0; j < C; ++j) {
        cf acc = {0.f, 0.f};
        for (int m = 0; m < C; ++m) acc = acc + cmul(Ai[i][m], S[m][j]);
        B[i][j] = acc;
        if (i == j) tau = tau + acc;
      }
    float uu[C];
    for (int c = 0; c < C; ++c) uu[c] = u[n * C + c];
    const float tden = tau.re * tau.re + tau.im * tau.im;
    const cf itau_c = {tau.re / tden, tau.im / tden};  // 1 / conj(tau)
    cf Gv[C], w[C];
    cf Gtau = {0.f, 0.f};
    for (int i = 0; i < C; ++i) {
      cf v = {0.f, 0.f};
      for (int j = 0; j < C; ++j) v = v + cscale(B[i][j], uu[j]);
      w[i] = cmul(v, cconj(itau_c));  // v / tau
      const cf gw = {g_w[(idx * C + i) * 2], g_w[(idx * C + i) * 2 + 1]};
      Gv[i] = cmul(gw, itau_c);
      Gtau = Gtau - cmul(cmul(gw, cconj(w[i])), itau_c);
    }
    for (int j = 0; j < C; ++j) {
      float g = 0.f;
      for (int i = 0; i < C; ++i) g += B[i][j].re * Gv[i].re + B[i][j].im * Gv[i].im;
      g_u_part[idx * C + j] = g;
    }
    cf GB[C][C], X[C][C];
    for (int i = 0; i < C; ++i)
      for (int j = 0; j < C; ++j) {
        GB[i][j] = cscale(Gv[i], uu[j]);
        if (i == j) GB[i][j] = GB[i][j] + Gtau;
      }
    for (int i = 0; i < C; ++i)
      for (int j = 0; j < C; ++j) {  // X = A^-H G_B
        cf acc = {0.f, 0.f};
        for (int m = 0; m < C; ++m) acc = acc + cmul(cconj(Ai[m][i]), GB[m][j]);
        X[i][j] = acc;
      }
    float* gs = g_cov_s + idx * C * C * 2;
    float* gn = g_cov_n + idx * C * C * 2;
    for (int i = 0; i < C; ++i)
      for (int j = 0; j < C; ++j) {
        gs[(i * C + j) * 2] = X[i][j].re, gs[(i * C + j) * 2 + 1] = X[i][j].im;
        cf acc = {0.f, 0.f};  // -(X B^H)[i][j] = -sum_m X[i][m] conj(B[j][m])
        for (int m = 0; m < C; ++m) acc = acc - cmul(X[i][m], cconj(B[j][m]));
        gn[(i * C + j) * 2] = acc.re, gn[(i * C + j) * 2 + 1] = acc.im;
      }
  }
};

// y[n,t,f] = sum_c conj(w[n,f,c]) x[n,c,t,f]  ->  G_w[c] = sum_t conj(G_y[t]) x[c,t]; one (n, f)
// per index (lanes along f: coalesced rows of the bin-fastest store)
struct BeamformBackwardWeight {
  const float* store;  // [N, C, T, F, 2] with element strides
  const float* g_y;    // [N, T, F, 2]
  float* g_w;          // [N, F, C, 2]
  int64_t C, T, F, stride_n, stride_c, stride_t;
  APS_HD void operator()(int64_t idx) const {
    const int64_t f = idx % F, n = idx / F;
    for (int64_t c = 0; c < C; ++c) {
      float re = 0.f, im = 0.f;
      const float* xs = store + n * stride_n + c * stride_c + 2 * f;
      for (int64_t t = 0; t < T; ++t) {
        const float gr = g_y[((n * T + t) * F + f) * 2], gi = g_y[((n * T + t) * F + f) * 2 + 1];
        const float xr = xs[t * stride_t], xi = xs[t * stride_t + 1];
        re += gr * xr + gi * xi;   // conj(G) x
        im += gr * xi - gi * xr;
      }
      g_w[(idx * C + c) * 2] = re, g_w[(idx * C + c) * 2 + 1] = im;
    }
  }
};

// Covariance R = sum_t m'_t x x^H / max(sum_t m'_t, EPS) with the processed mask m' of
// _process_mask (padded frames zeroed, m / (max_t |m| + EPS)); mvdr.py:42-61, 103-116.  One (n, f)
// per index and mask, two sweeps over the frames:
//   q_t = Re sum_{c,c'} conj(G[c,c']) x_c conj(x_c'),  r = Re <G, R>;   g_m'_t = (q_t - [sum m' > EPS] r) / den
//   m' = m / s, s = max|m| + EPS:  g_m_t = g_m'_t / s - [|m_t| = max] sign(m_t) (sum_t' g_m'_t' m_t') / (s^2 n_max)
template <int C>
struct CovarianceBackward {
  const float* store;
  const float* mask;   // raw [N, T, F]
  const int64_t* lens; // valid frames or null
  const float* cov;    // [N, F, C, C, 2] forward output
  const float* g_cov;  // [N, F, C, C, 2]
  float* g_mask;       // [N, T, F]
  int64_t T, F, stride_n, stride_c, stride_t;
  int mask_norm;
  // [N, T, F] or null: SUBTRACTED from the gradient of the processed mask before _process_mask's own
  // adjoint -- the branch of the implicit noise mask, Rn = estimate_covar(1 - m', X) (mvdr.py:135): its
  // gradient w.r.t. (1 - m') comes from a call of this functor on the complement and enters here
  const float* g_sub;
  APS_HD void operator()(int64_t idx) const {
    const int64_t f = idx % F, n = idx / F;
    int64_t len = T;
    if (lens) len = lens[n] < 0 ? 0 : (lens[n] > T ? T : lens[n]);
    const float* mk = mask + n * T * F + f;
    float* gm = g_mask + n * T * F + f;
    const float* gs = g_sub ? g_sub + n * T * F + f : nullptr;
    float peak = 0.f;
    int nmax = 0;
    if (mask_norm) {
      for (int64_t t = 0; t < len; ++t) peak = fmaxf(peak, fabsf(mk[t * F]));
      for (int64_t t = 0; t < T; ++t) {  // zeroed padded frames take part in the tie count at 0
        const float v = t < len ? fabsf(mk[t * F]) : 0.f;
        if (v == peak) ++nmax;
      }
    }
    const float s = mask_norm ? peak + kEps : 1.f;
    float msum = 0.f;
    for (int64_t t = 0; t < len; ++t) msum += mk[t * F] / s;
    const bool clamped = !(msum > kEps);
    const float den = clamped ? kEps : msum;
    cf G[C][C];
    float r = 0.f;
    const float* pg = g_cov + idx * C * C * 2;
    const float* pr = cov + idx * C * C * 2;
    for (int i = 0; i < C; ++i)
      for (int j = 0; j < C; ++j) {
        G[i][j] = {pg[(i * C + j) * 2], pg[(i * C + j) * 2 + 1]};
        r += G[i][j].re * pr[(i * C + j) * 2] + G[i][j].im * pr[(i * C + j) * 2 + 1];
      }
    if (clamped) r = 0.f;  // clamp(min=EPS) passes no gradient to the denominator
    float dot = 0.f;       // sum_t g_m'_t m_t
    const float* xs = store + n * stride_n + 2 * f;
    for (int64_t t = 0; t < T; ++t) {
      if (t >= len) {
        gm[t * F] = 0.f;
        continue;
      }
      cf x[C];
      for (int c = 0; c < C; ++c) x[c] = {xs[c * stride_c + t * stride_t], xs[c * stride_c + t * stride_t + 1]};
      float q = 0.f;
      for (int i = 0; i < C; ++i)
        for (int j = 0; j < C; ++j) {
          const cf p = cmul(x[i], cconj(x[j]));
          q += G[i][j].re * p.re + G[i][j].im * p.im;
        }
      const float g = (q - r) / den - (gs ? gs[t * F] : 0.f);
      gm[t * F] = g;
      dot += g * mk[t * F];
    }
    if (!mask_norm) return;
    const float back = dot / (s * s * (float)(nmax > 0 ? nmax : 1));
    for (int64_t t = 0; t < len; ++t) {
      const float m = mk[t * F];
      float g = gm[t * F] / s;
      if (fabsf(m) == peak) g -= (m > 0.f ? 1.f : (m < 0.f ? -1.f : 0.f)) * back;
      gm[t * F] = g;
    }
  }
};

// ----------------------------------------------------------------------------------------------
// MlEnhTask (aps/task/ml.py:14-122): log-pdf of the complex angular central Gaussian of the masked
// observations, one (n, f) per index.  With R = sum_t m_t x x^H / max(sum_t m_t, EPS) from
// aps_mvdr_covariance (mask_norm = 0):
//   B = C R + eps I (Hermitian),  D = max(det B, eps),  K_t = max(Re x_t^H B^-1 x_t, eps),
//   log_pdf[n, t, f] = -C log K_t - log D
// (the reference takes det B from the eigenvalues of the real 2C x 2C embedding, ml.py:14-35: the
// same number for a Hermitian matrix).  Adjoint: G_B = C sum_t [K_t > eps] (g_t / K_t) y_t y_t^H
// - [D > eps] (sum_t g_t) B^-H, y_t = B^-1 x_t;  G_R = C G_B.
// ----------------------------------------------------------------------------------------------
template <int C>
struct CacgmmCommon {
  // B^-1 by Gauss-Jordan with partial pivoting; returns det B (complex)
  // singular (optional): set when a pivot is zero or not finite -- where th.inverse raises (aps/cplx.py:268-278)
  APS_HD static cf invert(cf (&A)[C][C], cf (&Ai)[C][C], bool* singular = nullptr) {
    cf det = {1.f, 0.f};
    for (int i = 0; i < C; ++i)
      for (int j = 0; j < C; ++j) Ai[i][j] = {i == j ? 1.f : 0.f, 0.f};
    for (int k = 0; k < C; ++k) {
      int piv = k;
      float best = A[k][k].re * A[k][k].re + A[k][k].im * A[k][k].im;
      for (int r = k + 1; r < C; ++r) {
        const float mag = A[r][k].re * A[r][k].re + A[r][k].im * A[r][k].im;
        if (mag > best) best = mag, piv = r;
      }
      if (piv != k) det = cscale(det, -1.f);
      for (int j = 0; j < C; ++j) {
        const cf t0 = A[k][j], t1 = A[piv][j];
        A[k][j] = t1, A[piv][j] = t0;
        const cf s0 = Ai[k][j], s1 = Ai[piv][j];
        Ai[k][j] = s1, Ai[piv][j] = s0;
      }
      det = cmul(det, A[k][k]);
      // 1 / pivot with the pivot scaled by its larger part first: |p|^2 leaves the fp32 range for |p| > 1.8e19 or
      // < 1e-19, where th.inverse of the real embedding still succeeds; singular = the pivot ITSELF zero or not finite
      const float big = fmaxf(fabsf(A[k][k].re), fabsf(A[k][k].im));
      if (singular && !(big > 0.f && big <= 3.4028234e38f)) *singular = true;
      const float rs = 1.0f / big, xr = A[k][k].re * rs, xi = A[k][k].im * rs, ks = rs / (xr * xr + xi * xi);
      const cf inv = {xr * ks, -xi * ks};
      for (int j = 0; j < C; ++j) A[k][j] = cmul(A[k][j], inv), Ai[k][j] = cmul(Ai[k][j], inv);
      for (int i = 0; i < C; ++i) {
        if (i == k) continue;
        const cf fct = A[i][k];
        for (int j = 0; j < C; ++j) {
          A[i][j] = A[i][j] - cmul(fct, A[k][j]);
          Ai[i][j] = Ai[i][j] - cmul(fct, Ai[k][j]);
        }
      }
    }
    return det;
  }
  APS_HD static void load_b(const float* cov, int64_t idx, float eps, cf (&B)[C][C]) {
    const float* p = cov + idx * C * C * 2;
    for (int i = 0; i < C; ++i)
      for (int j = 0; j < C; ++j) {
        // (B + B^H) / 2 of ml.py:60: the covariance kernel's output is Hermitian already
        const float re = 0.5f * (p[(i * C + j) * 2] + p[(j * C + i) * 2]);
        const float im = 0.5f * (p[(i * C + j) * 2 + 1] - p[(j * C + i) * 2 + 1]);
        B[i][j] = {(float)C * re + (i == j ? eps : 0.f), (float)C * im};
      }
  }
};
template <int C>
struct CacgmmLogPdf {
  const float* store;  // [N, C, T, F, 2] with element strides
  const float* cov;    // [N, F, C, C, 2]
  float* log_pdf;      // [N, T, F]
  int64_t T, F, stride_n, stride_c, stride_t;
  float eps;
  APS_HD void operator()(int64_t idx) const {
    const int64_t f = idx % F, n = idx / F;
    cf B[C][C], Bi[C][C];
    CacgmmCommon<C>::load_b(cov, idx, eps, B);
    const cf det = CacgmmCommon<C>::invert(B, Bi);
    const float D = det.re > eps ? det.re : eps;
    const float logD = logf(D);
    const float* xs = store + n * stride_n + 2 * f;
    for (int64_t t = 0; t < T; ++t) {
      cf x[C];
      for (int c = 0; c < C; ++c)
        x[c] = {xs[c * stride_c + t * stride_t], xs[c * stride_c + t * stride_t + 1]};
      float K = 0.f;
      for (int i = 0; i < C; ++i) {
        cf y = {0.f, 0.f};
        for (int j = 0; j < C; ++j) y = y + cmul(Bi[i][j], x[j]);
        K += x[i].re * y.re + x[i].im * y.im;  // Re conj(x_i) y_i
      }
      K = K > eps ? K : eps;
      log_pdf[(n * T + t) * F + f] = -(float)C * logf(K) - logD;
    }
  }
};
template <int C>
struct CacgmmLogPdfBackward {
  const float* store;
  const float* cov;
  const float* g_log_pdf;  // [N, T, F]
  float* g_cov;            // [N, F, C, C, 2]
  int64_t T, F, stride_n, stride_c, stride_t;
  float eps;
  APS_HD void operator()(int64_t idx) const {
    const int64_t f = idx % F, n = idx / F;
    cf B[C][C], Bi[C][C], G[C][C];
    CacgmmCommon<C>::load_b(cov, idx, eps, B);
    const cf det = CacgmmCommon<C>::invert(B, Bi);
    for (int i = 0; i < C; ++i)
      for (int j = 0; j < C; ++j) G[i][j] = {0.f, 0.f};
    float gsum = 0.f;
    const float* xs = store + n * stride_n + 2 * f;
    for (int64_t t = 0; t < T; ++t) {
      const float g = g_log_pdf[(n * T + t) * F + f];
      gsum += g;
      cf x[C], y[C];
      for (int c = 0; c < C; ++c)
        x[c] = {xs[c * stride_c + t * stride_t], xs[c * stride_c + t * stride_t + 1]};
      float K = 0.f;
      for (int i = 0; i < C; ++i) {
        cf acc = {0.f, 0.f};
        for (int j = 0; j < C; ++j) acc = acc + cmul(Bi[i][j], x[j]);
        y[i] = acc;
        K += x[i].re * acc.re + x[i].im * acc.im;
      }
      if (!(K > eps)) continue;  // clamp(min=eps) passes no gradient
      const float s = (float)C * g / K;
      for (int i = 0; i < C; ++i)
        for (int j = 0; j < C; ++j) G[i][j] = G[i][j] + cscale(cmul(y[i], cconj(y[j])), s);
    }
    if (det.re > eps) {
      for (int i = 0; i < C; ++i)
        for (int j = 0; j < C; ++j) G[i][j] = G[i][j] - cscale(cconj(Bi[j][i]), gsum);  // B^-H
    }
    float* out = g_cov + idx * C * C * 2;
    for (int i = 0; i < C; ++i)
      for (int j = 0; j < C; ++j) {
        // through (B + B^H) / 2 and B = C R + eps I
        const cf h = cscale(G[i][j] + cconj(G[j][i]), 0.5f * (float)C);
        out[(i * C + j) * 2] = h.re, out[(i * C + j) * 2 + 1] = h.im;
      }
  }
};

// ---------------------------------------------------------------------------------------------
// DCCRN masks (aps/sse/bss/dccrn.py:217-242), backward of aps_dccrn_mask.  index = row r (one T-F
// bin); all S speakers of the bin in one thread so that g_store (the sum over speakers) needs no
// atomics.  Complex:  A = sqrt(mr^2 + mi^2 + eps), rho = nl(A) / A, (a, b) = rho (mr, mi);
//   g_mr = ga rho + (ga mr + gb mi) rho' mr / A,   rho' = (nl'(A) A - nl(A)) / A^2   (g_mi alike)
// with (ga, gb) = g_out, or conj(X) g_out when the mask was applied to the spectrogram X.
// ---------------------------------------------------------------------------------------------
APS_HD float mask_nl_value(float v, int nl) {
  if (nl == 1) return v > 0.f ? v : 0.f;
  if (nl == 2) return tanhf(v);
  if (nl == 3) return v > 20.f ? v : log1pf(expf(v));
  if (nl == 4) return 1.0f / (1.0f + expf(-v));
  return v;
}
APS_HD float mask_nl_slope(float v, int nl) {
  if (nl == 1) return v > 0.f ? 1.f : 0.f;
  if (nl == 2) {
    const float t = tanhf(v);
    return 1.0f - t * t;
  }
  if (nl == 3) return v > 20.f ? 1.f : 1.0f / (1.0f + expf(-v));
  if (nl == 4) {
    const float s = 1.0f / (1.0f + expf(-v));
    return s * (1.0f - s);
  }
  return 1.f;
}
struct DccrnMaskBackward {
  const float* dec;    // [rows, 2S] (cplx) or [rows, S]
  const float* store;  // [rows, 2] or null (mode "freq")
  const float* g_out;  // [S, rows, 2]; real masks in mode "freq": [S, rows]
  float* g_dec;        // like dec
  float* g_store;      // [rows, 2] or null
  int64_t rows;
  int S, nl, apply, cplx;
  float eps;
  APS_HD void operator()(int64_t r) const {
    float xr = 0.f, xi = 0.f, gxr = 0.f, gxi = 0.f;
    if (apply) {
      xr = store[r * 2];
      xi = store[r * 2 + 1];
    }
    for (int s = 0; s < S; ++s) {
      const float* go = g_out + ((int64_t)s * rows + r) * (cplx || apply ? 2 : 1);
      if (cplx) {
        const float mr = dec[r * 2 * S + s], mi = dec[r * 2 * S + S + s];
        const float A = sqrtf(mr * mr + mi * mi + eps);
        const float g = mask_nl_value(A, nl);
        const float rho = g / A;
        float ga = go[0], gb = go[1];
        if (apply) {  // o = X (a + ib)
          gxr += go[0] * rho * mr + go[1] * rho * mi;
          gxi += -go[0] * rho * mi + go[1] * rho * mr;
          ga = go[0] * xr + go[1] * xi;
          gb = -go[0] * xi + go[1] * xr;
        }
        const float drho = (mask_nl_slope(A, nl) * A - g) / (A * A);
        const float dot = (ga * mr + gb * mi) * drho / A;
        g_dec[r * 2 * S + s] = ga * rho + dot * mr;
        g_dec[r * 2 * S + S + s] = gb * rho + dot * mi;
      } else {
        const float v = dec[r * S + s];
        float gm = go[0];
        if (apply) {
          const float m = mask_nl_value(v, nl);
          gxr += go[0] * m;
          gxi += go[1] * m;
          gm = go[0] * xr + go[1] * xi;
        }
        g_dec[r * S + s] = gm * mask_nl_slope(v, nl);
      }
    }
    if (g_store) {
      g_store[r * 2] = gxr;
      g_store[r * 2 + 1] = gxi;
    }
  }
};

// ---------------------------------------------------------------------------------------------
// ComplexTensor.__matmul__ / inverse (aps/cplx.py:242-278) on the small matrices the reference uses
// them for ((...) x C x C covariance-sized operands): one output element / one matrix per index.
// ---------------------------------------------------------------------------------------------
struct CplxMatmul {
  const float* a_re;  // [B, M, K]
  const float* a_im;  // [B, M, K] or null (real operand)
  const float* b_re;  // [B or 1, K, N]
  const float* b_im;  // or null
  float* c_re;        // [B, M, N]
  float* c_im;
  int64_t M, K, N, a_batch, b_batch;  // batch strides in elements (0 = broadcast)
  APS_HD void operator()(int64_t idx) const {
    const int64_t n = idx % N, m = (idx / N) % M, b = idx / (N * M);
    const float* ar = a_re + b * a_batch + m * K;
    const float* ai = a_im ? a_im + b * a_batch + m * K : nullptr;
    const float* br = b_re + b * b_batch + n;
    const float* bi = b_im ? b_im + b * b_batch + n : nullptr;
    float re = 0.f, im = 0.f;
    for (int64_t k = 0; k < K; ++k) {
      const float xr = ar[k], xi = ai ? ai[k] : 0.f;
      const float yr = br[k * N], yi = bi ? bi[k * N] : 0.f;
      re += xr * yr - xi * yi;
      im += xi * yr + xr * yi;
    }
    c_re[idx] = re;
    c_im[idx] = im;
  }
};
template <int C>
struct CplxInverse {
  const float* a_re;  // [B, C, C]
  const float* a_im;
  float* o_re;
  float* o_im;
  int32_t* singular;  // sticky count of matrices with a zero / non-finite pivot, or null
  APS_HD void operator()(int64_t idx) const {
    cf A[C][C], Ai[C][C];
    for (int i = 0; i < C; ++i)
      for (int j = 0; j < C; ++j) A[i][j] = {a_re[(idx * C + i) * C + j], a_im[(idx * C + i) * C + j]};
    bool bad = false;
    CacgmmCommon<C>::invert(A, Ai, &bad);  // Gauss-Jordan with partial pivoting, in registers
    if (bad && singular) {
#if defined(__HIP_DEVICE_COMPILE__)
      atomicAdd(singular, 1);
#else
      *singular += 1;
#endif
    }
    for (int i = 0; i < C; ++i)
      for (int j = 0; j < C; ++j) {
        o_re[(idx * C + i) * C + j] = Ai[i][j].re;
        o_im[(idx * C + i) * C + j] = Ai[i][j].im;
      }
  }
};

}  // namespace grad
}  // namespace aps
#endif  // APS_AMD_GRAD_CORE_H_
