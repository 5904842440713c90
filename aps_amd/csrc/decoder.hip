// Step kernels of the RNN attention decoder (aps/asr/base/decoder.py:69-218,
// aps/asr/base/attention.py:76-259).  The decoder is a sequential loop over target positions; per
// step the projections run on the GEMM (nn.hip) and these two kernels do the rest:
//   * lstm_cell_kernel: gates -> (c, h) for one time step with carried state (nn.LSTM called with
//     hx on a length-1 sequence, decoder.py:128-135),
//   * att_step_kernel: one workgroup per utterance evaluates the scores of all encoder frames
//     (context / dot / location-aware forms), the masked softmax and the context vector.
#include "common.h"

namespace aps {

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

// pre [N, 4H] = x W_ih^T + b_ih + h W_hh^T + b_hh (torch gate order i | f | g | o)
__global__ __launch_bounds__(256) void lstm_cell_kernel(const float* __restrict__ pre,
                                                        const float* __restrict__ c_prev,
                                                        float* __restrict__ h_out,
                                                        float* __restrict__ c_out, int64_t total,
                                                        int H) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * 256) {
    const int64_t n = i / H;
    const int u = (int)(i % H);
    const float* p = pre + n * 4 * H + u;
    const float gi = sigm(p[0]), gf = sigm(p[H]), gg = tanhf(p[2 * H]), go = sigm(p[3 * H]);
    const float c = gf * (c_prev ? c_prev[i] : 0.f) + gi * gg;
    c_out[i] = c;
    h_out[i] = go * tanhf(c);
  }
}

// One time step of any nn.RNNBase cell on the gate pre-activations (the step-by-step form of the
// recurrences that have no persistent kernel: GRU, tanh / relu RNN, LSTMs of other sizes or with
// a projection).  gx = x_t W_ih^T + b_ih (row pitch ldx, taken from the whole-sequence GEMM),
// gh = h_{t-1} W_hh^T + b_hh; torch gate orders: GRU r | z | n, LSTM i | f | g | o.
// Packed-sequence semantics: a row with t >= len keeps its state and emits zeros.
//   mode 0 GRU: r = s(gx_r + gh_r), z = s(gx_z + gh_z), n = tanh(gx_n + r gh_n), h = (1 - z) n + z h'
//   mode 1 / 2: h = tanh / relu(gx + gh)          mode 3 LSTM: c = f c' + i g, h = o tanh(c)
__global__ __launch_bounds__(256) void rnn_step_kernel(const float* __restrict__ gx, int64_t ldx,
                                                       const float* __restrict__ gh,
                                                       const float* __restrict__ h_prev,
                                                       const float* __restrict__ c_prev,
                                                       const int64_t* __restrict__ lens, int64_t t,
                                                       float* __restrict__ h_out,
                                                       float* __restrict__ c_out,
                                                       float* __restrict__ y, int64_t ldy,
                                                       int64_t total, int H, int mode) {
  const int G = mode == 0 ? 3 : (mode == 3 ? 4 : 1);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * 256) {
    const int64_t n = i / H;
    const int u = (int)(i % H);
    const float hp = h_prev ? h_prev[i] : 0.f;
    const float cp = (mode == 3 && c_prev) ? c_prev[i] : 0.f;
    const bool live = !lens || t < lens[n];
    const float* px = gx + n * ldx + u;
    const float* ph = gh + n * (int64_t)G * H + u;
    float h = hp, c = cp;
    if (live) {
      if (mode == 0) {
        const float r = sigm(px[0] + ph[0]), z = sigm(px[H] + ph[H]);
        const float nn_ = tanhf(px[2 * H] + r * ph[2 * H]);
        h = (1.0f - z) * nn_ + z * hp;
      } else if (mode == 3) {
        const float gi = sigm(px[0] + ph[0]), gf = sigm(px[H] + ph[H]);
        const float gg = tanhf(px[2 * H] + ph[2 * H]), go = sigm(px[3 * H] + ph[3 * H]);
        c = gf * cp + gi * gg;
        h = go * tanhf(c);
      } else {
        const float v = px[0] + ph[0];
        h = mode == 1 ? tanhf(v) : fmaxf(v, 0.f);
      }
    }
    h_out[i] = h;
    if (c_out) c_out[i] = c;
    if (y) y[n * ldy + u] = live ? h : 0.f;
  }
}

struct AttStepArgs {
  const float* enc_part;   // [N, T, A]  enc_proj(enc_pad)
  const float* enc_pad;    // [N, T, D]
  const float* dec_part;   // [N, A]     dec_proj(dec_prev)
  const float* w;          // [A]        score vector (ctx / loc) or null (dot)
  const int64_t* enc_len;  // [N] or null
  const float* ali_prev;   // [N, T] or null (loc: null = the uniform initial alignment)
  const float* loc_f;      // [C, 2L+1]  location filter F.weight (loc) or null
  const float* loc_fb;     // [C]        F.bias
  const float* loc_att;    // [A, C]     att.weight (1 x 1 conv)
  float* ali;              // [N, T]
  float* ctx;              // [N, D]
  int32_t T, A, D, C, L;
  int32_t mode;            // 0 ctx, 1 dot, 2 loc
  float scale;             // dot: 1 / sqrt(A) or 1
  // multi-head forms (attention.py:266-531): blockIdx.y = head; head h owns columns h A .. of the
  // keys (enc_part) / queries / w / att rows, columns h D .. of the values (enc_pad), its own C
  // location filters, row (n H + h) of the alignments.  H = 1: the single-head layouts above.
  int32_t H;
};

__global__ __launch_bounds__(256) void att_step_kernel(AttStepArgs a) {
  extern __shared__ float s_att[];  // score [T] | previous alignment [T] | location features [T][C]
  const int n = blockIdx.x, h = blockIdx.y, tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
  const int T = a.T, A = a.A, D = a.D, H = a.H;
  const int64_t nh = (int64_t)n * H + h;
  a.enc_part += (int64_t)h * A, a.enc_pad += (int64_t)h * D, a.dec_part += (int64_t)h * A;
  if (a.w) a.w += (int64_t)h * A;
  if (a.loc_f) a.loc_f += (int64_t)h * a.C * (2 * a.L + 1);
  if (a.loc_fb) a.loc_fb += (int64_t)h * a.C;
  if (a.loc_att) a.loc_att += (int64_t)h * A * a.C;
  const int64_t ldk = (int64_t)H * A, ldv = (int64_t)H * D;
  float* s_score = s_att;
  float* s_prev = s_att + T;
  float* s_loc = s_att + 2 * T;
  const int len = a.enc_len ? (int)min((int64_t)T, max((int64_t)0, a.enc_len[n])) : T;
  if (a.mode == 2) {
    // initial alignment: uniform over the valid frames (attention.py:121-128)
    for (int t = tid; t < T; t += 256)
      s_prev[t] = a.ali_prev ? a.ali_prev[nh * T + t] : (t < len ? 1.0f / (float)len : 0.f);
    __syncthreads();
    const int K = 2 * a.L + 1;
    for (int i = tid; i < T * a.C; i += 256) {  // F: Conv1d(1, C, 2L + 1, padding L)
      const int t = i / a.C, c = i % a.C;
      float acc = a.loc_fb ? a.loc_fb[c] : 0.f;
      const int k0 = max(0, a.L - t), k1 = min(K, T + a.L - t);
      for (int k = k0; k < k1; ++k) acc += a.loc_f[c * K + k] * s_prev[t + k - a.L];
      s_loc[i] = acc;
    }
    __syncthreads();
  }
  const float* dp = a.dec_part + (int64_t)n * ldk;
  for (int t = wv; t < T; t += 4) {  // a wavefront per frame, lanes along the attention dimension
    const float* ep = a.enc_part + ((int64_t)n * T + t) * ldk;
    float acc = 0.f;
    for (int j = ln; j < A; j += 64) {
      if (a.mode == 1) {
        acc += ep[j] * dp[j];
      } else {
        float v = ep[j] + dp[j];
        if (a.mode == 2) {
          float lp = 0.f;
          for (int c = 0; c < a.C; ++c) lp += a.loc_att[j * a.C + c] * s_loc[t * a.C + c];
          v += lp;
        }
        acc += a.w[j] * tanhf(v);
      }
    }
    acc = wave_sum(acc);
    if (ln == 0) s_score[t] = (t < len) ? acc * a.scale : -INFINITY;
  }
  __syncthreads();
  // masked softmax over the frames (attention.py:56-70)
  float m = -INFINITY;
  for (int t = tid; t < T; t += 256) m = fmaxf(m, s_score[t]);
  __shared__ float s_red[8];
  m = wave_max(m);
  if (ln == 0) s_red[wv] = m;
  __syncthreads();
  m = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
  float sum = 0.f;
  for (int t = tid; t < T; t += 256) {
    const float e = (s_score[t] > -INFINITY) ? expf(s_score[t] - m) : 0.f;
    s_score[t] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  if (ln == 0) s_red[4 + wv] = sum;
  __syncthreads();
  sum = s_red[4] + s_red[5] + s_red[6] + s_red[7];
  const float inv = sum > 0.f ? 1.0f / sum : 0.f;
  for (int t = tid; t < T; t += 256) {
    const float p = s_score[t] * inv;
    s_score[t] = p;
    a.ali[nh * T + t] = p;
  }
  __syncthreads();
  // context: sum_t ali[t] enc_pad[n, t, :]
  for (int d = tid; d < D; d += 256) {
    const float* xp = a.enc_pad + (int64_t)n * T * ldv + d;
    float acc = 0.f;
    for (int t = 0; t < len; ++t) acc += s_score[t] * xp[(int64_t)t * ldv];
    a.ctx[(int64_t)n * ldv + (int64_t)h * D + d] = acc;
  }
}

static unsigned grid_for(int64_t total) {
  int64_t blocks = (total + 255) / 256;
  return (unsigned)(blocks > 8192 ? 8192 : blocks);
}

}  // namespace aps

using namespace aps;

extern "C" int aps_lstm_cell(const float* pre, const float* c_prev, float* h_out, float* c_out,
                             int64_t N, int64_t H, void* stream) {
  APS_CHECK_ARG(pre && h_out && c_out && N > 0 && H > 0 && H < (1 << 30));
  hipLaunchKernelGGL(lstm_cell_kernel, dim3(grid_for(N * H)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), pre, c_prev, h_out, c_out, N * H, (int)H);
  return aps_launch_status();
}

extern "C" int aps_rnn_step(const float* gx, int64_t ldx, const float* gh, const float* h_prev,
                            const float* c_prev, const int64_t* lens, int64_t t, float* h_out,
                            float* c_out, float* y, int64_t ldy, int64_t N, int64_t H, int32_t mode,
                            void* stream) {
  APS_CHECK_ARG(gx && gh && h_out && N > 0 && H > 0 && H < (1 << 28) && mode >= 0 && mode <= 3);
  APS_CHECK_ARG(ldx >= (mode == 0 ? 3 : (mode == 3 ? 4 : 1)) * H && (!y || ldy >= H) && t >= 0);
  APS_CHECK_ARG(mode != 3 || c_out);
  hipLaunchKernelGGL(rnn_step_kernel, dim3(grid_for(N * H)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), gx, ldx, gh, h_prev, c_prev, lens, t, h_out,
                     c_out, y, ldy, N * H, (int)H, (int)mode);
  return aps_launch_status();
}

extern "C" int aps_att_step(const float* enc_part, const float* enc_pad, const float* dec_part,
                            const float* w, const int64_t* enc_len, const float* ali_prev,
                            const float* loc_filter, const float* loc_filter_bias,
                            const float* loc_att, float* ali, float* ctx, int64_t N, int64_t T,
                            int64_t A, int64_t D, int64_t C, int64_t L, int32_t mode, float scale,
                            void* stream) {
  APS_CHECK_ARG(enc_part && enc_pad && dec_part && ali && ctx && N > 0 && N <= 0x7fffffff);
  APS_CHECK_ARG(T > 0 && A > 0 && D > 0 && mode >= 0 && mode <= 2);
  APS_CHECK_ARG(mode == 1 || w);
  APS_CHECK_ARG(mode != 2 || (loc_filter && loc_att && C > 0 && L >= 0));
  const size_t lds = (size_t)(2 * T + (mode == 2 ? T * C : 0)) * sizeof(float);
  if (lds > 64 * 1024) return APS_ERR_UNSUPPORTED;
  AttStepArgs a{enc_part, enc_pad, dec_part, w, enc_len, ali_prev, loc_filter, loc_filter_bias,
                loc_att, ali, ctx, (int32_t)T, (int32_t)A, (int32_t)D, (int32_t)C, (int32_t)L, mode,
                scale, 1};
  hipLaunchKernelGGL(att_step_kernel, dim3((unsigned)N), dim3(256), lds,
                     static_cast<hipStream_t>(stream), a);
  return aps_launch_status();
}

extern "C" int aps_att_step_heads(const float* key, const float* value, const float* dec_part,
                                  const float* w, const int64_t* enc_len, const float* ali_prev,
                                  const float* loc_filter, const float* loc_filter_bias,
                                  const float* loc_att, float* ali, float* ctx, int64_t N,
                                  int64_t T, int64_t H, int64_t A, int64_t Dv, int64_t C, int64_t L,
                                  int32_t mode, float scale, void* stream) {
  APS_CHECK_ARG(key && value && dec_part && ali && ctx && N > 0 && N <= 0x7fffffff);
  APS_CHECK_ARG(T > 0 && A > 0 && Dv > 0 && H > 0 && H <= 65535 && mode >= 0 && mode <= 2);
  APS_CHECK_ARG(mode == 1 || w);
  APS_CHECK_ARG(mode != 2 || (loc_filter && loc_att && C > 0 && L >= 0));
  const size_t lds = (size_t)(2 * T + (mode == 2 ? T * C : 0)) * sizeof(float);
  if (lds > 64 * 1024) return APS_ERR_UNSUPPORTED;
  AttStepArgs a{key, value, dec_part, w, enc_len, ali_prev, loc_filter, loc_filter_bias,
                loc_att, ali, ctx, (int32_t)T, (int32_t)A, (int32_t)Dv, (int32_t)C, (int32_t)L,
                mode, scale, (int32_t)H};
  hipLaunchKernelGGL(att_step_kernel, dim3((unsigned)N, (unsigned)H), dim3(256), lds,
                     static_cast<hipStream_t>(stream), a);
  return aps_launch_status();
}
