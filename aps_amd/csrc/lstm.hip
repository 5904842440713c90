// Persistent LSTM layer for gfx950: the recurrence of the RNN mask estimator
// (PyTorchRNNEncoder -> nn.LSTM, aps/asr/base/encoder.py:87-184, aps/asr/base/component.py:26-55,
// 145-190) as ONE launch per layer and direction instead of a GEMM + pointwise launch per step.
//
//   gates_t = pre_t + h_{t-1} W_hh^T + b_hh,   pre = x W_ih^T + b_ih  (one batched GEMM, aps_linear)
//   i, f, g, o = sigmoid, sigmoid, tanh, sigmoid  (torch gate order)
//   c_t = f c_{t-1} + i g,  h_t = o tanh(c_t)
//
// Decomposition: workgroup b owns the 4 hidden units 4b .. 4b+3, i.e. 16 rows of W_hh (4 gates x 4
// units), for the whole sequence.  Its slice of W_hh lives in VGPRs (4 waves split K = H, each lane
// H/16 values: the B operand of v_mfma_f32_16x16x4_f32, exact fp32), the cell state c in the
// registers of the 4 N gate threads.  Per step a workgroup needs all of h_{t-1} ([N, H], gathered
// from the layer output y itself) and produces 4 columns of h_t.
//
// Inter-workgroup hand-off per step (G = H / 4 workgroups, all resident: G <= 256 CUs):
//   producer: h_t chunk -> y with 16-byte write-through (sc1) stores, every storing wave drains
//             (s_waitcnt vmcnt(0)), __syncthreads(), lane 0 stores flag[b] = t + 1 (relaxed, agent)
//   consumer: one wave polls the G flags (relaxed agent loads, s_sleep) until all >= t, then every
//             wave gathers h_{t-1} with 16-byte sc1 loads (L1-bypassing, so no acquire fence)
// Flags are zeroed by a memset node ahead of the launch; spins are bounded (a timeout word in the
// workspace turns a lost workgroup into a reported error, not a hang).
//
// Sequence lengths follow the packed-sequence semantics of the reference: outputs at t >= len are
// zero; the reverse direction of a bidirectional layer starts at each utterance's own last frame.
#include "common.h"

namespace aps {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kLstmUnits = 4;            // hidden units per workgroup
constexpr int kLstmRows = 4 * kLstmUnits;  // W_hh rows per workgroup (the MFMA N dimension)
constexpr unsigned kSpinLimit = 1u << 22;

struct LstmArgs {
  const float* pre;     // [N, T, 4H]
  const float* w_hh;    // [4H, H]
  const float* b_hh;    // [4H] or null
  const int64_t* lens;  // [N] or null
  float* y;             // [N, T, ldy], this direction's columns start at y
  unsigned* flags;      // [G] step flags, [G] = timeout word
  int32_t N, T, H, ldy, reverse;
};

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_f(float x) {
  // 1 - 2 / (e^{2x} + 1): saturates cleanly (e^{2x} = inf -> 1, 0 -> -1), abs error ~1e-7
  return 1.0f - 2.0f / (__expf(2.0f * x) + 1.0f);
}

template <int KREGS, int MT>
__global__ __launch_bounds__(256) void lstm_layer_kernel(LstmArgs a) {
  constexpr int H = 16 * KREGS;
  constexpr int PITCH = H + 2;  // == 2 mod 32: the MFMA A fetch (row l & 15, k l >> 4) is conflict free
  constexpr int ROWS = 16 * MT;
  extern __shared__ __attribute__((aligned(16))) float s_dyn[];
  float* s_h = s_dyn;                  // [ROWS][PITCH]
  float* s_red = s_dyn + ROWS * PITCH;  // [4][ROWS][kLstmRows + 1]
  const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
  const int b = blockIdx.x, G = gridDim.x;
  const int u0 = b * kLstmUnits;
  const int N = a.N, T = a.T;

  // ---- resident W_hh slice: lane (j = ln & 15, kk = ln >> 4) holds W[row(j)][wv H/4 + 4 s + kk]
  float wreg[KREGS];
  {
    const int j = ln & 15;
    const int row = (j >> 2) * H + u0 + (j & 3);
    const float* wp = a.w_hh + (int64_t)row * H + wv * (H / 4) + (ln >> 4);
#pragma unroll
    for (int s = 0; s < KREGS; ++s) wreg[s] = wp[4 * s];
  }
  // ---- gate role: thread (n, u)
  const int gn = tid >> 2, gu = tid & 3;
  const bool gate_thread = gn < N;
  const int len = gate_thread ? (a.lens ? (int)min((int64_t)T, max((int64_t)0, a.lens[gn])) : T) : 0;
  float bias[4] = {0.f, 0.f, 0.f, 0.f};
  if (gate_thread && a.b_hh) {
#pragma unroll
    for (int g = 0; g < 4; ++g) bias[g] = a.b_hh[g * H + u0 + gu];
  }
  float c = 0.f;

  // buffer descriptor over y for the sc1 (write-through / L1-bypassing) 16-byte accesses
  const uint32_t y_bytes = (uint32_t)((int64_t)N * T * a.ldy * 4);
  auto rsrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, y_bytes, 0x00020000);
  unsigned* tmo = a.flags + G;

  // gather roles: float4 chunk q of row r, 16 per thread at ROWS = 32, H = 512
  constexpr int CH = H / 4;                 // float4 chunks per row
  constexpr int NLOAD = ROWS * CH / 256;    // H % 16 == 0 and ROWS % 16 == 0 -> exact
  // per-row previous-step time index, recomputed each step from lens (rows = utterances)

  for (int s = 0; s < T; ++s) {
    // ---- prefetch this step's input pre-activations (independent of the hand-off)
    const bool valid = gate_thread && s < len;
    const int t_cur = a.reverse ? len - 1 - s : s;
    float p[4] = {0.f, 0.f, 0.f, 0.f};
    if (valid) {
      const float* pp = a.pre + ((int64_t)gn * T + t_cur) * 4 * H + u0 + gu;
#pragma unroll
      for (int g = 0; g < 4; ++g) p[g] = pp[g * H];
    }
    float part[4] = {0.f, 0.f, 0.f, 0.f};
    if (s > 0) {
      // ---- wait until every workgroup has published step s - 1
      if (wv == 0) {
        unsigned spins = 0;
        bool failed = __hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        while (!failed) {
          bool ok = true;
          for (int i = ln; i < G; i += 64)
            ok &= __hip_atomic_load(a.flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >=
                  (unsigned)s;
          if (__all(ok)) break;
          __builtin_amdgcn_s_sleep(1);
          if (++spins > kSpinLimit) {
            failed = true;
            if (ln == 0) __hip_atomic_store(tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
      __syncthreads();
      // ---- gather h_{s-1}: row r of utterance r from y[r, t_prev(r), :]
      u32x4 v[NLOAD];
#pragma unroll
      for (int i = 0; i < NLOAD; ++i) {
        const int idx = tid + 256 * i;
        const int r = idx / CH, q = idx % CH;
        int rl = 0;
        if (r < N) rl = a.lens ? (int)min((int64_t)T, max((int64_t)0, a.lens[r])) : T;
        v[i] = u32x4{0u, 0u, 0u, 0u};
        if (s < rl) {
          const int tp = a.reverse ? rl - s : s - 1;
          const uint32_t off = (uint32_t)((((int64_t)r * T + tp) * a.ldy + 4 * q) * 4);
          v[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 16);
        }
      }
#pragma unroll
      for (int i = 0; i < NLOAD; ++i) {
        const int idx = tid + 256 * i;
        const int r = idx / CH, q = idx % CH;
        float2* dst = reinterpret_cast<float2*>(s_h + r * PITCH + 4 * q);
        dst[0] = make_float2(__uint_as_float(v[i].x), __uint_as_float(v[i].y));
        dst[1] = make_float2(__uint_as_float(v[i].z), __uint_as_float(v[i].w));
      }
      __syncthreads();
      // ---- partial products over this wave's K quarter
      f32x4 acc[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      const float* hp = s_h + (ln & 15) * PITCH + wv * (H / 4) + (ln >> 4);
#pragma unroll
      for (int k = 0; k < KREGS; ++k) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(hp[m * 16 * PITCH + 4 * k], wreg[k], acc[m],
                                                         0, 0, 0);
      }
      // D layout: lane l, register r -> (batch row 4 (l >> 4) + r, gate row l & 15)
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          s_red[(wv * ROWS + m * 16 + 4 * (ln >> 4) + r) * (kLstmRows + 1) + (ln & 15)] = acc[m][r];
      __syncthreads();
      if (gate_thread) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float t = 0.f;
#pragma unroll
          for (int w = 0; w < 4; ++w) t += s_red[(w * ROWS + gn) * (kLstmRows + 1) + g * 4 + gu];
          part[g] = t;
        }
      }
    }
    // ---- gates, cell update, publish
    float h = 0.f;
    if (valid) {
      const float gi = sigmoid_f(p[0] + part[0] + bias[0]);
      const float gf = sigmoid_f(p[1] + part[1] + bias[1]);
      const float gg = tanh_f(p[2] + part[2] + bias[2]);
      const float go = sigmoid_f(p[3] + part[3] + bias[3]);
      c = gf * c + gi * gg;
      h = go * tanh_f(c);
    }
    // the 4 units of an utterance sit in 4 adjacent lanes: lane gu == 0 stores all 16 bytes
    const float h1 = __shfl_down(h, 1, 64), h2 = __shfl_down(h, 2, 64), h3 = __shfl_down(h, 3, 64);
    if (gate_thread && gu == 0) {
      const int t_out = (s < len) ? t_cur : s;  // padded frames: zeros at their own index
      const uint32_t off = (uint32_t)((((int64_t)gn * T + t_out) * a.ldy + u0) * 4);
      u32x4 o = {__float_as_uint(h), __float_as_uint(h1), __float_as_uint(h2), __float_as_uint(h3)};
      __builtin_amdgcn_raw_buffer_store_b128(o, rsrc, off, 0, 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its sc1 stores
    __syncthreads();
    if (tid == 0)
      __hip_atomic_store(a.flags + b, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <int KREGS>
static int launch_lstm(const LstmArgs& a, hipStream_t st) {
  constexpr int H = 16 * KREGS;
  const int G = H / kLstmUnits;
  const int MT = (a.N + 15) / 16;
  const size_t lds = (size_t)(16 * MT) * (H + 2 + 4 * (kLstmRows + 1)) * sizeof(float);
  if (lds > 160 * 1024) return APS_ERR_UNSUPPORTED;
  if (hipMemsetAsync(a.flags, 0, (size_t)(G + 1) * sizeof(unsigned), st) != hipSuccess)
    return APS_ERR_LAUNCH;
  switch (MT) {
#define APS_LSTM_CASE(M)                                                                      \
  case M:                                                                                     \
    if (lds > 64 * 1024 &&                                                                    \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_layer_kernel<KREGS, M>),      \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
      return APS_ERR_LAUNCH;                                                                  \
    hipLaunchKernelGGL((lstm_layer_kernel<KREGS, M>), dim3(G), dim3(256), lds, st, a);        \
    break;
    APS_LSTM_CASE(1)
    APS_LSTM_CASE(2)
    APS_LSTM_CASE(3)
    APS_LSTM_CASE(4)
#undef APS_LSTM_CASE
    default:
      return APS_ERR_UNSUPPORTED;
  }
  return aps_launch_status();
}

}  // namespace aps

using namespace aps;

extern "C" int64_t aps_lstm_workspace(int64_t H) {
  if (H <= 0 || H % 16) return -1;
  return (H / kLstmUnits + 1) * (int64_t)sizeof(unsigned);
}

extern "C" int aps_lstm_layer(const float* pre, const float* w_hh, const float* b_hh,
                              const int64_t* lens, float* y, int64_t N, int64_t T, int64_t H,
                              int64_t ldy, int32_t reverse, void* workspace, void* stream) {
  APS_CHECK_ARG(pre && w_hh && y && workspace && N > 0 && T > 0 && H > 0 && ldy >= H);
  APS_CHECK_ARG(ldy % 4 == 0 && ((uintptr_t)y & 15) == 0);
  if (N > 64 || N * T * ldy * 4 >= ((int64_t)1 << 31)) return APS_ERR_UNSUPPORTED;
  LstmArgs a{pre, w_hh, b_hh, lens, y, static_cast<unsigned*>(workspace),
             (int32_t)N, (int32_t)T, (int32_t)H, (int32_t)ldy, reverse};
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (H) {
    case 128: return launch_lstm<8>(a, st);
    case 256: return launch_lstm<16>(a, st);
    case 320: return launch_lstm<20>(a, st);
    case 384: return launch_lstm<24>(a, st);
    case 512: return launch_lstm<32>(a, st);
    case 640: return launch_lstm<40>(a, st);
    case 768: return launch_lstm<48>(a, st);
    case 1024: return launch_lstm<64>(a, st);
    default: return APS_ERR_UNSUPPORTED;
  }
}

// 1 when a bounded spin of the last aps_lstm_layer call on this workspace expired (a workgroup
// was not resident); the layer output is then invalid.  Reads the word with a blocking copy.
extern "C" int aps_lstm_timed_out(const void* workspace, int64_t H, void* stream) {
  if (!workspace || H <= 0 || H % 16) return APS_ERR_INVALID;
  unsigned v = 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (hipMemcpyAsync(&v, static_cast<const unsigned*>(workspace) + H / kLstmUnits, sizeof(v),
                     hipMemcpyDeviceToHost, st) != hipSuccess)
    return APS_ERR_LAUNCH;
  if (hipStreamSynchronize(st) != hipSuccess) return APS_ERR_LAUNCH;
  return v != 0;
}
