// Spatial feature layers of the enhancement front end (aps/transform/enh.py:146-384):
//   FixedBeamformer  y[n, b, f, t] = sum_c conj(w[b, c, f]) x[n, c, f, t]
//   DfTransform      af[n, d, t, f] = mean_p cos(ipd_p[n, t, f] - (phi[n, d, l_p, f] - phi[n, d, r_p, f]))
// Both are streaming kernels: every input value is read from HBM once, every output written once.
#include "common.h"

namespace aps {

// ------------------------------------------------------------------------------------------
// Fixed beamformer.  x as two real tensors [N, C, F, T] (the reference keeps real / imag apart),
// w [B, C, F] real / imag, out [N, B, F, T] (all beams) or [N, F, T] (beam[n] given).
// A workgroup owns (n, f) and 256 frames per pass: the C channel values of a frame stay in
// registers across the beams (kBeamC channels per register block), the weights of (b, :, f) are
// wave-uniform scalar loads.  HBM: reads 2 C T floats per (n, f), writes 2 B T.
// ------------------------------------------------------------------------------------------
constexpr int kBeamC = 8;

struct BeamArgs {
  const float* xr;
  const float* xi;
  const float* wr;
  const float* wi;
  const int64_t* beam;  // nullptr: all B beams
  float* yr;
  float* yi;
  int64_t N, T;
  int C, F, B;
};

__global__ __launch_bounds__(256) void fixed_beam_kernel(BeamArgs a) {
  const int f = blockIdx.x;
  const int64_t n = blockIdx.y;
  const int64_t t = (int64_t)blockIdx.z * 256 + threadIdx.x;
  const bool live = t < a.T;
  const int64_t tc = live ? t : a.T - 1;
  const int64_t chan = (int64_t)a.F * a.T;
  const float* xr = a.xr + (n * a.C * a.F + f) * a.T + tc;
  const float* xi = a.xi + (n * a.C * a.F + f) * a.T + tc;
  int b0 = 0, b1 = a.B;
  if (a.beam) {
    b0 = (int)a.beam[n];
    b1 = b0 + 1;
  }
  // the reference's four sums per beam: sum r wr, sum i wi, sum i wr, sum r wi (enh.py:372-380)
  auto emit = [&](int b, float rr, float ii, float ir, float ri) {
    if (!live) return;
    const int64_t o = a.beam ? (n * a.F + f) * a.T + t : ((n * a.B + b) * a.F + f) * a.T + t;
    a.yr[o] = rr + ii;
    a.yi[o] = ir - ri;
  };
  if (a.C <= kBeamC) {  // the usual arrays: the frame's channels are loaded once for all beams
    float r[kBeamC], i[kBeamC];
#pragma unroll
    for (int c = 0; c < kBeamC; ++c) {
      const int cc = min(c, a.C - 1);
      r[c] = xr[cc * chan];
      i[c] = xi[cc * chan];
    }
    for (int b = b0; b < b1; ++b) {
      float rr = 0.f, ii = 0.f, ir = 0.f, ri = 0.f;
#pragma unroll
      for (int c = 0; c < kBeamC; ++c)
        if (c < a.C) {
          const float wr = a.wr[((int64_t)b * a.C + c) * a.F + f];
          const float wi = a.wi[((int64_t)b * a.C + c) * a.F + f];
          rr += r[c] * wr, ii += i[c] * wi, ir += i[c] * wr, ri += r[c] * wi;
        }
      emit(b, rr, ii, ir, ri);
    }
    return;
  }
  for (int b = b0; b < b1; ++b) {  // wide arrays: channels re-read per beam (L1 / L2 hits)
    float rr = 0.f, ii = 0.f, ir = 0.f, ri = 0.f;
    for (int c = 0; c < a.C; ++c) {
      const float r = xr[c * chan], i = xi[c * chan];
      const float wr = a.wr[((int64_t)b * a.C + c) * a.F + f];
      const float wi = a.wi[((int64_t)b * a.C + c) * a.F + f];
      rr += r * wr, ii += i * wi, ir += i * wr, ri += r * wi;
    }
    emit(b, rr, ii, ir, ri);
  }
}

// ------------------------------------------------------------------------------------------
// Directional (angle) feature.  p: phase [N, C, T, F]; doa[n * doa_stride + d] in radians;
// neg_omega[f] = -pi sr f / (F - 1); out[(n * D + d) * T + t][out_off + f], row pitch ld_out.
// Geometry "7@" (enh.py:211-229): centre microphone 0 and six on a circle of radius 0.0425 m,
//   tau = R [0, -cos a, -cos(pi/3 - a), -cos(2pi/3 - a), cos a, cos(pi/3 - a), cos(2pi/3 - a)] / v
//   phi[c, f] = tau[c] * (-omega[f]),  dif_p = phi[l_p] - phi[r_p],  af = mean_p cos(ipd_p - dif_p)
// A workgroup owns one utterance and a tile of 4 frames; a thread takes every 256th (frame, bin)
// cell of the tile, keeps the P pair differences of 4 cells in registers (their loads in flight
// together) and sweeps the D directions over them.
// ------------------------------------------------------------------------------------------
constexpr int kDfMaxPairs = 16, kDfMaxDoas = 64, kDfFrames = 4, kDfUnroll = 4;

struct DfArgs {
  const float* p;
  const float* doa;
  const float* neg_omega;
  float* out;
  int64_t N, T, doa_stride, ld_out, out_off;
  int C, F, D, P;
  float radius, velocity;
  int l[kDfMaxPairs], r[kDfMaxPairs];
};

__global__ __launch_bounds__(256) void directional_feature_kernel(DfArgs a) {
  __shared__ float s_tau[kDfMaxDoas][8];
  const int64_t n = blockIdx.y;
  const int64_t t0 = (int64_t)blockIdx.x * kDfFrames;
  for (int d = threadIdx.x; d < a.D; d += 256) {
    const float ang = a.doa[n * a.doa_stride + d];
    const float k_pi = 3.14159265358979323846f;  // MATH_PI of the reference rounds to this float
    const float c0 = cosf(ang), c1 = cosf(k_pi / 3 - ang), c2 = cosf(2 * k_pi / 3 - ang);
    const float g[7] = {0.f, -c0, -c1, -c2, c0, c1, c2};
#pragma unroll
    for (int c = 0; c < 7; ++c) s_tau[d][c] = a.radius * g[c] / a.velocity;
  }
  __syncthreads();
  // the tile's (frame, bin) cells as one flat range: coalesced across row ends (F = 257 would
  // leave a second workgroup per row with a single bin), kDfUnroll cells per thread in flight
  const int64_t chan = a.T * a.F;
  const int rows = (int)min((int64_t)kDfFrames, a.T - t0);
  const int cells = rows * a.F;
  const float inv_p = 1.0f / (float)a.P;
  const float* pn = a.p + n * a.C * chan + t0 * a.F;
  for (int e0 = threadIdx.x; e0 < cells; e0 += 256 * kDfUnroll) {
    float ipd[kDfUnroll][kDfMaxPairs];
#pragma unroll
    for (int u = 0; u < kDfUnroll; ++u) {
      const int e = min(e0 + 256 * u, cells - 1);
#pragma unroll
      for (int q = 0; q < kDfMaxPairs; ++q)
        if (q < a.P) ipd[u][q] = pn[a.l[q] * chan + e] - pn[a.r[q] * chan + e];
    }
#pragma unroll
    for (int u = 0; u < kDfUnroll; ++u) {
      const int e = e0 + 256 * u;
      if (e >= cells) break;
      const int k = e / a.F, f = e - k * a.F;
      const float nw = a.neg_omega[f];
      for (int d = 0; d < a.D; ++d) {
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < kDfMaxPairs; ++q)
          if (q < a.P) {
            const float dif = s_tau[d][a.l[q]] * nw - s_tau[d][a.r[q]] * nw;
            acc += __cosf(ipd[u][q] - dif);  // v_cos_f32: |error| ~1e-6 on |x| < 256 rad
          }
        a.out[((n * a.D + d) * a.T + t0 + k) * a.ld_out + a.out_off + f] = acc * inv_p;
      }
    }
  }
}

}  // namespace aps

extern "C" int aps_fixed_beamform(const float* real, const float* imag, const float* w_real,
                                  const float* w_imag, const int64_t* beam, float* out_real,
                                  float* out_imag, int64_t N, int64_t C, int64_t F, int64_t T,
                                  int64_t B, void* stream) {
  APS_CHECK_ARG(real && imag && w_real && w_imag && out_real && out_imag);
  APS_CHECK_ARG(N > 0 && C > 0 && F > 0 && T > 0 && B > 0 && N <= 65535 && F <= INT32_MAX &&
                C <= 4096 && B <= INT32_MAX && (T + 255) / 256 <= 65535);
  aps::BeamArgs a{real, imag, w_real, w_imag, beam, out_real, out_imag, N, T, (int)C, (int)F, (int)B};
  dim3 grid((unsigned)F, (unsigned)N, (unsigned)((T + 255) / 256));
  hipLaunchKernelGGL(aps::fixed_beam_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return aps_launch_status();
}

extern "C" int aps_directional_feature(const float* phase, const float* doa, int64_t doa_stride,
                                       const float* neg_omega, const int32_t* index_l,
                                       const int32_t* index_r, int32_t num_pairs, float* out,
                                       int64_t ld_out, int64_t out_offset, int64_t N, int64_t C,
                                       int64_t T, int64_t F, int64_t D, float radius, float velocity,
                                       void* stream) {
  APS_CHECK_ARG(phase && doa && neg_omega && index_l && index_r && out);
  APS_CHECK_ARG(N > 0 && C > 0 && T > 0 && F > 0 && D > 0 && N <= 65535 && velocity > 0.f);
  if (num_pairs < 1 || num_pairs > aps::kDfMaxPairs || D > aps::kDfMaxDoas) return APS_ERR_UNSUPPORTED;
  APS_CHECK_ARG(ld_out >= out_offset + F && out_offset >= 0 && (T + aps::kDfFrames - 1) / aps::kDfFrames <= 65535);
  aps::DfArgs a{};
  a.p = phase, a.doa = doa, a.neg_omega = neg_omega, a.out = out;
  a.N = N, a.T = T, a.doa_stride = doa_stride, a.ld_out = ld_out, a.out_off = out_offset;
  a.C = (int)C, a.F = (int)F, a.D = (int)D, a.P = num_pairs;
  a.radius = radius, a.velocity = velocity;
  for (int q = 0; q < num_pairs; ++q) {
    // the "7@" delays exist for 7 microphones; the phase tensor must hold every indexed channel
    APS_CHECK_ARG(index_l[q] >= 0 && index_l[q] < 7 && index_r[q] >= 0 && index_r[q] < 7 &&
                  index_l[q] < C && index_r[q] < C);
    a.l[q] = index_l[q], a.r[q] = index_r[q];
  }
  dim3 grid((unsigned)((T + aps::kDfFrames - 1) / aps::kDfFrames), (unsigned)N);
  hipLaunchKernelGGL(aps::directional_feature_kernel, grid, dim3(256), 0,
                     static_cast<hipStream_t>(stream), a);
  return aps_launch_status();
}
