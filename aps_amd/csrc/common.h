// Shared device helpers + launch/error plumbing for the aps_amd HIP kernels (gfx950 only).
#ifndef APS_AMD_COMMON_H_
#define APS_AMD_COMMON_H_

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>

#include "../../include/aps_amd.h"
#include "fft_core.h"

#define APS_WAVE 64

#define APS_CHECK_ARG(cond) \
  do {                      \
    if (!(cond)) return APS_ERR_INVALID; \
  } while (0)

static inline int aps_launch_status() {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return APS_OK;
  fprintf(stderr, "[aps_amd] HIP error after a launch: %s\n", hipGetErrorString(e));
  return APS_ERR_LAUNCH;
}

// One-time per-DEVICE state of a launcher (LDS opt-in done, residency capacity of a kernel): a
// process may drive several GPUs, and hipFuncSetAttribute / the occupancy query apply to the current
// device only.  Atomic words, so concurrent caller threads are safe (worst case the idempotent
// query / opt-in runs twice).  0 = not set yet.
constexpr int kApsMaxDevices = 64;
struct ApsPerDevice {
  std::atomic<int> v[kApsMaxDevices];
  int get(int dev) const { return v[dev].load(std::memory_order_acquire); }
  void set(int dev, int x) { v[dev].store(x, std::memory_order_release); }
};
static inline int aps_current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kApsMaxDevices) return -1;
  return dev;
}
// opt a kernel into `bytes` of dynamic LDS on the current device, once per device
static inline bool aps_lds_opt_in(ApsPerDevice& done, const void* kernel, int bytes) {
  const int dev = aps_current_device();
  if (dev < 0) return false;
  if (done.get(dev)) return true;
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    fprintf(stderr, "[aps_amd] LDS opt-in of %d bytes refused: %s\n", bytes, hipGetErrorString(e));
    return false;
  }
  done.set(dev, 1);
  return true;
}

// Fill `words` 32-bit words with `value` by a KERNEL.  hipMemsetAsync must not be used on a path that
// can be captured into a hipGraph: on ROCm 7.0 / 7.2 (torch 2.10) a memset node recorded on a stream
// that still had eager work queued in front of the capture stops executing from the third replay on
// (scripts/memset_node_repro.py: 398 of 400 replays leave the buffer untouched, any size from 4 B
// to 32 MB; a fill kernel in its place never fails).  That was the "replica corruption" of round 1:
// the LSTM's sentinel re-arm silently did nothing and consumers read the previous replay's values.
template <int kUnused = 0>
__global__ __launch_bounds__(256) void aps_fill_u32_kernel(uint32_t* __restrict__ p, uint32_t value,
                                                           size_t words) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t quads = words / 4;
  if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
    uint4* q = reinterpret_cast<uint4*>(p);
    const uint4 v = make_uint4(value, value, value, value);
    for (size_t k = i; k < quads; k += stride) q[k] = v;
    for (size_t k = quads * 4 + i; k < words; k += stride) p[k] = value;
  } else {
    for (size_t k = i; k < words; k += stride) p[k] = value;
  }
}
static inline int aps_fill_u32(void* p, uint32_t value, size_t words, hipStream_t st) {
  if (words == 0) return APS_OK;
  size_t blocks = (words / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL((aps_fill_u32_kernel<0>), dim3((unsigned)blocks), dim3(256), 0, st,
                     static_cast<uint32_t*>(p), value, words);
  return hipGetLastError() == hipSuccess ? APS_OK : APS_ERR_LAUNCH;
}

// float32 machine epsilon: aps/const.py:17 (EPSILON)
#define APS_EPSILON 1.1920928955078125e-07f

namespace aps {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ cf ld_cf(const float* p) {
  float2 v = *reinterpret_cast<const float2*>(p);
  return {v.x, v.y};
}

__device__ __forceinline__ void st_cf(float* p, cf v) {
  *reinterpret_cast<float2*>(p) = make_float2(v.re, v.im);
}

// Hardware transcendental forms (v_rcp_f32 / v_rsq_f32 / v_sqrt_f32 / v_log_f32, ~1 ulp): the
// feature kernels are VALU-issue bound (rocprof: SQ_ACTIVE_INST_VALU ~ 90 % of the issue slots),
// and the IEEE-exact library forms cost 8-20 instructions each for corner cases (denormals,
// last-bit rounding) that cannot occur on the operands used here or sit far below the 1e-4 bar.
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fast_log(float x) {  // natural log, x a normal float or 0 / inf
  return __builtin_amdgcn_logf(x) * 0.69314718055994531f;
}

// |x| of a complex value with the squares guarded against under / overflow only where needed:
// fp32 |X| of audio spectra lives in ~[1e-12, 1e6], whose squares are normal numbers.
__device__ __forceinline__ float cabs_fast(cf x) { return fast_sqrt(x.re * x.re + x.im * x.im); }

// x / |x| without overflow / underflow: scale by the larger component first, so the squared
// norm lies in [1, 2] and the raw v_rsq_f32 is exact to ~1 ulp.
__device__ __forceinline__ float2 unit_vector(cf x) {
  const float m = fmaxf(fabsf(x.re), fabsf(x.im));
  if (!(m > 0.f)) {
    // atan2(+-0, +0) = +-0 -> (1, 0); atan2(+-0, -0) = +-pi -> (-1, 0); NaN propagates
    const float r = (m == 0.f) ? (signbit(x.re) ? -1.f : 1.f) : m;
    return make_float2(r, (m == 0.f) ? 0.f : m);
  }
  const float inv = fast_rcp(m);
  const float r = x.re * inv, i = x.im * inv;
  const float s = fast_rsq(r * r + i * i);
  return make_float2(r * s, i * s);
}

// log(clamp(x, eps)) / log(lower_bound + x); the clamp propagates NaN like th.clamp
__device__ __forceinline__ float log_feature(float v, float eps, float lower_bound) {
  return (lower_bound > 0.f) ? fast_log(lower_bound + v) : fast_log(v < eps ? eps : v);
}

}  // namespace aps
#endif  // APS_AMD_COMMON_H_
