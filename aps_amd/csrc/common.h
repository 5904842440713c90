// Shared device helpers + launch/error plumbing for the aps_amd HIP kernels (gfx950 only).
#ifndef APS_AMD_COMMON_H_
#define APS_AMD_COMMON_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/aps_amd.h"
#include "fft_core.h"

#define APS_WAVE 64

#define APS_CHECK_ARG(cond) \
  do {                      \
    if (!(cond)) return APS_ERR_INVALID; \
  } while (0)

static inline int aps_launch_status() {
  return hipGetLastError() == hipSuccess ? APS_OK : APS_ERR_LAUNCH;
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
  if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
    return false;
  done.set(dev, 1);
  return true;
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
