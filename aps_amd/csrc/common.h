// Shared device helpers + launch/error plumbing for the aps_amd HIP kernels (gfx950 only).
#ifndef APS_AMD_COMMON_H_
#define APS_AMD_COMMON_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

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

}  // namespace aps
#endif  // APS_AMD_COMMON_H_
