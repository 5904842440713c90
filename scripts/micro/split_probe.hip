// Accuracy probe for fp32 GEMMs evaluated on the bf16 matrix pipe: each fp32 operand is split by
// truncation into bf16 planes h + m + l (exact: 8 + 8 + 8 significant bits) and the product is the
// sum of 3 (h h, h m, m h) or 6 (+ h l, l h, m m) v_mfma_f32_32x32x16_bf16.  One wave per 32 x 32
// output tile, operands straight from global memory (no tiling: this measures arithmetic only).
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o split_probe.so split_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

struct Planes {
  s16x8 h, m, l;
};

__device__ inline Planes split8(const float* p) {
  Planes o;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float x = p[i];
    const uint32_t hb = __float_as_uint(x) & 0xffff0000u;
    const float r1 = x - __uint_as_float(hb);
    const uint32_t mb = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(mb);
    o.h[i] = (short)(hb >> 16);
    o.m[i] = (short)(mb >> 16);
    o.l[i] = (short)(__float_as_uint(r2) >> 16);
  }
  return o;
}

__device__ inline f32x16 mm(s16x8 a, s16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                 __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// fp16 two-plane split: h = rn_f16(x), l = rn_f16(x - h): 11 + 11 significant bits while x stays in
// the fp16 range; 3 products h h + h l + l h
struct HPlanes {
  f16x8 h, l;
};
__device__ inline HPlanes hsplit8(const float* p) {
  HPlanes o;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float x = p[i];
    const _Float16 h = (_Float16)x;
    o.h[i] = h;
    o.l[i] = (_Float16)(x - (float)h);
  }
  return o;
}

// mode 0: fp32 MFMA, 5: fp16 x3, 1: 3 products, 2: 6 products one accumulator, 3: 6 products, h h apart from
// the corrections, 4: 6 products, small terms first
extern "C" __global__ void split_probe_kernel(const float* A, const float* W, float* C, int M, int N,
                                              int K, int mode) {
  const int ln = threadIdx.x, r = ln & 31, kh = ln >> 5;
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  f32x16 acc = {0}, acc2 = {0};
  if (mode == 0) {
    for (int k = 0; k < K; k += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(size_t)(m0 + r) * K + k + kh],
                                                 W[(size_t)(n0 + r) * K + k + kh], acc, 0, 0, 0);
  } else if (mode == 5) {
    for (int k = 0; k < K; k += 16) {
      const HPlanes a = hsplit8(A + (size_t)(m0 + r) * K + k + kh * 8);
      const HPlanes b = hsplit8(W + (size_t)(n0 + r) * K + k + kh * 8);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h, b.l, acc2, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.l, b.h, acc2, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h, b.h, acc, 0, 0, 0);
    }
    for (int e = 0; e < 16; ++e) acc[e] += acc2[e];
  } else {
    for (int k = 0; k < K; k += 16) {
      const Planes a = split8(A + (size_t)(m0 + r) * K + k + kh * 8);
      const Planes b = split8(W + (size_t)(n0 + r) * K + k + kh * 8);
      if (mode == 1) {
        acc = mm(a.h, b.h, acc);
        acc = mm(a.h, b.m, acc);
        acc = mm(a.m, b.h, acc);
      } else if (mode == 2) {
        acc = mm(a.h, b.h, acc);
        acc = mm(a.h, b.m, acc);
        acc = mm(a.m, b.h, acc);
        acc = mm(a.h, b.l, acc);
        acc = mm(a.l, b.h, acc);
        acc = mm(a.m, b.m, acc);
      } else if (mode == 3) {
        acc = mm(a.h, b.h, acc);
        acc2 = mm(a.h, b.m, acc2);
        acc2 = mm(a.m, b.h, acc2);
        acc2 = mm(a.h, b.l, acc2);
        acc2 = mm(a.l, b.h, acc2);
        acc2 = mm(a.m, b.m, acc2);
      } else {
        acc2 = mm(a.m, b.m, acc2);
        acc2 = mm(a.h, b.l, acc2);
        acc2 = mm(a.l, b.h, acc2);
        acc = mm(a.h, b.m, acc);
        acc = mm(a.m, b.h, acc);
        acc = mm(a.h, b.h, acc);
      }
    }
    for (int e = 0; e < 16; ++e) acc[e] += acc2[e];
  }
  for (int e = 0; e < 16; ++e) {
    const int row = m0 + (e & 3) + 8 * (e >> 2) + 4 * kh;
    C[(size_t)row * N + n0 + r] = acc[e];
  }
}

extern "C" int split_probe(const float* A, const float* W, float* C, int M, int N, int K, int mode,
                           void* stream) {
  hipLaunchKernelGGL(split_probe_kernel, dim3(N / 32, M / 32), dim3(64), 0, (hipStream_t)stream, A, W,
                     C, M, N, K, mode);
  return (int)hipGetLastError();
}
