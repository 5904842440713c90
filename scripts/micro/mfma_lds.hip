// What costs the GEMM K step?  16 x v_mfma_f32_32x32x2_f32 per step with, cumulatively:
//   mode 0: register operands only          mode 1: + 8 ds_read_b128 operand fetches per step
//   mode 2: + 4 ds_write_b128 per step      mode 3: + s_barrier per step
//   mode 4: + 4 buffer loads (L2 resident) per step
// 1 workgroup (4 waves) per CU; reports ns per step (16 MFMAs = 1024 cycles).
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void step_loop(float* out, const float* src, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 128 * 36];
  const int tid = threadIdx.x, ln = tid & 63, wv = tid >> 6;
  for (int i = tid; i < 2 * 128 * 36; i += 256) lds[i] = 1.0f + i * 1e-7f;
  __syncthreads();
  f32x16 acc0, acc1;
  for (int e = 0; e < 16; ++e) acc0[e] = acc1[e] = 0.f;
  const float* pa = lds + ((wv >> 1) * 32 + (ln & 31)) * 36 + (ln >> 5) * 4;
  const float* pb = lds + (64 + (wv & 1) * 32 + (ln & 31)) * 36 + (ln >> 5) * 4;
  float* pw = lds + (tid >> 3) * 36 + (tid & 7) * 4;
  f32x4 g[4];
  for (int q = 0; q < 4; ++q) g[q] = f32x4{1.f, 2.f, 3.f, 4.f};
  const float* gp = src + (size_t)blockIdx.x * 4096 + tid * 4;
  for (int it = 0; it < iters; ++it) {
    const int buf = (it & 1) * 128 * 36;
    f32x4 a[4], b[4];
    if (MODE >= 1) {
      for (int q = 0; q < 4; ++q) {
        a[q] = *reinterpret_cast<const f32x4*>(pa + buf + q * 8);
        b[q] = *reinterpret_cast<const f32x4*>(pb + buf + q * 8);
      }
    } else {
      for (int q = 0; q < 4; ++q) a[q] = b[q] = g[q];
    }
    if (MODE >= 4) {
      for (int q = 0; q < 4; ++q)
        g[q] = *reinterpret_cast<const f32x4*>(gp + ((it + q) & 3) * 1024);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][0], b[q][0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][1], b[q][1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][2], b[q][2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][3], b[q][3], acc1, 0, 0, 0);
    }
    if (MODE >= 2) {
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<f32x4*>(pw + (128 * 36 - buf) + q * 32 * 36) = g[q];
    }
    if (MODE >= 3) __syncthreads();
  }
  float s = 0.f;
  for (int e = 0; e < 16; ++e) s += acc0[e] + acc1[e];
  out[blockIdx.x * 256 + tid] = s + g[0][0];
}

template <int MODE>
static void run(int cus, const float* src) {
  const int iters = 20000;
  float* out;
  hipMalloc(&out, (size_t)cus * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(step_loop<MODE>, dim3(cus), dim3(256), 0, 0, out, src, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(step_loop<MODE>, dim3(cus), dim3(256), 0, 0, out, src, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  printf("mode %d, %3d workgroups: %6.1f ns per step of 16 MFMAs (%5.1f %% of 1024 cycles at 2.4 GHz)\n",
         MODE, cus, ms * 1e6 / iters, 100.0 * (1024.0 / 2.4) / (ms * 1e6 / iters));
  hipFree(out);
}

int main() {
  float* src;
  hipMalloc(&src, (size_t)256 * 4096 * sizeof(float));
  hipMemset(src, 0, (size_t)256 * 4096 * sizeof(float));
  for (int cus : {1, 256}) {
    run<0>(cus, src);
    run<1>(cus, src);
    run<2>(cus, src);
    run<3>(cus, src);
    run<4>(cus, src);
  }
  return 0;
}
