// Sustained v_mfma_f32_32x32x2_f32 issue rate on gfx950 (no memory traffic): NACC accumulators per
// wave, WPS wavefronts per SIMD.  Build + run:  hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak && ./mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
#pragma unroll
  for (int q = 0; q < NACC; ++q)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
  float a = a0 + threadIdx.x * 1e-6f, b = b0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
      acc[r % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[r % NACC], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < NACC; ++q)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[q][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
static void run(int wgs_per_cu, int cus) {
  const int iters = 4000, grid = wgs_per_cu * cus;
  float* out;
  hipMalloc(&out, (size_t)grid * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(256), 0, 0, out, 100, 1.0f, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)grid * 4 * iters * 16 * (2.0 * 32 * 32 * 2);
  const double per_mfma_ns = ms * 1e6 / ((double)iters * 16 * wgs_per_cu);
  printf("NACC %d, %d wave(s)/SIMD: %7.3f ms, %6.1f TFLOP/s, %5.1f ns per MFMA slot (64 cycles at %.2f GHz)\n",
         NACC, wgs_per_cu, ms, flop / ms / 1e9, per_mfma_ns, 64.0 / per_mfma_ns);
  hipFree(out);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("%s, %d CUs, clock %d MHz\n", p.name, p.multiProcessorCount, p.clockRate / 1000);
  for (int w = 1; w <= 3; ++w) {
    run<1>(w, p.multiProcessorCount);
    run<2>(w, p.multiProcessorCount);
    run<4>(w, p.multiProcessorCount);
  }
  return 0;
}
