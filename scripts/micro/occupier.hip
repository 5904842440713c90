// A resident, mostly idle kernel of a chosen footprint: what does ANOTHER stream's work pay for sharing the
// chip with it?  (Round 4: the mask estimator's persistent LSTM costs the second batch's GEMMs 0.4+ ms per
// step in the two-batches-in-flight mode and neither its retry traffic nor its register count explained it;
// scripts/occupier_probe.py runs the step's GEMM sequence beside this kernel.)
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC scripts/micro/occupier.hip -o scripts/micro/occupier.so
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int REGS>
__global__ __launch_bounds__(256) void occupy_kernel(long long cycles, float* sink, const f32x4* src,
                                                     long long src_vec, int loads_per_round, int sleep) {
  extern __shared__ float s_occ[];
  float r[REGS];
#pragma unroll
  for (int i = 0; i < REGS; ++i) r[i] = (float)(threadIdx.x + i);
  if (threadIdx.x == 0) s_occ[0] = 1.f;
  // the allocation is what the kernel descriptor claims: touching the highest register claims them all
  if (REGS >= 200) asm volatile("v_mov_b32 v239, 0" ::: "v239");
  else if (REGS >= 100) asm volatile("v_mov_b32 v127, 0" ::: "v127");
  const long long t0 = wall_clock64();  // 100 MHz
  long long at = (long long)(blockIdx.x * 256 + threadIdx.x) * 97;
  while (wall_clock64() - t0 < cycles) {
    for (int k = 0; k < loads_per_round; ++k) {  // gather-like traffic: 16 bytes per lane, rows 2 KB apart
      at = (at + 1031) % src_vec;
      const f32x4 v = __builtin_nontemporal_load(src + at);
      r[k % REGS] += v.x + v.w;
    }
    for (int q = 0; q < sleep; ++q) __builtin_amdgcn_s_sleep(64);
#pragma unroll
    for (int i = 0; i < REGS; ++i) r[i] = r[i] * 1.0001f + 0.5f;  // keeps the registers live
  }
  float s = s_occ[0];
#pragma unroll
  for (int i = 0; i < REGS; ++i) s += r[i];
  if (s == 12345.678f) sink[0] = s;
}

extern "C" int occupy(int regs, int blocks, int lds_bytes, double ms, float* sink, const void* src,
                      long long src_bytes, int loads_per_round, int sleep, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long cycles = (long long)(ms * 1e5);  // wall_clock64: 100 MHz
  const f32x4* s4 = static_cast<const f32x4*>(src);
  const long long nvec = src_bytes / 16;
#define OCC(R)                                                                                          \
  {                                                                                                     \
    if (lds_bytes > 64 * 1024)                                                                          \
      hipFuncSetAttribute(reinterpret_cast<const void*>(&occupy_kernel<R>),                             \
                          hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);                       \
    hipLaunchKernelGGL((occupy_kernel<R>), dim3(blocks), dim3(256), lds_bytes, st, cycles, sink, s4,   \
                       nvec, loads_per_round, sleep);                                                   \
  }
  if (regs <= 32) OCC(24) else if (regs <= 128) OCC(100) else OCC(210)
#undef OCC
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
