// Does the buffer descriptor's range check on gfx950 include the SCALAR offset of a raw buffer access?
// (gemm_fp16x2.hip's epilogue puts the output ROW into soffset and relies on "rows past M fall outside
// the descriptor".)  A 1 KB descriptor over a 2 KB allocation: stores at voffset 0 with soffset = 1024
// + 4 lane -- if the guard half changes, soffset is NOT range checked.
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/soffset_range.hip -o aps_amd/csrc/_micro/soffset_range
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void probe(uint32_t* buf, uint32_t* seen, int soff) {
  auto rsrc = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 1024u, 0x00020000);
  const int lane = threadIdx.x;
  // (a) store with everything in soffset; (b) load the same way
  __builtin_amdgcn_raw_buffer_store_b32(0xdeadbeefu, rsrc, lane * 4, soff, 0);
  seen[lane] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane * 4, soff, 0);
}

int main() {
  uint32_t *buf, *seen, h[512], hs[64];
  hipMalloc(&buf, 2048);
  hipMalloc(&seen, 256);
  for (int soff : {0, 1024, 1020}) {
    hipMemset(buf, 0x11, 2048);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, buf, seen, soff);
    hipDeviceSynchronize();
    hipMemcpy(h, buf, 2048, hipMemcpyDeviceToHost);
    hipMemcpy(hs, seen, 256, hipMemcpyDeviceToHost);
    int in = 0, out = 0, lz = 0;
    for (int i = 0; i < 256; ++i) in += h[i] == 0xdeadbeefu;
    for (int i = 256; i < 512; ++i) out += h[i] == 0xdeadbeefu;
    for (int i = 0; i < 64; ++i) lz += hs[i] == 0;
    printf("soffset %4d: %3d words written inside the descriptor, %3d OUTSIDE it; %2d of 64 loads returned 0\n",
           soff, in, out, lz);
  }
  printf("(soffset IS range checked iff the 'OUTSIDE' column stays 0)\n");
  return 0;
}
