#!/bin/bash
# libaps_amd_wg<k>.so: the shipped objects with gemm_fp16x2.hip compiled for k workgroups per CU
# (APS_FP16X2_MIN_WG); A/B runs on one box through APS_AMD_LIB.  Run after the normal build.
set -e
cd "$(dirname "$0")/../aps_amd/csrc"
for k in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mcode-object-version=5 -Wno-unused-value \
     -DAPS_FP16X2_MIN_WG=$k -c gemm_fp16x2.hip -o /tmp/gemm_fp16x2_wg$k.o
  objs=$(ls _obj/*.o | grep -v gemm_fp16x2.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/gemm_fp16x2_wg$k.o -o libaps_amd_wg$k.so
  ls -la libaps_amd_wg$k.so
done
