#!/bin/bash
# The two libraries of the co-residency regression run (scripts/replica_diff.py):
#   libaps_amd_dist_nopk.so    shipped flags (no packed-fp32 instructions) + the 32-row bf16 GEMM trigger
#   libaps_amd_dist_pkstft.so  the same with ONLY the STFT compiled with packed-fp32 instructions
# Run after the normal build; select with APS_AMD_LIB, APS_GEMM_SPLIT_LAYOUT=1 APS_SPLIT_TM=32.
set -e
cd "$(dirname "$0")/../aps_amd/csrc"
HC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mcode-object-version=5 -w -I$PWD"
NP="-Xclang -target-feature -Xclang -packed-fp32-ops"
$HC $NP -DAPS_DEBUG_DISTURBANCE -c gemm_split.hip -o /tmp/dist_gemm_split_nopk.o 2>/dev/null
$HC -c stft.hip -o /tmp/dist_stft_pk.o
base=$(ls _obj/*.o | grep -v "/gemm_split.o" | grep -v "/stft.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $base /tmp/dist_gemm_split_nopk.o _obj/stft.o -o libaps_amd_dist_nopk.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $base /tmp/dist_gemm_split_nopk.o /tmp/dist_stft_pk.o -o libaps_amd_dist_pkstft.so
ls -la libaps_amd_dist_*.so
