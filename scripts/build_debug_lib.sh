#!/bin/bash
# libaps_amd_dbg.so: the library with -DAPS_DEBUG_DISTURBANCE (STFT LDS read-back counters, the 32-row
# split GEMM behind APS_SPLIT_TM=32); select it with APS_AMD_LIB.  Experiments only.
set -e
cd "$(dirname "$0")/../aps_amd/csrc"
mkdir -p _obj_dbg
for f in aps_core stft feats mvdr nn lstm context conv decoder spatial augment grad lstm_grad gemm_split; do
  [ -f $f.hip ] || continue
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mcode-object-version=5 -Wno-unused-value \
     -DAPS_DEBUG_DISTURBANCE -c $f.hip -o _obj_dbg/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC _obj_dbg/*.o -o libaps_amd_dbg.so
ls -la libaps_amd_dbg.so
