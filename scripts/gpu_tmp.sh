#!/bin/bash
R=$(pwd)
for lib in libaps_amd.so libaps_amd_noepi.so libaps_amd.so; do
  echo "== $lib"
  APS_AMD_LIB=$R/aps_amd/csrc/$lib APS_MEGA_TRACE=1 timeout 300 python scripts/mega_probe.py 12 2>&1 | grep -E "1 in flight|8 in flight|in all|ff1_up|ff1_dn0|qkv|staging"
done
