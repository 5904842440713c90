#!/bin/bash
# round 4, visit 5: cycle trace of the panel kernel's workgroups
set -u
O=gpurun_out/r04_s5
mkdir -p $O
export TMPDIR=/tmp
export APS_AMD_LIB=$PWD/aps_amd/csrc/libaps_amd_ptrace.so
for args in "2016 512 512 1" "2016 1024 512 1 ln" "2016 1536 512 1 ln" "2016 1024 512 2 ln" "8064 1024 512 3 ln" "8064 1024 512 4 ln" "31872 2048 512 3"; do
  timeout 120 python scripts/panel_trace.py $args 2>&1 | grep -v amdgpu.ids | tee -a $O/trace.txt
done
