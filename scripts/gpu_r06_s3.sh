#!/bin/bash
# Round 6, visit 4: the workgroup trace again without the in-kernel atomic (host-assigned slots), forms e and f
set -u
R=$(pwd)
O=gpurun_out/r06_s3
mkdir -p $O
for f in e f; do
  APS_PANEL_FORM=$f APS_AMD_LIB=$R/aps_amd/csrc/libaps_amd_ptrace.so timeout 900 python scripts/panel_trace_under_load.py > $O/panel_trace_under_load_$f.txt 2>&1
  grep -E "^====|N=512 K=512|launch span" $O/panel_trace_under_load_$f.txt | head -40
done
