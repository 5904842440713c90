#!/bin/bash
set -u
O=gpurun_out/r06_m3
mkdir -p $O
APS_MEGA_TRACE=1 timeout 300 python scripts/mega_probe.py 12 2>&1 | grep -v amdgpu | tail -22 | tee $O/mega_probe_trace.txt
for w in 4 6; do
  timeout 300 python scripts/pipeline_stage_times.py $w 2 head 2>&1 | grep -v amdgpu | tee $O/stages_w$w.txt
done
