#!/bin/bash
mkdir -p gpurun_out/r05_t4
for cfg in "3 2 head" "3 2 worker" "2 2 head" "4 2 head"; do
timeout 200 python scripts/pipeline_stage_times.py $cfg 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r05_t4/stage_times.txt
done
