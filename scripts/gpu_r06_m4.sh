#!/bin/bash
set -u
O=gpurun_out/r06_m4
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mega.py -x -q 2>&1 | tail -12
APS_MEGA_TRACE=1 timeout 300 python scripts/mega_probe.py 12 2>&1 | grep -v amdgpu | tail -26 | tee $O/mega_probe_trace.txt
