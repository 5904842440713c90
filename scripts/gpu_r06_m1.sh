#!/bin/bash
set -u
O=gpurun_out/r06_m1
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mega.py -x -q 2>&1 | tail -25
timeout 600 python scripts/mega_probe.py 12 2>&1 | grep -v amdgpu | tee $O/mega_probe.txt
