#!/bin/bash
set -u
O=gpurun_out/r04_s26; mkdir -p $O
timeout 600 python scripts/occupier_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/occupier_probe.txt
