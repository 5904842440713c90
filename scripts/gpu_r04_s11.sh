#!/bin/bash
set -u
O=gpurun_out/r04_s11
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python scripts/stream_overlap_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/stream_overlap.txt
timeout 300 python -m pytest tests/test_mask_nonlinear.py tests/test_gpu_dccrn.py -m gpu -q --tb=short 2>&1 | tail -3
