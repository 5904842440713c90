#!/bin/bash
set -u
O=gpurun_out/r04_s21; mkdir -p $O
timeout 300 python scripts/gemm_sequence_overlap.py 2>&1 | grep -v amdgpu.ids | tee $O/gemm_sequence_overlap.txt
