#!/bin/bash
mkdir -p gpurun_out/r05_t3
timeout 300 python scripts/step_torch_ops.py > gpurun_out/r05_t3/torch_ops.txt 2>&1
cat gpurun_out/r05_t3/torch_ops.txt | tail -40
