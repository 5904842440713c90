#!/bin/bash
set -u
O=gpurun_out/r04_s31; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_spatial.py -x -q -m gpu 2>&1 | tail -6 | tee $O/pytest.txt
