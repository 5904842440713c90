#!/bin/bash
set -u
O=gpurun_out/r04_s27; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "step_recurrences or lstm" 2>&1 | tail -25 > $O/pytest.txt; tail -25 $O/pytest.txt
