#!/bin/bash
# round 4, visit 20: MVDR backward with the implicit noise mask; the MVDR / task tests that share the adjoint
set -u
O=gpurun_out/r04_s20
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_backward.py tests/test_gpu_parity.py tests/test_gpu_tasks.py -x -q -m gpu -k "mvdr or covar or implicit or ml or task or process_mask or stand_alone" 2>&1 | tail -15 > $O/pytest.txt; tail -6 $O/pytest.txt
