#!/bin/bash
set -u
O=gpurun_out/r05_t1
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_decoder.py -q -m gpu 2>&1 | tail -25 > $O/pytest_decoder.txt
cat $O/pytest_decoder.txt
timeout 600 python -m pytest tests/test_gpu_train.py -q -m gpu -k "additive_mask or cross_attention" 2>&1 | tail -15 > $O/pytest_masks.txt
cat $O/pytest_masks.txt
