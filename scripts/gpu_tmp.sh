#!/bin/bash
timeout 1200 python -m pytest tests/test_gpu_joint.py tests/test_gpu_replicas.py tests/test_gpu_mega.py -x -q 2>&1 | tail -5
