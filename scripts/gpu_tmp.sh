#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "fused_with_features" 2>&1 | grep -E "Error|assert|differs|rel" | head -12
