#!/bin/bash
# round 4, visit 10: the whole GPU suite on the current tree + smoke
set -u
O=gpurun_out/r04_s10
mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --tb=short > $O/pytest_gpu.log 2>&1; tail -15 $O/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -2 | tee $O/smoke.log
