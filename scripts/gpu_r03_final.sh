#!/bin/bash
# Last visit of round 3: the whole GPU suite, smoke, the driver-style bench line (the full profile visit is
# scripts/gpu_r03_full.sh).
set -u
O=gpurun_out/r03_final
mkdir -p $O
export TMPDIR=/tmp
echo "== pytest -m gpu =="
timeout 600 python -m pytest tests -m gpu -q --tb=short > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
echo "== smoke =="
timeout 120 python __graft_entry__.py --smoke 2>&1 | tail -1 | tee $O/smoke.log
echo "== bench (driver style) =="
timeout 300 python bench.py --steps 20 --warmup 5 2> $O/bench_joint.err | tail -1 > $O/bench_joint.json; cut -c1-200 $O/bench_joint.json
