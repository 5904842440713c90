#!/bin/bash
# Last visit of round 3 after the training-path additions (no forward kernel changed since
# scripts/gpu_r03_full.sh ran): the whole GPU suite, smoke, the driver-style bench line, the training line.
set -u
O=gpurun_out/r03_final
mkdir -p $O
export TMPDIR=/tmp
echo "== pytest -m gpu =="
timeout 1500 python -m pytest tests -m gpu -q --tb=short > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
echo "== smoke =="
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -2 | tee $O/smoke.log
echo "== bench (driver style) =="
timeout 600 python bench.py --steps 20 --warmup 5 2> $O/bench_joint.err | tail -1 > $O/bench_joint.json; cut -c1-400 $O/bench_joint.json
echo "== bench train =="
timeout 600 python bench.py --workload train 2> $O/bench_train.err | tail -1 > $O/bench_train.json; cut -c1-300 $O/bench_train.json
