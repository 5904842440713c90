#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, kernel-trace profile.  Outputs -> gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu ==" 
timeout 900 python -m pytest tests -m gpu -q --tb=short ${PYARGS:-} > gpurun_out/pytest_gpu.log 2>&1; grep -E '^(FAILED|ERROR|[0-9]+ (passed|failed))|AssertionError|Error:' gpurun_out/pytest_gpu.log | tail -${PYTAIL:-60}
echo "== smoke =="
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== bench =="
timeout 600 python bench.py --steps 200 --warmup 24 2>&1 | tail -3 | tee gpurun_out/bench.log
echo "== rocprofv3 kernel trace =="
rm -rf gpurun_out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o trace -- \
   python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 12 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1)
find gpurun_out/prof -name "*kernel_stats*" | head -3
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -30 "$f" | cut -c1-220
tail -2 gpurun_out/prof_bench.log
