#!/bin/bash
# DCCRN (BASELINE configs[2]): bench line + rocprofv3 kernel stats.  Outputs -> gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 400 python bench.py --workload dccrn 2>&1 | tail -2 | tee gpurun_out/dccrn_bench.log
rm -rf gpurun_out/prof_dccrn
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_dccrn -o trace -- \
   python $GRAFT_REPO_ROOT/bench.py --workload dccrn --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_dccrn_bench.log 2>&1)
f=$(find gpurun_out/prof_dccrn -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -16 "$f" | cut -c1-200
tail -1 gpurun_out/prof_dccrn_bench.log
