#!/bin/bash
set -u
O=gpurun_out/r02_quick
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace1 -o trace -- \
   python $R/bench.py --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --replicas 1 "$@" > $R/$O/trace1.log 2>&1)
python scripts/trace_by_grid.py $(find $O/trace1 -name "*kernel_trace.csv" | head -1) > $O/by_grid.txt
head -40 $O/by_grid.txt
rm -rf $O/trace1
