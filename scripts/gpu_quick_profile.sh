#!/bin/bash
# default bench line + kernel-trace stats of the same command (one stream and two): gpurun_out/r02_quick/
set -u
O=gpurun_out/r02_quick
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python bench.py 2> $O/bench_joint.err | tail -1 > $O/bench_joint.json; cut -c1-400 $O/bench_joint.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_joint1 -o trace -- \
   python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --replicas 1 > $R/$O/bench_joint1_under_rocprof.json 2>&1)
rm -f $O/prof_joint1/*kernel_trace.csv
head -25 $(find $O/prof_joint1 -name "*kernel_stats.csv" | head -1) | cut -c1-150
