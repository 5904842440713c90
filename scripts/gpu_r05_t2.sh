#!/bin/bash
# Round 5: kernel trace of the pipelined headline (3 worker streams + the LSTM stream)
set -u
R=$(pwd)
O=gpurun_out/r05_t2
mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/tr -o t -- \
   python $R/bench.py --merged-group 0 --steps 40 --warmup 5 --no-cpu-baseline > $R/$O/tr.log 2>&1)
f=$(find $O/tr -name "*kernel_trace.csv" | head -1)
g=$(find $O/tr -name "*kernel_stats.csv" | head -1)
[ -n "$g" ] && cp "$g" $O/joint32_pipeline_kernel_stats.csv
grep '^{"metric"' $O/tr.log | tail -1 > $O/joint32_pipeline_line_under_rocprof.json
python scripts/trace_overlap.py "$f" 0.6 > $O/joint32_pipeline_overlap.txt 2>&1
head -2 "$f" | cut -c1-400
cat $O/joint32_pipeline_overlap.txt
python -c "
import json; d=json.loads(open('$O/joint32_pipeline_line_under_rocprof.json').read()); print(d['value'], d['ms_per_step'])"
rm -rf $O/tr
