#!/bin/bash
set -u
O=gpurun_out/r05_v6
mkdir -p $O
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --merged-group 0 2> $O/bench.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('host_input'))"
done
tail -3 $O/bench.err
