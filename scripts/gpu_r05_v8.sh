#!/bin/bash
set -u
O=gpurun_out/r05_v8
mkdir -p $O
echo "HSA_ENABLE_SDMA=${HSA_ENABLE_SDMA:-unset}"
for sd in 1 0; do
for w in head own; do
HSA_ENABLE_SDMA=$sd APS_HOST_INPUT_STREAM=$w timeout 300 python bench.py --no-cpu-baseline --merged-group 0 2> $O/bench_$w$sd.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d.get('host_input'); print('sdma=$sd', '$w', d['value'], d['ms_per_step'], h.get('value'), h.get('ms_per_step'), h.get('error'))"
done
done
