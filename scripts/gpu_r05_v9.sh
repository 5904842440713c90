#!/bin/bash
set -u
O=gpurun_out/r05_v9
mkdir -p $O
for cfg in "3 head" "2 own" "2 head"; do
set -- $cfg
APS_BENCH_PIPELINE=$1 APS_HOST_INPUT_STREAM=$2 timeout 300 python bench.py --no-cpu-baseline --merged-group 0 2> $O/bench.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d.get('host_input'); print('workers $1 copy on $2: resident', d['value'], d['ms_per_step'], 'host-fed', h.get('value'), h.get('ms_per_step'), h.get('error'))"
done
