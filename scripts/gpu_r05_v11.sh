#!/bin/bash
set -u
O=gpurun_out/r05_v11
mkdir -p $O
for w in head worker; do
APS_HOST_INPUT_STREAM=$w timeout 300 python bench.py --no-cpu-baseline --merged-group 0 2> $O/bench.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['host_input']; print('copy on $w: resident', d['value'], d['ms_per_step'], 'host-fed', h.get('value'), h.get('ms_per_step'))"
done
APS_PIPE_MID=worker APS_HOST_INPUT_STREAM=head timeout 300 python bench.py --no-cpu-baseline --merged-group 0 2> $O/bench.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['host_input']; print('mid worker, copy on head: resident', d['value'], d['ms_per_step'], 'host-fed', h.get('value'), h.get('ms_per_step'))"
timeout 100 python scripts/pipeline_stage_times.py 3 2 head 2>&1 | grep -v amdgpu
