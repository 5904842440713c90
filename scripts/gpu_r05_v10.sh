#!/bin/bash
set -u
O=gpurun_out/r05_v10
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_replicas.py tests/test_gpu_joint.py -q -m gpu -x -k "pipelined or staged" 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline --merged-group 0 2> $O/bench.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['launch']); print(d['host_input'])"
timeout 100 python scripts/pipeline_stage_times.py 3 2 head 2>&1 | grep -v amdgpu
