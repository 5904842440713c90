#!/bin/bash
set -u
O=gpurun_out/r05_v4
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_replicas.py -q -m gpu -x 2>&1 | tail -3
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --merged-group 0 2> $O/bench.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], 'one', d.get('single_stream_ms_per_step'), 'x2', d['whole_step_replicas']['value'])"
done
