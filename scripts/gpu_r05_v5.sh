#!/bin/bash
set -u
O=gpurun_out/r05_v5
mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_joint.py -q -m gpu -x -k "staged or replay" 2>&1 | tail -5
timeout 300 python bench.py --no-cpu-baseline --merged-group 0 2> $O/bench.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], 'checks', d.get('replay_checks'), 'queues', d['config'].get('hardware_queues'))"
